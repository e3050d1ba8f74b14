// ref_math_harness.cpp -- builds oracle/_ref/libref_math.so FROM THE REFERENCE'S OWN HEADERS where they lie
// (/root/reference/src/common/angle_functions.hpp and interpolation.hpp).  These two headers are the only part of
// the hot path that compiles without lcm-gen output (they are self-contained templates / inline functions); the
// .cpp files on the path all include generated lcmtypes/*.hpp and are unbuildable in this image.
// TEST INFRASTRUCTURE: used by tests/test_oracle_pins.py to pin oracle/botlab_oracle.cpp's restatement of
// wrap_to_pi / angle_diff / angle_sum / interpolate_pose_by_time.  Nothing is copied from the reference.
#include <cstdint>
#include <common/angle_functions.hpp>
#include <common/interpolation.hpp>

struct HarnessPose { int64_t utime; float x, y, theta; };   // this harness's own pose record (24 bytes)

extern "C" {
float ref_wrap_to_pi(float a) { return wrap_to_pi(a); }
double ref_angle_diff(double l, double r) { return angle_diff(l, r); }
double ref_angle_sum(double a, double b) { return angle_sum(a, b); }
void ref_interpolate_pose(int64_t t, const HarnessPose* b, const HarnessPose* e, HarnessPose* out)
{
    *out = interpolate_pose_by_time(t, *b, *e);
}
}
