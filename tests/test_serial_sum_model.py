"""The serially-rounded float accumulator of estimatePosteriorPose (particle_filter.cpp:151-152) as the kernels reproduce it
(botlab_amd/csrc/bl_serial_sum.h: records per predicted binade, composition scans, replays) -- the CPU model of the whole scheme,
lane for lane, against the plain loop on random and adversarial term sequences with honest, sloppy and nonsensical binade
predictions.  No GPU."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_scheme_equals_plain_loop(tmp_path):
    exe = str(tmp_path / "serial_sum_model")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-o", exe, os.path.join(ROOT, "tests", "cpp", "serial_sum_model.cpp")])
    out = subprocess.check_output([exe, "30"]).decode()
    fields = dict(zip(out.split()[0::2], out.split()[1::2]))
    assert fields["failures"] == "0" and int(fields["cases"]) >= 900, out
    assert int(fields["fitted"]) > 0 and int(fields["replays"]) > 0 and int(fields["exact_steps"]) > 0, out
