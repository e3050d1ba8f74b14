"""Generates tests/golden/lcm_types.json: the member lists (name, type, variable-length dimension) of the seven LCM
message types on the hot path's boundary, parsed from the reference's lcmtypes/*.lcm.  Run in the build container only
(the GPU box has no /root/reference); the JSON is data (an interface description), not source."""
import json
import os
import re

REF = "/root/reference/lcmtypes"
NAMES = ["pose_xyt_t", "odometry_t", "lidar_t", "particle_t", "particles_t", "occupancy_grid_t", "robot_path_t"]


def parse(path):
    text = re.sub(r"//[^\n]*", "", open(path).read())
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    body = text[text.index("{") + 1:text.rindex("}")]
    members = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        m = re.match(r"(\w+)\s+(\w+)\s*(?:\[\s*(\w+)\s*\])?$", decl)
        assert m, decl
        members.append([m.group(2), m.group(1), m.group(3)])
    return members


if __name__ == "__main__":
    out = {n: parse(os.path.join(REF, n + ".lcm")) for n in NAMES}
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "lcm_types.json"), "w") as f:
        json.dump(out, f, indent=1)
    print({k: len(v) for k, v in out.items()})
