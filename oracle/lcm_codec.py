"""CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE) for row f1 of SURVEY.md section 8: the LCM wire format of the seven
message types on the hot path's boundary, restated independently of botlab_amd/csrc/bl_lcm.hip (struct.pack, and the
fingerprints computed from the member lists below).

LCM 1.4.0 (docker/Dockerfile:30-31 of the reference) is not vendored there and not installed here; its lcm-gen and
liblcm are the "third-party algorithm" of this row.  What is restated is LCM's published format:
  * lcm-gen's struct hash (lcmgen.c, lcm_struct_hash / hash_update / hash_string_update): v = 0x12345678; per member the
    member name, the type name if primitive, the number of dimensions, and per dimension its mode (0 constant, 1 variable)
    and size string; v = ((v << 8) ^ (v >> 55)) + c per character with a signed 64-bit v;
  * the generated _computeHash: base + the hashes of struct-typed members, rotated left by one bit;
  * encode(): fingerprint, then the members in order, big-endian, arrays element by element, nested structs without a
    fingerprint; log events as written by lcm-logger (sync 0xEDA1DA01, event number, timestamp, lengths, channel, data).
PARITY UNPINNED: the reference checkout holds no LCM-encoded byte (its .log files are absent) and no generated lcmtypes
header or class, so nothing here could be checked against real LCM output.

Member lists: lcmtypes/{pose_xyt_t,odometry_t,lidar_t,particle_t,particles_t,occupancy_grid_t,robot_path_t}.lcm of the
reference (tests/golden/make_lcm_fixture.py re-reads those files and checks this table against them)."""
import struct

PRIMITIVES = {"int8_t": "b", "int16_t": "h", "int32_t": "i", "int64_t": "q", "float": "f", "double": "d", "byte": "B", "boolean": "b"}

# name -> [(member, type, variable-length dimension or None)]
TYPES = {
    "pose_xyt_t": [("utime", "int64_t", None), ("x", "float", None), ("y", "float", None), ("theta", "float", None)],
    "odometry_t": [("utime", "int64_t", None), ("x", "float", None), ("y", "float", None), ("theta", "float", None)],
    "lidar_t": [("utime", "int64_t", None), ("num_ranges", "int32_t", None), ("ranges", "float", "num_ranges"),
                ("thetas", "float", "num_ranges"), ("times", "int64_t", "num_ranges"), ("intensities", "float", "num_ranges")],
    "particle_t": [("pose", "pose_xyt_t", None), ("parent_pose", "pose_xyt_t", None), ("weight", "double", None)],
    "particles_t": [("utime", "int64_t", None), ("num_particles", "int32_t", None), ("particles", "particle_t", "num_particles")],
    "occupancy_grid_t": [("utime", "int64_t", None), ("origin_x", "float", None), ("origin_y", "float", None),
                         ("meters_per_cell", "float", None), ("width", "int32_t", None), ("height", "int32_t", None),
                         ("num_cells", "int32_t", None), ("cells", "int8_t", "num_cells")],
    "robot_path_t": [("utime", "int64_t", None), ("path_length", "int32_t", None), ("path", "pose_xyt_t", "path_length")],
}
MASK = (1 << 64) - 1


def _signed(v):
    v &= MASK
    return v - (1 << 64) if v >> 63 else v


def _hash_update(v, c):
    return _signed(((v << 8) ^ (v >> 55)) + c)          # v is a signed 64-bit value: >> is arithmetic


def _hash_string_update(v, s):
    v = _hash_update(v, len(s))
    for ch in s.encode():
        v = _hash_update(v, ch)
    return v


def base_hash(name):
    v = 0x12345678
    for member, typ, dim in TYPES[name]:
        v = _hash_string_update(v, member)
        if typ in PRIMITIVES:
            v = _hash_string_update(v, typ)
        v = _hash_update(v, 1 if dim else 0)
        if dim:
            v = _hash_update(v, 1)                      # LCM_VAR
            v = _hash_string_update(v, dim)
    return v


def fingerprint(name):
    h = base_hash(name) & MASK
    for _, typ, _ in TYPES[name]:
        if typ not in PRIMITIVES:
            h = (h + _pre_rotation(typ)) & MASK
    return ((h << 1) & MASK) + (h >> 63)


def _pre_rotation(name):
    # generated code adds the nested type's _computeHash(), i.e. its ROTATED hash
    return fingerprint(name)


def encode_body(name, msg):
    """msg: dict member -> value (lists for arrays, dicts for nested structs)."""
    out = b""
    for member, typ, dim in TYPES[name]:
        v = msg[member]
        items = v if dim else [v]
        if dim:
            assert len(items) == msg[dim], (member, len(items), msg[dim])
        for it in items:
            out += struct.pack(">" + PRIMITIVES[typ], it) if typ in PRIMITIVES else encode_body(typ, it)
    return out


def encode(name, msg):
    return struct.pack(">Q", fingerprint(name)) + encode_body(name, msg)


def decode(name, data):
    assert struct.unpack_from(">Q", data, 0)[0] == fingerprint(name)
    msg, off = _decode_body(name, data, 8)
    assert off == len(data)
    return msg


def _decode_body(name, data, off):
    msg = {}
    for member, typ, dim in TYPES[name]:
        n = msg[dim] if dim else 1
        items = []
        for _ in range(n):
            if typ in PRIMITIVES:
                items.append(struct.unpack_from(">" + PRIMITIVES[typ], data, off)[0])
                off += struct.calcsize(PRIMITIVES[typ])
            else:
                it, off = _decode_body(typ, data, off)
                items.append(it)
        msg[member] = items if dim else items[0]
    return msg, off


def log_event(event_number, timestamp_us, channel, data):
    ch = channel.encode()
    return struct.pack(">IqqII", 0xEDA1DA01, event_number, timestamp_us, len(ch), len(data)) + ch + data
