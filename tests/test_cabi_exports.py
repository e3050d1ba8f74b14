"""The C-ABI library loads without a GPU and exports every symbol include/botlab_hip.h declares (no compute calls)."""
import ctypes
import os
import re

import botlab_amd
from botlab_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "botlab_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bl_[a-z0-9_]+)\s*\(", text)))


def test_library_loads_and_exports_every_declared_symbol():
    lib = _capi.load()
    names = _declared_symbols()
    assert len(names) >= 40
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/botlab_hip.h but not exported"
    # and the Python binding covers the same set
    assert sorted(_capi.SIGNATURES) == names
    assert lib.bl_version().startswith(b"botlab_hip")


def test_struct_layouts_match_lcm_types():
    # lcmtypes/pose_xyt_t.lcm, particle_t.lcm
    assert ctypes.sizeof(botlab_amd.Pose) == 24
    assert ctypes.sizeof(botlab_amd.Particle) == 56
    assert botlab_amd.Particle.weight.offset == 48
    assert botlab_amd.PARTICLE_DTYPE.itemsize == 56


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        return
    lib = _capi.load()
    h = ctypes.c_void_p()
    rc = lib.bl_ctx_create(0, None, ctypes.byref(h))
    assert rc != 0 and lib.bl_last_error()


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "botlab_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.lower().replace("no cpu fallback", ""), os.path.join(dirpath, f)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "include")):
        for f in files:
            assert "oracle" not in open(os.path.join(dirpath, f)).read().lower()
