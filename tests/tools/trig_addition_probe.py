"""The Monte-Carlo measurement behind the default fast path of k_mcl_main (bl_mcl.hip, ray_cells_fast): 1e10 random (pose theta,
ray theta) pairs through bl_debug_trig_addition_probe; prints the JSON kept as profiles/r04_trig_addition_probe.json."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import botlab_amd as bl
from botlab_amd._capi import check
ctx = bl.default_context()
ms, mc, eps, n = C.c_float(), C.c_float(), C.c_float(), C.c_uint64()
check(ctx.lib.bl_debug_trig_addition_probe(ctx.h, 10_000_000_000, 4, C.byref(ms), C.byref(mc), C.byref(eps), C.byref(n)))
print(json.dumps(dict(pairs_checked=int(n.value), max_sin_err=float(ms.value), max_cos_err=float(mc.value), eps_used=float(eps.value),
                      analytic_bound=8.2e-7, margin=float(eps.value) / max(float(ms.value), float(mc.value)) - 1.0,
                      command="python tests/tools/trig_addition_probe.py")))
