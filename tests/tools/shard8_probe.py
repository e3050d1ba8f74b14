"""Debug probe: eight in-process ranks (peer-store exchange) against one rank; prints where they differ."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch.multiprocessing as mp
import test_gpu_sharded_two_ranks as T

if __name__ == "__main__":
    n, steps, world = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    out = tempfile.mkdtemp()
    mp.spawn(T._run_single, args=(out, n, steps), nprocs=1, join=True)
    mp.spawn(T._run_inproc, args=(world, out, n, steps), nprocs=1, join=True)
    e1 = np.load(os.path.join(out, "est_w1_r0.npy"))
    one = np.load(os.path.join(out, "parts_w1_r0.npy"))
    parts = []
    for r in range(world):
        e = np.load(os.path.join(out, f"est_w{world}_r{r}.npy"))
        print("rank", r, "est equal per step:", [bool((e[k] == e1[k]).all()) for k in range(len(e1))], "theta diff", [float(e[k][3] - e1[k][3]) for k in range(len(e1))])
        parts.append(np.load(os.path.join(out, f"parts_w{world}_r{r}.npy")))
    allp = np.concatenate(parts)
    print("particles equal:", allp.tobytes() == one.tobytes())
    if allp.tobytes() != one.tobytes():
        a = allp.view(np.uint8).reshape(len(allp), -1); b = one.view(np.uint8).reshape(len(one), -1)
        bad = np.nonzero((a != b).any(axis=1))[0]
        print("differing particles:", len(bad), bad[:10], bad[-5:])
