"""Per-iteration search statistics of the select-based iterations (diagnostic build: tools/build_ab.sh stats -DO3D_SEARCH_STATS=1,
O3D_REG_LIB=open3d_slam_private_amd/lib_ab/libstats.so python tools/tools_search_stats.py [n_src n_tgt seed]).  GPU box only."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, ".")
from open3d_slam_private_amd import capi, synth  # noqa: E402

NAMES = ["level scans", "bricks in boxes", "non-empty rows in boxes", "rows kept", "candidates", "batches", "halo candidates", "searches"]


def stats(lib, reset=True):
    out = (ctypes.c_ulonglong * 64)()
    assert lib.o3d_debug_search_stats(out, 1 if reset else 0) == 0
    return np.array(list(out), dtype=np.float64)


if __name__ == "__main__":
    n_src = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    n_tgt = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    lib = ctypes.CDLL(os.environ["O3D_REG_LIB"])
    sc = synth.make_scene(n_src, n_tgt, seed=seed)
    p = capi.shipped_params()
    p.disable_fused = 1
    prev = np.zeros(64)
    for k in range(1, 4):
        p.fixed_iters = k
        reg = capi.Registration(p)
        reg.set_target(sc.tgt_xyz, sc.tgt_nrm)
        reg.set_source(sc.src_xyz, sc.src_nrm)
        stats(lib)
        reg.register(np.eye(4))
        cur = stats(lib)
        if k == 1:
            print("grid:", reg.info() if hasattr(reg, "info") else "")
        d = cur - prev
        prev = cur
        n = max(d[7], 1.0)
        print(f"iteration {k - 1}: " + ", ".join(f"{NAMES[i]} {d[i] / n:.1f}" for i in range(7)) + f" per search ({int(d[7])} searches)", flush=True)
        for nm, o in (("candidates", 8), ("rows kept", 24), ("bricks", 40)):
            hh = d[o:o + 16]
            print(f"    level scans by {nm} (0, 1, 2-3, 4-7, ...): " + " ".join(f"{int(v)}" for v in hh[:int(np.max(np.nonzero(hh)[0], initial=0)) + 1]))
        sec = d[56:63]
        print("    wave time by section (halo, box + directory, row slots, compaction, candidate scan, minimum + level logic, ball update): "
              + " ".join(f"{100 * v / max(sec.sum(), 1):.1f}%" for v in sec) + f"  total {sec.sum() * 0.01:.0f} wave-us")
        reg.close()
