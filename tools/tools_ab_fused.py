"""A/B of the fused kernels in one binary: debug_flags 16 = k_iter_fused (no temporal-coherence shortcut), 0 = k_iter_coherent.
Prints registration time, kernel times, the share of points the coherent kernel had to search (O3D_COH_STATS) and checks
that both give identical poses / ids / d2 / weights.   usage: python tools/tools_ab_fused.py c3"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["O3D_COH_STATS"] = "1"
from open3d_slam_private_amd import capi, synth
from bench import WORKLOADS, ITERS

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
n_src, n_tgt, seed = WORKLOADS[wl]
sc = synth.make_scene(n_src, n_tgt, seed=seed)
dev = torch.device("cuda", 0)
d_t, d_tn = torch.from_numpy(sc.tgt_xyz).to(dev), torch.from_numpy(sc.tgt_nrm).to(dev)
d_s, d_sn = torch.from_numpy(sc.src_xyz).to(dev), torch.from_numpy(sc.src_nrm).to(dev)
torch.cuda.synchronize()
ref = None
for flags in (16, 0, 16, 0):
    p = capi.shipped_params()
    p.use_xicp = 0
    p.fixed_iters = ITERS
    p.debug_flags = flags
    reg = capi.Registration(p)
    reg.set_target_device(d_t.data_ptr(), 3, n_tgt, d_tn.data_ptr(), 3)
    reg.set_source_device(d_s.data_ptr(), 3, n_src, d_sn.data_ptr(), 3)
    ms = []
    for k in range(8):
        T, res = reg.register(np.eye(4))
        if k >= 2:
            ms.append(res.loop_ms)
    ids, d2, w = reg.correspondences()
    reg.close()
    p.profile_loop = 1
    reg = capi.Registration(p)
    reg.set_target_device(d_t.data_ptr(), 3, n_tgt, d_tn.data_ptr(), 3)
    reg.set_source_device(d_s.data_ptr(), 3, n_src, d_sn.data_ptr(), 3)
    reg.register(np.eye(4))
    _, pr = reg.register(np.eye(4))
    reg.close()
    if ref is None:
        ref = (T, ids, d2, w)
    same = (np.array_equal(T, ref[0]), np.array_equal(ids, ref[1]), np.array_equal(d2.view(np.uint32), ref[2].view(np.uint32)),
            np.array_equal(w, ref[3]))
    km = pr.prof_ms[0] / max(pr.prof_launches[0], 1) * 1e3
    kf = (pr.prof_ms[1] + pr.prof_ms[2]) / max(pr.prof_launches[1], 1) * 1e3
    kf1 = pr.prof_ms[1] / max(pr.prof_launches[1], 1) * 1e3
    print(f"{wl} flags {flags:2d}: loop min {min(ms):.3f} med {sorted(ms)[len(ms) // 2]:.3f} ms | k_match {pr.prof_launches[0]:2d} x {km:6.1f} us, "
          f"fused {pr.prof_launches[1]:2d} x {kf:6.1f} us (check {kf1:5.1f}) | stalls {res.n_band_stalls} | same T/ids/d2/w {same}", flush=True)
