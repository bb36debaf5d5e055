import sys, numpy as np
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open3d_slam_private_amd import capi, synth
sc = synth.make_scene(400000, 1000000, seed=1236)
for n in (25000, 100000, 400000):
    for nohalo, G, nofuse in ((0,8,1),(0,8,0),(0,4,0)):
        p = capi.shipped_params(); p.disable_halo = nohalo; p.lanes_per_point = G; p.disable_fused = nofuse; p.fixed_iters = 20
        reg = capi.Registration(p)
        reg.set_target(sc.tgt_xyz, sc.tgt_nrm); reg.set_source(sc.src_xyz[:n], sc.src_nrm[:n])
        T, res = reg.register(np.eye(4))
        Tit = np.array(res.T_iter_last, np.float32).reshape(4, 4).T
        r1 = reg.profile_kernels(Tit, 20)
        print(f"n={n} G={G} nofuse={nofuse} stalls={res.n_band_stalls} converged: match={r1['match_ms']*1e3:.1f}us select={r1['select_ms']*1e3:.1f}us lin={r1['linearize_ms']*1e3:.1f}us loop={res.loop_ms:.3f}ms iters={res.iterations}", flush=True)
