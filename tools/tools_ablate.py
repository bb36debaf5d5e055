import sys, numpy as np
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open3d_slam_private_amd import capi, synth
sc = synth.make_scene(100000, 1000000, seed=1236)
for dbg in (0, 4):
    p = capi.shipped_params(); p.debug_flags = dbg
    reg = capi.Registration(p)
    reg.set_target(sc.tgt_xyz, sc.tgt_nrm); reg.set_source(sc.src_xyz, sc.src_nrm)
    p2 = capi.shipped_params()
    T, res = capi.Registration(p2).register(np.eye(4)) if False else (None, None)
    reg.prepare(np.eye(4))
    r = reg.profile_kernels(np.eye(4), 20)
    print("debug", dbg, {k: round(v*1e3,1) for k,v in r.items()}, flush=True)
