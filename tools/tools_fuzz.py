"""Randomised parity sweep (not part of the test suite): random scene sizes, bin sizes, max distances, trim ratios and
initial guesses; correspondences of one linearisation bit-exact against the oracle's kd-tree, final pose within
1e-4 m / 1e-4 rad.  usage: python tools/tools_fuzz.py [n_cases] [seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (import order: torch before the library)
from oracle import oracle as orc
from open3d_slam_private_amd import capi, synth
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from test_gpu_parity import _check_linearize

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(n_cases):
    n_src = int(rng.choice([37, 300, 2000, 9000, 30000]))
    n_tgt = int(rng.choice([50, 700, 20000, 150000, 400000]))
    sc = synth.make_scene(n_src, n_tgt, seed=int(rng.integers(1, 10_000)))
    p = capi.shipped_params()
    p.max_dist = float(rng.choice([0.05, 0.2, 0.5, 2.0, np.inf]))
    p.trim_ratio = float(rng.choice([0.5, 0.75, 0.9, 1.0]))
    p.use_trimmed = int(rng.integers(0, 2))
    p.cell_size = float(rng.choice([0.0, 0.0, 0.07, 0.4]))
    p.lanes_per_point = int(rng.choice([0, 0, 4]))
    p.max_iter = 25
    T0 = np.eye(4, dtype=np.float32)
    if rng.integers(0, 2):
        a = rng.normal(scale=0.01, size=3)
        T0[:3, :3] = synth.rpy_to_R(*a)
        T0[:3, 3] = rng.normal(scale=0.05, size=3)
    reg = capi.Registration(p)
    reg.set_target(sc.tgt_xyz, sc.tgt_nrm)
    reg.set_source(sc.src_xyz, sc.src_nrm)
    ok = True
    try:
        reg.prepare(np.eye(4))
        Ti = np.eye(4, dtype=np.float32)
        Ti[:3, :3] = synth.rpy_to_R(*rng.normal(scale=0.01, size=3))
        Ti[:3, 3] = rng.normal(scale=0.05, size=3)
        try:
            # bit-exact ids / d2 / weights, H and b of one linearisation at a random pose (asserts inside)
            _check_linearize(reg, sc.tgt_xyz, sc.tgt_nrm, sc.src_xyz, sc.src_nrm, p.max_dist,
                             p.trim_ratio if p.use_trimmed else None, p.max_normal_angle, T_iter=Ti)
            same_ids = same_d2 = True
        except AssertionError as e:
            same_ids = same_d2 = False
            print("   linearize mismatch:", str(e)[:200])
        except RuntimeError as e:
            if isinstance(e, capi.RegError):
                if e.status != 3:
                    raise
                # no pair survives at this random pose (tiny clouds, small max_dist): nothing to compare in this step
                same_ids = same_d2 = True
                print("   (no pairs at the random linearisation pose)")
                e = None
            if e is None:
                pass
            else:
                # the oracle found nothing to work with at this pose: the product must say the same
                try:
                    reg.linearize(Ti)
                    same_ids = same_d2 = False
                except capi.RegError as ge:
                    same_ids = same_d2 = ge.status == 3
        ok = same_ids
        g_fail = o_fail = False
        try:
            T, res = reg.register(T0)
        except capi.RegError as ge:
            g_fail = ge.status == 3
            if not g_fail:
                raise
        try:
            To, ores = orc.icp_p2pl(sc.tgt_xyz, sc.tgt_nrm, sc.src_xyz, sc.src_nrm, T0, max_dist=p.max_dist,
                                    trim_ratio=p.trim_ratio if p.use_trimmed else None,
                                    max_normal_angle=p.max_normal_angle, max_iter=p.max_iter, min_diff_rot=p.min_diff_rot,
                                    min_diff_trans=p.min_diff_trans, smooth_len=p.smooth_len, n_threads=16)
            o_fail = ores.status == 3
        except RuntimeError:
            o_fail = True
        if g_fail or o_fail:
            ok = ok and g_fail == o_fail
            detail = f"ids {same_ids} no correspondences: product {g_fail} oracle {o_fail}"
        else:
            dt, dr = synth.pose_error(T, To)
            ok = ok and dt <= 1e-4 and dr <= 1e-4 and res.iterations == ores.iterations
            detail = f"ids {same_ids} d2 {same_d2} pose {dt:.1e}/{dr:.1e} it {res.iterations}/{ores.iterations}"
    except capi.RegError as e:
        detail = f"unexpected RegError {e.status}: {e}"
        ok = False
    print(f"case {case:2d}: n {n_src:6d}->{n_tgt:6d} md {p.max_dist:4.2f} trim {p.use_trimmed}/{p.trim_ratio:.2f} cell {p.cell_size:.2f} "
          f"lanes {p.lanes_per_point} : {'ok ' if ok else 'BAD'} {detail}", flush=True)
    bad += 0 if ok else 1
    reg.close()
print("mismatches:", bad)
sys.exit(1 if bad else 0)
