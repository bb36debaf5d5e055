"""Soak run (not part of the test suite): many registrations on ONE handle with changing readings, priors and maps;
every 10th one is repeated on a second handle with the fused path off and must give the same pose bit for bit.
usage: python tools/tools_soak.py [n_registrations] [seed]   (GPU box)"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import torch  # noqa: F401,E402
from open3d_slam_private_amd import capi, synth  # noqa: E402

n_reg = int(sys.argv[1]) if len(sys.argv) > 1 else 500
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
scenes = [synth.make_scene(60_000, 600_000, seed=s) for s in (5, 6)]
p = capi.shipped_params()
q = capi.shipped_params()
q.disable_fused = 1
reg, ref = capi.Registration(p), capi.Registration(q)
cur = -1
t0 = time.time()
stalls = bad = checked = 0
by_size = {}
for i in range(n_reg):
    if i % 50 == 0:
        cur = (cur + 1) % len(scenes)
        sc = scenes[cur]
        reg.set_target(sc.tgt_xyz, sc.tgt_nrm)
        ref.set_target(sc.tgt_xyz, sc.tgt_nrm)
    lo = int(rng.integers(0, 20_000))
    hi = int(rng.integers(lo + 5_000, 60_000))
    T0 = np.eye(4, dtype=np.float32)
    scale = float(rng.choice([0.002, 0.01, 0.05]))
    T0[:3, :3] = synth.rpy_to_R(*rng.normal(scale=scale, size=3))
    T0[:3, 3] = rng.normal(scale=5 * scale, size=3)
    reg.set_source(sc.src_xyz[lo:hi], sc.src_nrm[lo:hi])
    T, res = reg.register(T0)
    assert np.isfinite(T).all(), i
    stalls += res.n_band_stalls
    b = by_size.setdefault(((hi - lo) // 10_000, scale), [0, 0, 0])
    b[0] += 1
    b[1] += res.n_band_stalls
    b[2] += res.iterations
    if i % 10 == 0:
        ref.set_source(sc.src_xyz[lo:hi], sc.src_nrm[lo:hi])
        T2, res2 = ref.register(T0)
        checked += 1
        if not (np.array_equal(T, T2) and res.iterations == res2.iterations):
            bad += 1
            print(f"MISMATCH at {i}: iterations {res.iterations} vs {res2.iterations}", flush=True)
    if i % 100 == 99:
        print(f"{i + 1} registrations, {time.time() - t0:.1f} s, band stalls {stalls}, checked {checked}, mismatches {bad}", flush=True)
print(f"done: {n_reg} registrations, band stalls {stalls}, checked {checked}, mismatches {bad}")
for k in sorted(by_size):
    c, st, it = by_size[k]
    print(f"  reading {k[0] * 10}-{k[0] * 10 + 10} k points, prior scale {k[1]}: {c} registrations, {st / c:.2f} stalls, {it / c:.1f} iterations each")
sys.exit(1 if bad else 0)
