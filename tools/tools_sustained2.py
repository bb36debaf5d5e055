"""Per-registration wall times of N back-to-back registrations: prints the outliers (> 1.3 x median) with their index."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import bench
from open3d_slam_private_amd import capi, synth
n_reg = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
wl = sys.argv[2] if len(sys.argv) > 2 else "c3"
n, m, s = bench.WORKLOADS[wl]
sc = synth.make_scene(n, m, seed=s)
ds = bench.DeviceScene(torch, torch.device("cuda", 0), sc)
reg = ds.make_reg(capi, bench.chain_params(capi, 0))
T0 = np.eye(4, dtype=np.float32)
w = np.zeros(n_reg); lm = np.zeros(n_reg)
for i in range(n_reg):
    t0 = time.perf_counter()
    _, res = reg.register(T0)
    w[i] = time.perf_counter() - t0
    lm[i] = res.loop_ms
med = np.median(w)
print(f"median wall {1e3 * med:.3f} ms, mean {1e3 * w.mean():.3f} ms, median loop {np.median(lm):.3f} ms")
out = np.nonzero(w > 1.3 * med)[0]
print("outliers:", [(int(i), round(1e3 * w[i], 2), round(float(lm[i]), 2)) for i in out][:60])
