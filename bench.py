#!/usr/bin/env python3
"""Headline benchmark: ICP iterations/sec of scan-to-map registration on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3]

A "step" is ONE registration of the synthetic scan against the synthetic map with exactly
ITERS (=20) Gauss-Newton iterations of the reference's shipped chain (param/icp.yaml: knn 1, maxDist 0.5,
TrimmedDist 0.9, SurfaceNormal 1.57, point-to-plane): R2 reading prep + 20 x (R3 transform, R4 exact 1-NN,
R5 trimmed quantile + normal filter, R7 normal equations, R8 6x6 solve, R9 pose update) + R10.
Inputs are resident in HBM before the timed region; the target voxel-bin build (== kd-tree build of the
reference) is excluded on both sides and reported as target_build_ms.

N=1: workload C2 (100k -> 1M points, BASELINE.json configs[1]).
N>1: WEAK scaling -- every rank holds its own 100k-point slice of an N*100k-point reading (point
partitioned), the 1M-point map is replicated; per iteration the ranks all-reduce the trimmed-quantile
histograms and the 32-double (H, b, e, counts) record over RCCL.  `value` counts 100k-point iteration
units: N_gpus * ITERS * K / t.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ITERS = 20
WORKLOADS = {"c2": (100_000, 1_000_000, 1234 + 2), "c3": (200_000, 5_000_000, 1234 + 3),
             "tiny": (10_000, 100_000, 1234 + 1)}
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
# algorithmic bytes per reading point and launch:
#   k_match_g8  : src xyz 12 + matched tgt xyz 12 + pos 4 + d2 4 written            = 32
#   k_iter_fused: src xyz 12 + src normal 12 + matched tgt xyz 12 + tgt normal 12 (P2Pl) = 48
KERNEL_BYTES_PER_POINT = {"k_match_g8": 32, "k_iter_fused": 48}
ITER_BYTES_PER_POINT = 64    # P2Pl 48 B + 16 B (id, d2 written and re-read) for the split-kernel variant


def cpu_baseline(sc, n_src, threads):
    """The oracle (faithful C restatement; the reference itself cannot be built here) timed on the host cores:
    same clouds, same chain, kd-tree build excluded.  Bounded sample: the first `n_src` reading points."""
    from oracle import oracle as orc
    out = {}
    for name, nt, iters in (("omp", threads, ITERS), ("1t", 1, 4)):
        _, r = orc.icp_p2pl(sc.tgt_xyz, sc.tgt_nrm, sc.src_xyz[:n_src], sc.src_nrm[:n_src], max_dist=0.5,
                            trim_ratio=0.9, max_normal_angle=1.57, fixed_iters=iters, n_threads=nt)
        out[name] = iters / r.loop_seconds
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--streams", type=int, default=1,
                    help="extra measurement (config 5 analogue): S independent registrations in flight, one handle + "
                         "HIP stream + host thread each; reported under \"batched\", never as `value`")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with WORLD_SIZE={args.gpus} (got {world})")

    import torch  # plumbing only: device buffers, streams, torch.distributed
    from open3d_slam_private_amd import capi, synth

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the registration path has no CPU fallback")
    if os.environ.get("O3D_BENCH_BACKEND", "nccl") != "nccl":
        local_rank = local_rank % torch.cuda.device_count()   # rehearsal: several ranks share the one GPU
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("O3D_BENCH_BACKEND", "nccl")   # "gloo": rehearsal of the N>1 path on a 1-GPU box
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    n_src, n_tgt, seed = WORKLOADS[args.workload]
    sc = synth.make_scene(n_src * world, n_tgt, seed=seed)
    lo, hi = rank * n_src, (rank + 1) * n_src

    # inputs resident in HBM before anything is timed
    d_tgt = torch.from_numpy(sc.tgt_xyz).to(dev)
    d_tnrm = torch.from_numpy(sc.tgt_nrm).to(dev)
    d_src = torch.from_numpy(sc.src_xyz[lo:hi]).to(dev)
    d_snrm = torch.from_numpy(sc.src_nrm[lo:hi]).to(dev)
    torch.cuda.synchronize()

    p = capi.shipped_params()
    p.fixed_iters = ITERS
    p.device = local_rank
    reg = capi.Registration(p)
    reg.set_target_device(d_tgt.data_ptr(), 3, n_tgt, d_tnrm.data_ptr(), 3)
    reg.set_source_device(d_src.data_ptr(), 3, n_src, d_snrm.data_ptr(), 3)
    info = reg.target_info()
    T_init = np.eye(4, dtype=np.float32)

    force_dist = os.environ.get("O3D_BENCH_FORCE_DIST") == "1"   # rehearsal: N>1 code path with one rank
    if world == 1 and not force_dist:
        def step():
            return reg.register(T_init)
    else:
        # stream-ordered loop: kernels of the library and RCCL all-reduces share torch's current stream; the host
        # never synchronises inside a registration (open3d_slam_private_amd/distributed.py)
        from open3d_slam_private_amd.distributed import (FusedStreamDistributedRegistration,
                                                         StreamDistributedRegistration)
        coll_dev = dev
        reg.set_stream(torch.cuda.current_stream().cuda_stream)
        sreg = StreamDistributedRegistration(reg, True, ITERS, dist=dist, device=dev)
        ag = None
        if dist is not None and dist.get_backend() != "nccl":      # rehearsal backends: no all_gather_into_tensor
            def ag(out, inp):
                parts = list(out.view(world, -1).unbind(0))
                dist.all_gather(parts, inp)
        freg = FusedStreamDistributedRegistration(reg, True, p.trim_ratio, ITERS, world, rank, dist=dist, device=dev,
                                                  all_gather=ag)

        from open3d_slam_private_amd.distributed import _DevArray
        cent = torch.as_tensor(_DevArray(reg.dist_centroid_sums(), (3,), "<i8"), device=dev)

        def prep():
            # stream-ordered: this rank's integer centroid sums, all-reduced on the device, then centring and
            # pre-transform with the centroid of the WHOLE reading -- no host round trip
            reg.dist_centroid_sums()
            if dist is not None:
                dist.all_reduce(cent)
            reg.dist_prepare(T_init, n_src * world)

        # self-check before anything is timed: the fused loop (one all-gather per settled iteration) must reproduce
        # the select-based loop (four all-reduces per iteration); otherwise time the latter
        prep()
        T_a, _ = sreg.run()
        use_fused = True
        try:
            prep()
            T_b, _ = freg.run()
            dt, dr = synth.pose_error(T_a, T_b)
            use_fused = dt <= 1e-5 and dr <= 1e-5
        except Exception as e:   # noqa: BLE001 -- any failure of the optional fast loop selects the plain one
            if rank == 0:
                print(f"[bench] fused multi-GPU loop unavailable ({e!r}); timing the select-based loop", file=sys.stderr)
            use_fused = False
        if dist is not None:
            flag = torch.tensor([1 if use_fused else 0], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            use_fused = bool(flag.item())
        loop_kind = "fused (1 all-gather / settled iteration)" if use_fused else "select-based (4 all-reduces / iteration)"

        def step():
            prep()
            return (freg if use_fused else sreg).run()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    T_final = out[0]

    # kernel-level numbers (outside the timed region): one more registration with HIP events bracketing every
    # search kernel on the handle's own stream (params.profile_loop) -> average launch duration over the SAME
    # 20-iteration trajectory the timed region ran; rocprofv3 --kernel-trace of this command must agree.
    kern = {}
    if world == 1:
        p2 = capi.shipped_params()
        p2.fixed_iters = ITERS
        p2.device = local_rank
        p2.profile_loop = 1
        preg = capi.Registration(p2)
        preg.set_target_device(d_tgt.data_ptr(), 3, n_tgt, d_tnrm.data_ptr(), 3)
        preg.set_source_device(d_src.data_ptr(), 3, n_src, d_snrm.data_ptr(), 3)
        preg.register(T_init)
        _, pres = preg.register(T_init)
        for idx, name in ((0, "k_match_g8"), (1, "k_iter_fused")):
            if pres.prof_launches[idx]:
                kern[name] = {"launches": int(pres.prof_launches[idx]), "total_ms": float(pres.prof_ms[idx]),
                              "avg_ms": float(pres.prof_ms[idx]) / int(pres.prof_launches[idx])}
        preg.close()
    else:
        prof = reg.profile_kernels(np.eye(4, dtype=np.float32), reps=20)
        kern["k_match_g8"] = {"launches": 20, "total_ms": 20 * prof["match_ms"], "avg_ms": prof["match_ms"]}
    dom = max(kern, key=lambda k: kern[k]["total_ms"])
    bytes_pp = KERNEL_BYTES_PER_POINT[dom]
    achieved = n_src * bytes_pp / (kern[dom]["avg_ms"] * 1e-3) / 1e9

    # optional: S independent registrations concurrently (replicas only, no collective) -- SURVEY 8e / config 5
    batched = None
    if world == 1 and args.streams > 1:
        import threading
        regs = []
        for _ in range(args.streams):
            r = capi.Registration(p)
            r.set_target_device(d_tgt.data_ptr(), 3, n_tgt, d_tnrm.data_ptr(), 3)
            r.set_source_device(d_src.data_ptr(), 3, n_src, d_snrm.data_ptr(), 3)
            r.register(T_init)
            regs.append(r)
        per = max(2, args.steps // 2)
        go = threading.Barrier(args.streams + 1)

        def work(r):
            go.wait()
            for _ in range(per):
                r.register(T_init)          # ctypes releases the GIL: the host threads really run in parallel
            go.wait()

        ths = [threading.Thread(target=work, args=(r,)) for r in regs]
        for t in ths:
            t.start()
        torch.cuda.synchronize()
        go.wait()
        tb0 = time.perf_counter()
        go.wait()
        tb = time.perf_counter() - tb0
        for t in ths:
            t.join()
        batched = {"streams": args.streams, "registrations": args.streams * per,
                   "iter_per_s": args.streams * per * ITERS / tb, "ms_per_registration_amortised": 1e3 * tb / (args.streams * per)}
        for r in regs:
            r.close()

    # secondary figure: the GICP cost (the north star's cost function; SURVEY 8d: 72 B/pt + 16 B/pt for the split
    # kernels) on the same clouds, covariances from the analytic normals.  Never `value`.
    gicp = None
    if world == 1 and not force_dist and args.workload in ("c2", "tiny"):
        pg = capi.default_params()
        pg.cost = capi.COST_GICP
        pg.use_trimmed = 0
        pg.max_dist = 0.5
        pg.fixed_iters = ITERS
        pg.device = local_rank
        greg = capi.Registration(pg)
        d_tcov = torch.from_numpy(sc.tgt_cov).to(dev)
        d_scov = torch.from_numpy(sc.src_cov[lo:hi]).to(dev)
        torch.cuda.synchronize()
        greg.set_target_device(d_tgt.data_ptr(), 3, n_tgt, None, 3, d_tcov.data_ptr())
        greg.set_source_device(d_src.data_ptr(), 3, n_src, None, 3, d_scov.data_ptr())
        for _ in range(2):
            greg.register(T_init)
        torch.cuda.synchronize()
        tg0 = time.perf_counter()
        g_steps = max(3, args.steps // 4)
        for _ in range(g_steps):
            Tg, gres = greg.register(T_init)
        torch.cuda.synchronize()
        tg = time.perf_counter() - tg0
        gt, gr = synth.pose_error(Tg, sc.T_true)
        gicp = {"value": ITERS * g_steps / tg, "unit": "iter/s", "ms_per_registration": 1e3 * tg / g_steps,
                "registrations": g_steps, "bytes_per_point": 88,
                "achieved_GBs_end_to_end": n_src * 88 * ITERS * g_steps / tg / 1e9,
                "pose_vs_truth": {"trans_m": gt, "rot_rad": gr}, "T": Tg.tolist(),
                "note": "GICP parity is unpinned against the reference (Open3D 0.15.1 arithmetic not in tree); "
                        "checked against the float64 oracle in tests/test_gpu_parity.py"}
        greg.close()

    # secondary figure: the same workload with the shipped degeneracyAwareness (R8x: first-iteration localizability
    # analysis + constrained solve) switched on.  Never `value` (SURVEY 8d defines the measured chain without it).
    xicp = None
    if world == 1 and not force_dist and args.workload in ("c2", "tiny"):
        px = capi.shipped_params()
        px.fixed_iters = ITERS
        px.device = local_rank
        px.use_xicp = 1
        xreg = capi.Registration(px)
        xreg.set_target_device(d_tgt.data_ptr(), 3, n_tgt, d_tnrm.data_ptr(), 3)
        xreg.set_source_device(d_src.data_ptr(), 3, n_src, d_snrm.data_ptr(), 3)
        for _ in range(2):
            xreg.register(T_init)
        torch.cuda.synchronize()
        tx0 = time.perf_counter()
        x_steps = max(3, args.steps // 4)
        for _ in range(x_steps):
            Tx, xres = xreg.register(T_init)
        torch.cuda.synchronize()
        tx = time.perf_counter() - tx0
        xicp = {"value": ITERS * x_steps / tx, "unit": "iter/s", "ms_per_registration": 1e3 * tx / x_steps,
                "localizable": list(xres.localizable), "n_constraints": int(xres.n_constraints),
                "same_pose_as_plain_chain": bool(np.array_equal(Tx, T_final))}
        xreg.close()

    # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process; the number comes from
    # the committed rocprofv3 --pmc passes of THIS command (tools/collect_profiles.sh -> profiles/r01_pmc_traffic.json;
    # FETCH_SIZE doubled as the gfx950 guide prescribes and as the k_stream calibration in that file confirms).
    traffic, traffic_src = None, None
    try:
        pj = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
        if args.workload == "c2" and world == 1:
            for kname, kv in pj["kernels"].items():
                if kname.startswith(dom):
                    traffic = kv["hbm_bytes_per_launch_corrected"]
                    traffic_src = "profiles/r01_pmc_traffic.json (2*FETCH_SIZE + WRITE_SIZE, bytes per launch)"
    except (OSError, KeyError, ValueError):
        pass

    if rank == 0:
        value = world * ITERS * args.steps / elapsed
        line = {
            "metric": "ICP iterations/sec (scan-to-map, 100k-point reading units)",
            "value": value, "unit": "iter/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: scan-to-map point-to-plane ICP {n_src}x{world} -> {n_tgt} points, "
                                   f"{ITERS} iterations/registration, shipped icp.yaml chain (exact 1-NN, maxDist 0.5, "
                                   "Trimmed 0.9, SurfaceNormal 1.57)",
                       "n_source_per_gpu": n_src, "n_target": n_tgt, "iterations_per_step": ITERS,
                       "parallelism": (f"point-partitioned x{world}, {loop_kind}" if (world > 1 or force_dist) else "single GPU"),
                       "cell_size_m": info.cell_size, "n_bricks": info.n_bricks},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": n_src * bytes_pp,
                         "kernel_ms": kern[dom]["avg_ms"], "launches": kern[dom]["launches"],
                         "bytes_per_point": bytes_pp,
                         # measured, DESIGN.md section 6: duration = ~11 us fixed (one wave's chain of dependent round
                         # trips) + 13.5 us per 100 k reading points; VALU -14 % changed it by -1 %
                         "limiter": "latency of dependent round trips, then Infinity-Cache/HBM fetches of the candidate "
                                    "records an exact search has to look at (profiles/r01_search_kernel_counters.txt, "
                                    "profiles/r01_search_kernels_vs_reading_size.txt)"},
            "kernels": kern,
            "roofline_iteration": {"bytes_per_point": ITER_BYTES_PER_POINT,
                                   "achieved_GBs_end_to_end": n_src * ITER_BYTES_PER_POINT * ITERS * args.steps / elapsed / 1e9},
            "target_build_ms": float(reg.last_result.target_build_ms) if world == 1 else None,
            "band_stalls_last_step": int(reg.last_result.n_band_stalls) if world == 1 else None,
            "batched": batched,
            "gicp": gicp,
            "xicp": xicp,
        }
        if not args.no_cpu_baseline:
            cores = os.cpu_count() or 1
            threads = min(cores, 64)
            cb = cpu_baseline(sc, n_src, threads)
            line["cpu_baseline"] = {"value": cb["omp"], "unit": "iter/s", "cores": threads, "kind": "port",
                                    "sample": f"same {n_src}->{n_tgt} clouds and chain, {ITERS} iterations once, "
                                              "OpenMP oracle (kd-tree build excluded)",
                                    "value_1thread": cb["1t"]}
            line["speedup_vs_cpu_omp"] = value / world / cb["omp"]
            # final-pose parity against the oracle on the same inputs (whole reading)
            from oracle import oracle as orc
            To, _ = orc.icp_p2pl(sc.tgt_xyz, sc.tgt_nrm, sc.src_xyz, sc.src_nrm, max_dist=0.5, trim_ratio=0.9,
                                 max_normal_angle=1.57, fixed_iters=ITERS, n_threads=threads)
            dt, dr = synth.pose_error(T_final, To)
            et, er = synth.pose_error(T_final, sc.T_true)
            line["pose_vs_oracle"] = {"trans_m": dt, "rot_rad": dr}
            line["pose_vs_truth"] = {"trans_m": et, "rot_rad": er}
            if gicp is not None:
                Tgo, gor = orc.icp_gicp(sc.tgt_xyz, sc.tgt_cov, sc.src_xyz, sc.src_cov, max_dist=0.5, fixed_iters=ITERS,
                                        n_threads=threads)
                gicp["cpu_oracle_iter_per_s"] = ITERS / gor.loop_seconds   # kd-tree build excluded
                gdt, gdr = synth.pose_error(np.asarray(gicp["T"], np.float32), Tgo)
                gicp["pose_vs_oracle"] = {"trans_m": gdt, "rot_rad": gdr}
        if gicp is not None:
            gicp.pop("T", None)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
