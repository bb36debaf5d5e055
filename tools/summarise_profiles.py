#!/usr/bin/env python3
"""Turns the rocprofv3 CSVs collected by tools/collect_profiles.sh into the small tracked files under profiles/."""
import collections, csv, glob as _glob, json, os, shutil, sys


class glob:  # newest first: gpurun merges new runs next to older ones
    @staticmethod
    def glob(p):
        return sorted(_glob.glob(p), key=os.path.getmtime, reverse=True)


out = sys.argv[1]
TAG = sys.argv[2] if len(sys.argv) > 2 else "r03"          # round tag of the files written under profiles/
WL = sys.argv[3] if len(sys.argv) > 3 else "c3"            # workload the bench command ran (bench.py default: c3)
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
os.makedirs(dst, exist_ok=True)


def counters(d):
    fs = glob.glob(os.path.join(out, d, "*", "*counter_collection.csv"))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    if fs:
        for r in csv.DictReader(open(fs[0])):
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


stats = glob.glob(os.path.join(out, "trace", "*", "*kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], os.path.join(dst, f"{TAG}_bench_{WL}_kernel_stats.csv"))
trace = glob.glob(os.path.join(out, "trace", "*", "*kernel_trace.csv"))
dur = collections.defaultdict(list)
if trace:
    for r in csv.DictReader(open(trace[0])):
        dur[r["Kernel_Name"].split("(")[0].replace("void ", "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
fetch, write, calib = counters("pmc_fetch"), counters("pmc_write"), counters("calib_fetch")
# calibration: FETCH_SIZE is reported in KiB
cal = {}
for k, known in (("k_stream", float(1 << 30)), ("k_gather16", float((1 << 30) // 4))):
    v = calib.get(k, {}).get("FETCH_SIZE")
    if v:
        cal[k] = {"known_useful_bytes": known, "FETCH_SIZE_KiB_mean": sum(v) / len(v),
                  "reported_over_known": (sum(v) / len(v)) * 1024.0 / known}
summary = {"command": "python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras", "workload": WL, "calibration": cal, "kernels": {}}
stream_ratio = cal.get("k_stream", {}).get("reported_over_known", 0.5)
for k in sorted(set(fetch) | set(write)):
    if not any(x in k for x in ("k_match", "k_iter_fused", "k_coh_", "k_linearize", "k_reduce_update", "k_select", "k_hist", "k_tail")):
        continue
    f = fetch.get(k, {}).get("FETCH_SIZE", [])
    w = write.get(k, {}).get("WRITE_SIZE", [])
    d = dur.get(k, [])
    summary["kernels"][k] = {
        "launches": len(f),
        "avg_us": sum(d) / len(d) if d else None,
        "FETCH_SIZE_KiB_mean": sum(f) / len(f) if f else None,
        "WRITE_SIZE_KiB_mean": sum(w) / len(w) if w else None,
        # gfx950: FETCH_SIZE counts 128-byte requests as 64 B for wide (16 B/lane) reads -> divide by the measured ratio
        "hbm_bytes_per_launch_corrected": ((sum(f) / len(f)) * 1024.0 / stream_ratio if f else 0.0) + ((sum(w) / len(w)) * 1024.0 if w else 0.0),
    }
ks = summary["kernels"]
# the persistent tail kernel: one launch runs many iterations -> per-iteration figures from the bench line of the same command
tail_its = None
try:
    for line in open(os.path.join(out, "bench_trace.log")):
        if line.startswith('{"metric"'):
            tail_its = json.loads(line)["kernels"]["k_tail"]["iterations"]
except (OSError, KeyError, ValueError):
    pass
for k, v in ks.items():
    if k.startswith("k_tail") and tail_its:
        v["iterations_per_launch"] = tail_its
        v["avg_us_per_iteration"] = (v["avg_us"] or 0) / tail_its
        v["hbm_bytes_per_iteration_corrected"] = v["hbm_bytes_per_launch_corrected"] / tail_its
chk = next((v for k, v in ks.items() if k.startswith("k_coh_check")), None)
sea = next((v for k, v in ks.items() if k.startswith("k_coh_search")), None)
if chk and sea and chk["launches"]:
    ks["fused_pair"] = {"launches": chk["launches"], "avg_us": (chk["avg_us"] or 0) + (sea["avg_us"] or 0) * sea["launches"] / chk["launches"],
                        "hbm_bytes_per_launch_corrected": chk["hbm_bytes_per_launch_corrected"] +
                        sea["hbm_bytes_per_launch_corrected"] * sea["launches"] / chk["launches"],
                        "note": "k_coh_check + k_coh_search: the fused iteration's search + linearisation (two launches)"}
json.dump(summary, open(os.path.join(dst, f"{TAG}_pmc_traffic.json"), "w"), indent=1)
with open(os.path.join(dst, f"{TAG}_bench_{WL}_trace_summary.txt"), "w") as fh:
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        v2 = sorted(v)
        fh.write(f"{k[:60]:62s} n={len(v):5d} tot={sum(v)/1000:9.3f}ms avg={sum(v)/len(v):9.2f}us med={v2[len(v)//2]:9.2f} min={v2[0]:8.2f} max={v2[-1]:9.2f}\n")
for line in open(os.path.join(out, "bench_trace.log")):
    if line.startswith('{"metric"'):
        open(os.path.join(dst, f"{TAG}_bench_{WL}_line.json"), "w").write(line)
nst = glob.glob(os.path.join(out, "normals", "*", "*kernel_stats.csv"))
if nst:
    rows = list(csv.DictReader(open(nst[0])))
    with open(os.path.join(dst, f"{TAG}_normals_1M_k10_kernel_stats.csv"), "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=rows[0].keys())
        w.writeheader()
        for r in rows:
            r["Name"] = r["Name"][:100]
            w.writerow(r)
    with open(os.path.join(dst, f"{TAG}_normals_1M_k10_line.txt"), "w") as fh:
        fh.writelines(l for l in open(os.path.join(out, "normals.log")) if l.startswith(("GPU", "CPU")))
tp = os.path.join(out, "target_prep.log")
if os.path.exists(tp):
    with open(os.path.join(dst, f"{TAG}_target_prep_5M_line.txt"), "w") as fh:
        fh.writelines(l for l in open(tp) if l.startswith(("GPU", "host")))
bp = os.path.join(out, "bench_plain.json")
if os.path.exists(bp):
    for line in open(bp):
        if line.startswith('{"metric"'):
            open(os.path.join(dst, f"{TAG}_bench_{WL}_line_unprofiled.json"), "w").write(line)
tl = os.path.join(out, "timeline.txt")
if os.path.exists(tl):
    shutil.copy(tl, os.path.join(dst, f"{TAG}_registration_timeline.txt"))
sc = os.path.join(out, "scaling.txt")
if os.path.exists(sc):
    shutil.copy(sc, os.path.join(dst, f"{TAG}_search_kernels_vs_reading_size.txt"))
print(json.dumps(summary, indent=1))
