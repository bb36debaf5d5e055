"""Per-dispatch counter values (tools/collect_counters_r3.sh): the search kernel's launches of the LAST registration in
dispatch order (iteration 0, 1, 2, ...), the persistent tail kernel's launches, and the kernel durations from the trace."""
import csv, glob, os, sys, collections
root = sys.argv[1]
want = sys.argv[2:] or ["k_match_g8", "k_tail"]
tab = collections.defaultdict(dict)   # (kernel, dispatch order index) -> counter -> value
def newest(pattern):
    # gpurun merges every run's files next to the older ones: one file per pass directory, the most recent
    by_dir = {}
    for f in glob.glob(pattern, recursive=True):
        d = os.path.relpath(f, root).split(os.sep)[0]
        if d not in by_dir or os.path.getmtime(f) > os.path.getmtime(by_dir[d]):
            by_dir[d] = f
    return sorted(by_dir.values())


for f in newest(os.path.join(root, "**", "*counter_collection.csv")):
    per = collections.defaultdict(float)
    names = {}
    for row in csv.DictReader(open(f)):
        d = int(row["Dispatch_Id"])
        per[(d, row["Counter_Name"])] += float(row["Counter_Value"])
        names[d] = row["Kernel_Name"]
    order = collections.defaultdict(list)
    for d in sorted(names):
        for w in want:
            if w in names[d]:
                order[w].append(d)
    for w, ds in order.items():
        for k, d in enumerate(ds):
            for (dd, c), v in per.items():
                if dd == d:
                    tab[(w, k)][c] = v
dur = collections.defaultdict(list)
for f in newest(os.path.join(root, "trace", "**", "*kernel_trace.csv")):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    for r in rows:
        for w in want:
            if w in r["Kernel_Name"]:
                dur[w].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for w in want:
    n = max([k for (ww, k) in tab if ww == w], default=-1) + 1
    if n == 0:
        continue
    per_reg = n // 3 if n % 3 == 0 else n
    print(f"== {w}: {n} dispatches ({per_reg} per registration); the last registration, in dispatch order")
    cs = sorted({c for (ww, k), v in tab.items() if ww == w for c in v})
    print("   " + " ".join(f"{c[-18:]:>18s}" for c in ["us(trace)"] + cs))
    for k in range(n - per_reg, n):
        d = dur[w][k] if k < len(dur[w]) else float("nan")
        print("   " + f"{d:18.1f} " + " ".join(f"{tab[(w, k)].get(c, float('nan')):18.0f}" for c in cs))
