"""Mean counter values per kernel from the passes of tools/collect_counters.sh."""
import csv, glob, os, sys, collections
root = sys.argv[1]
want = sys.argv[2:] or ["k_iter_fused", "k_reduce_update", "k_match_g8"]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    per_dispatch = collections.defaultdict(float)
    names = {}
    for row in csv.DictReader(open(f)):
        key = (f, row["Dispatch_Id"], row["Counter_Name"])
        per_dispatch[key] += float(row["Counter_Value"])
        names[(f, row["Dispatch_Id"])] = row["Kernel_Name"]
    for (ff, d, c), v in per_dispatch.items():
        acc[names[(ff, d)]][c].append(v)
for k, cs in sorted(acc.items()):
    if not any(w in k for w in want):
        continue
    print(k[:60])
    for c, v in sorted(cs.items()):
        print(f"   {c:40s} mean {sum(v) / len(v):16.1f}  n {len(v)}")
