#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace CSV: per-kernel stats and the timeline of one late iteration."""
import csv, sys, glob, collections
d = sys.argv[1]
f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
stats = collections.defaultdict(list)
for r in rows:
    stats[r['Kernel_Name'].split('(')[0][:50]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000)
for k, v in sorted(stats.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print(f"{k:52s} n={len(v):5d} tot={sum(v)/1000:9.3f}ms avg={sum(v)/len(v):9.2f}us med={v2[len(v)//2]:9.2f} min={v2[0]:8.2f} max={v2[-1]:9.2f}")
m = [i for i, r in enumerate(rows) if 'k_match' in r['Kernel_Name']]
if len(m) > 30:
    i0 = m[-25]
    t0 = int(rows[i0]['Start_Timestamp'])
    print("--- timeline (us) from a late match kernel")
    for r in rows[i0:i0 + 14]:
        s = int(r['Start_Timestamp'])
        print(f"{(s-t0)/1000:9.1f} {(int(r['End_Timestamp'])-s)/1000:8.1f}  {r['Kernel_Name'][:48]}  grid={r['Grid_Size_X']} vgpr={r['VGPR_Count']} lds={r['LDS_Block_Size']}")
    print("match durations of last registration:", [round((int(rows[i]['End_Timestamp'])-int(rows[i]['Start_Timestamp']))/1000,1) for i in m[-20:]])
