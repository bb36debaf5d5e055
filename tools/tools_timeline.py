#!/usr/bin/env python3
"""Timeline (start offset, duration, gap to the previous kernel) of the last registration in a rocprofv3 kernel trace."""
import csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
prep = [i for i, r in enumerate(rows) if 'k_prepare_source' in r['Kernel_Name']]
i0 = prep[-1] - 1 if prep else max(0, len(rows) - 80)   # k_centroid_sums, k_prepare_source, state copy, iterations
t0 = int(rows[i0]['Start_Timestamp'])
prev_end = t0
busy = 0
for r in rows[i0:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"{(s-t0)/1000:9.1f} dur {(e-s)/1000:7.1f} gap {(s-prev_end)/1000:6.1f}  {r['Kernel_Name'].split('(')[0][:40]} grid={r['Grid_Size_X']}")
    busy += e - s
    prev_end = e
print(f"span {(prev_end-t0)/1000:.1f} us, busy {busy/1000:.1f} us")
