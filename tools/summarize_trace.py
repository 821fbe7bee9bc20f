#!/usr/bin/env python3
"""Aggregate a rocprofv3 kernel_trace CSV by (kernel name, grid, workgroup): usage summarize_trace.py <dir-or-csv> [skip_first_n_dispatches | -last_n]
Prints calls, total and average duration per group, sorted by total time, and the sum of the gaps between consecutive
dispatches (end -> next start)."""
import csv, glob, os, sys

path = sys.argv[1]
if os.path.isdir(path):
    cands = glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
    path = cands[0]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows = rows[skip:]      # (negative: the last -skip dispatches only, e.g. the steps behind the model build)
agg = {}
for r in rows:
    name = r["Kernel_Name"]
    name = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:70]
    key = (name, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")))
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    e = agg.setdefault(key, [0, 0])
    e[0] += 1; e[1] += d
tot = sum(v[1] for v in agg.values())
gaps = sum(max(0, int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) for a, b in zip(rows, rows[1:]))
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print(f"dispatches {len(rows)}  kernel time {tot/1e3:.1f} us  gaps {gaps/1e3:.1f} us  span {span/1e3:.1f} us")
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{t/1e3:10.1f} us {n:6d} x {t/n/1e3:8.2f} us  {100*t/tot:5.1f}%  grid {k[1]},{k[2]},{k[3]} wg {k[4]}  {k[0]}")
