#!/usr/bin/env python3
"""One batch-1 step of a rocprofv3 --kernel-trace CSV as a list: every launch of the median step in start order with its queue, duration and the
gap to the previous kernel's end on the chip (negative = overlapped), plus the step's totals.  usage: python tools/trace_b1_steps.py <kernel_trace.csv>"""
import csv, re, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", ""))[:56]
    gs = int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0); wg = int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", 1)) or 1)
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, gs // max(wg, 1), r.get("Queue_Id", "?")))
rows.sort()
# a step starts at preprocess_kernel
starts = [i for i, r in enumerate(rows) if "preprocess_kernel" in r[2]]
steps = [rows[a:b] for a, b in zip(starts[:-1], starts[1:])]
steps = [s for s in steps if 100 < len(s) < 600]
if not steps: sys.exit("no steps found")
walls = sorted((s[-1][1] - s[0][0], i) for i, s in enumerate(steps))
w, i = walls[len(walls) // 2]
s = steps[i]
print(f"{len(steps)} steps; median step: {len(s)} launches, first start -> last end {w / 1e3:.1f} us, sum of durations {sum(e - b for b, e, *_ in s) / 1e3:.1f} us")
cur = s[0][0]; idle = 0
for b, e, n, blk, q in s:
    gap = b - cur
    if gap > 0: idle += gap
    print(f"{(b - s[0][0]) / 1e3:9.1f} us  q{q:>3}  {n:56s} blocks {blk:6d}  {(e - b) / 1e3:7.1f} us  gap {gap / 1e3:7.1f}")
    cur = max(cur, e)
print(f"idle (no kernel on the chip) {idle / 1e3:.1f} us of {w / 1e3:.1f}")
