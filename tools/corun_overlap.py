#!/usr/bin/env python3
"""rocprofv3 kernel trace of `kbench corun`: how much of the time two different kernels were in flight together, and whether their
intervals interleave at workgroup granularity (long overlapping intervals) or alternate (serialised dispatch).
usage: corun_overlap.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    n = r["Kernel_Name"]
    kind = "attn" if "attn" in n else ("gemm" if "gemm" in n else None)
    if kind:
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind, n[:40]))
ev.sort()
print(f"{len(ev)} gemm/attention dispatches")
# sweep: time with 0 / only gemm / only attn / both in flight
pts = []
for s, e, k, _ in ev:
    pts.append((s, 1, k)); pts.append((e, -1, k))
pts.sort()
cnt = collections.Counter(); acc = collections.Counter(); last = None
for t, d, k in pts:
    if last is not None:
        key = ("gemm" if cnt["gemm"] else "") + ("+attn" if cnt["attn"] else "")
        acc[key or "idle"] += t - last
    cnt[k] += d; last = t
tot = sum(acc.values())
for k, v in acc.items():
    print(f"  in flight {k:10s} {v / 1e6:9.3f} ms  {100.0 * v / tot:5.1f} %")
# per-kind mean duration when alone in flight over its whole interval vs overlapped with the other kind for > 50 % of its interval
def overlap(a, kind):
    s, e = a[0], a[1]
    o = 0
    for b in ev:
        if b[2] != kind:
            lo, hi = max(s, b[0]), min(e, b[1])
            if hi > lo:
                o += hi - lo
    return o / max(e - s, 1)
stat = collections.defaultdict(list)
for a in ev:
    f = overlap(a, a[2])
    stat[(a[2], "overlapped" if f > 0.5 else "alone")].append((a[1] - a[0]) / 1e3)
for k, v in sorted(stat.items()):
    print(f"  {k[0]:5s} {k[1]:10s}: {len(v):4d} dispatches, mean duration {sum(v) / len(v):9.1f} us")
