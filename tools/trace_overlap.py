#!/usr/bin/env python3
"""Timeline view of a rocprofv3 --kernel-trace CSV taken in the PRODUCTION mode (two sub-batch streams): how much of the wall time has 0 / 1 / 2+
kernels in flight, and the biggest idle gaps with the kernels either side of them.
usage: python tools/trace_overlap.py <kernel_trace.csv> [b1]"""
import csv
import re
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", ""))[:60]
    gs = int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)
    wg = int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", 1)) or 1)
    blocks = gs // max(wg, 1)
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, blocks, r.get("Queue_Id", "?")))
rows.sort()
# the window: batch-32 steps only = from the first to the last full-size launch of the qkv GEMM (persistent kernel: one block per CU;
# the batch-1 latency leg has fewer tiles than CUs), without the first quarter (warm-up step)
b1 = len(sys.argv) > 2 and sys.argv[2] == "b1"       # the batch-1 latency leg instead (its qkv GEMM has fewer tiles than CUs)
anchor = [r for r in rows if "gemm_pp128p_kernel<8" in r[2] and ((r[3] < 256) if b1 else (r[3] >= 256))] or [r for r in rows if "attn" in r[2]]
t_lo, t_hi = anchor[0][0], anchor[-1][1]
t_lo = t_lo + int(0.25 * (t_hi - t_lo))
win = [r for r in rows if r[0] >= t_lo and r[1] <= t_hi]
ev = []
for s, e, *_ in win:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
depth, last, hist = 0, ev[0][0], {}
for t, d in ev:
    hist[min(depth, 3)] = hist.get(min(depth, 3), 0) + (t - last)
    depth += d; last = t
wall = ev[-1][0] - ev[0][0]
print(f"window {wall / 1e6:.2f} ms, {len(win)} kernels, queues {sorted(set(r[4] for r in win))}")
for k in sorted(hist):
    print(f"  {k}{'+' if k == 3 else ''} kernels in flight: {hist[k] / 1e6:8.2f} ms  {100.0 * hist[k] / wall:5.1f} %")
busy_sum = sum(e - s for s, e, *_ in win)
print(f"  sum of kernel durations {busy_sum / 1e6:.2f} ms = {busy_sum / wall:.2f} x wall")
# idle gaps
gaps = []
cur_end, prev = win[0][1], win[0]
for r in win[1:]:
    if r[0] > cur_end:
        gaps.append((r[0] - cur_end, prev[2], r[2]))
    if r[1] > cur_end:
        cur_end, prev = r[1], r
gaps.sort(reverse=True)
print(f"idle gaps: {len(gaps)}, total {sum(g[0] for g in gaps) / 1e6:.2f} ms; largest:")
for g in gaps[:8]:
    print(f"  {g[0] / 1e3:8.1f} us   after {g[1]}   before {g[2]}")
# per-kernel time inflation is visible by comparing with the single-stream summary (tools/trace_summary.py)
agg = {}
for s, e, n, b, q in win:
    a = agg.setdefault((n, b), [0, 0]); a[0] += 1; a[1] += e - s
print("top kernels in the window (concurrent execution stretches them):")
for (n, b), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:10]:
    print(f"  {n:60s} blocks {b:6d} calls {c:4d} total {t / 1e6:8.2f} ms avg {t / c / 1e3:8.1f} us")
