"""Aggregate a rocprofv3 --kernel-trace CSV by (kernel, grid, workgroup): calls, total/avg time.
usage: python tools/trace_summary.py <kernel_trace.csv> [top_n]"""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", ""))
    name = re.sub(r"^void ", "", name)[:60]
    def dims(key):                                   # rocprofv3 7.x: per-dimension columns; older: one total
        if key + "_X" in r:
            return int(r[key + "_X"] or 1) * int(r.get(key + "_Y") or 1) * int(r.get(key + "_Z") or 1)
        return int(r.get(key) or 1)
    gs, wg = dims("Grid_Size"), dims("Workgroup_Size")
    k = (name, gs // max(wg, 1), wg)
    agg[k][0] += 1
    agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in agg.values())
print(f"total kernel time {tot/1e3:.2f} ms over {len(rows)} dispatches")
print("kernel,blocks,wg,calls,total_ms,avg_us,pct")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{k[0]},{k[1]},{k[2]},{v[0]},{v[1]/1e3:.3f},{v[1]/v[0]:.1f},{100*v[1]/tot:.1f}")
