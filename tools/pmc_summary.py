"""Join rocprofv3 --pmc CSVs (counter_collection + kernel_trace) into per-(kernel, grid) averages.
usage: python tools/pmc_summary.py <dir-with-pmc_*.csv> [min_grid_blocks]"""
import csv, sys, collections, re
d = sys.argv[1]
rows = list(csv.DictReader(open(f"{d}/pmc_counter_collection.csv")))
trace = {r["Dispatch_Id"]: r for r in csv.DictReader(open(f"{d}/pmc_kernel_trace.csv"))}
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
seen = set()
for r in rows:
    k = (re.sub(r"\(.*", "", r["Kernel_Name"])[:70], int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1))
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if (r["Dispatch_Id"], ) not in seen:
        seen.add((r["Dispatch_Id"], ))
        cnt[k] += 1
        t = trace.get(r["Dispatch_Id"])
        if t:
            agg[k]["_us"] += (int(t["End_Timestamp"]) - int(t["Start_Timestamp"])) / 1e3
names = sorted({r["Counter_Name"] for r in rows})
print("kernel,blocks,calls,avg_us," + ",".join(names))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["_us"])[:40]:
    n = cnt[k]
    print(f"{k[0]},{k[1]},{n},{v['_us'] / n:.1f}," + ",".join(f"{v[c] / n:.4g}" for c in names))
