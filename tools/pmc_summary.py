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
    # (template arguments contain ", ": written with ";" so that the summary stays a plain comma-separated table)
    k = (re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")).replace(", ", ";")[:70], int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1))
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if (r["Dispatch_Id"], ) not in seen:
        seen.add((r["Dispatch_Id"], ))
        cnt[k] += 1
        t = trace.get(r["Dispatch_Id"])
        if t:
            agg[k]["_us"] += (int(t["End_Timestamp"]) - int(t["Start_Timestamp"])) / 1e3
names = sorted({r["Counter_Name"] for r in rows})
derive = "SQ_VALU_MFMA_BUSY_CYCLES" in names and "GRBM_GUI_ACTIVE" in names
# MFMA utilisation at the clock the kernel actually ran at: busy cycles / (cycles per XCD-summed GUI_ACTIVE / 8 XCDs * 1024 SIMDs);
# effective clock = GUI_ACTIVE / 8 / duration
print("kernel,blocks,calls,avg_us," + ",".join(names) + (",mfma_util,clock_ghz" if derive else ""))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["_us"])[:40]:
    n = cnt[k]
    extra = ""
    if derive and v["GRBM_GUI_ACTIVE"] > 0:
        cyc = v["GRBM_GUI_ACTIVE"] / 8.0
        extra = f",{v['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):.3f},{cyc / (v['_us'] * 1e3):.2f}"
    print(f"{k[0]},{k[1]},{n},{v['_us'] / n:.1f}," + ",".join(f"{v[c] / n:.4g}" for c in names) + extra)
