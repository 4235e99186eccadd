"""Per-dispatch listing of rocprofv3 --pmc output for kernels whose name contains a substring.
usage: python tools/pmc_list.py <dir> <substr> [every_n]"""
import csv, sys, collections
d, sub = sys.argv[1], sys.argv[2]
every = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rows = list(csv.DictReader(open(f"{d}/pmc_counter_collection.csv")))
trace = {r["Dispatch_Id"]: r for r in csv.DictReader(open(f"{d}/pmc_kernel_trace.csv"))}
disp = collections.OrderedDict()
for r in rows:
    if sub not in r["Kernel_Name"]:
        continue
    disp.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = disp.get(int(r["Dispatch_Id"]), {}).get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
names = sorted({k for v in disp.values() for k in v})
print("dispatch,us," + ",".join(names))
for i, (k, v) in enumerate(sorted(disp.items())):
    if i % every:
        continue
    t = trace.get(str(k))
    us = (int(t["End_Timestamp"]) - int(t["Start_Timestamp"])) / 1e3 if t else -1
    print(f"{k},{us:.1f}," + ",".join(f"{v.get(c, 0):.4g}" for c in names))
