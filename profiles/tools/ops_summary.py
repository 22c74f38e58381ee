"""Per-(kernel, grid) launch durations of the registry-op kernels from a rocprofv3 --kernel-trace CSV.
python profiles/tools/ops_summary.py <trace dir>"""
import collections
import csv
import glob
import sys

f = (glob.glob(sys.argv[1] + "/*/*kernel_trace.csv") + glob.glob(sys.argv[1] + "/*kernel_trace.csv"))[0]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if not any(k in n for k in ("verify_", "kv_")):
        continue
    n = n.replace("void sd::", "").replace("sd::", "").split("(")[0]
    key = (n, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Y", ""), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")))
    agg.setdefault(key, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("| kernel | grid x (threads) | grid y | workgroup | launches | avg us | min us |")
print("|---|---|---|---|---|---|---|")
for (n, gx, gy, wg), v in sorted(agg.items()):
    print(f"| {n} | {gx} | {gy} | {wg} | {len(v)} | {sum(v) / len(v) / 1e3:.2f} | {min(v) / 1e3:.2f} |")
