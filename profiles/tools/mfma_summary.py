"""Per-kernel MFMA counters from the rocprofv3 --pmc passes of profiles/tools/mfma_counters.sh:
   python profiles/tools/mfma_summary.py gpurun_out/<tag>  > profiles/roundN_mfma_counters.md
Counters are summed over the chip per dispatch; duration = End - Start of the same dispatch (under the counter pass the kernels run
serialised: durations are a few % above the unprofiled trace)."""
import csv
import glob
import re
import sys
from collections import defaultdict

tag = sys.argv[1]


def short(name: str) -> str:
    m = re.search(r"(persist_forward_kernel|gemv_mfma_kernel|attention_mfma_kernel|gemm_\w+_kernel)<([^>]*)>", name)
    if m:
        return f"{m.group(1)}<{m.group(2)}>"
    m = re.search(r"sd::(?:\(anonymous namespace\)::)?(\w+)", name)
    return m.group(1) if m else None


def load(sub):
    f = (glob.glob(f"{tag}_{sub}/*/*counter_collection.csv") + glob.glob(f"{tag}_{sub}/*counter_collection.csv"))[0]
    per = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    seen = set()
    with open(f, newline="") as fh:
        for r in csv.DictReader(fh):
            k = short(r["Kernel_Name"])
            if k is None:
                continue
            per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return per, dur


busy, dur = load("busy")
mops, _ = load("mops")
rows = []
for k in busy:
    n = len(dur[k])
    us = sum(dur[k]) / n
    b = sum(busy[k]["SQ_VALU_MFMA_BUSY_CYCLES"]) / max(len(busy[k]["SQ_VALU_MFMA_BUSY_CYCLES"]), 1)
    insts = sum(mops.get(k, {}).get("SQ_INSTS_MFMA", [0])) / max(len(mops.get(k, {}).get("SQ_INSTS_MFMA", [0])), 1)
    mo = sum(mops.get(k, {}).get("SQ_INSTS_VALU_MFMA_MOPS_BF16", [0])) / max(len(mops.get(k, {}).get("SQ_INSTS_VALU_MFMA_MOPS_BF16", [0])), 1)
    util = b / (us * 1e-6 * 2.4e9 * 1024) * 100.0     # busy cycles / (duration x 2.4 GHz x 1024 SIMDs)
    flop = mo * 512.0
    rows.append((n * us, k, n, us, insts, b, util, flop, flop / (us * 1e-6) / 1e12 if us else 0.0))
rows.sort(reverse=True)
print("| kernel | launches | avg us (counter pass) | SQ_INSTS_MFMA / launch | SQ_VALU_MFMA_BUSY_CYCLES / launch | MFMA utilisation % | bf16 MFMA flop / launch | TFLOP/s |")
print("|---|---|---|---|---|---|---|---|")
for _, k, n, us, insts, b, util, flop, tf in rows:
    print(f"| {k} | {n} | {us:.2f} | {insts:.0f} | {b:.0f} | {util:.2f} | {flop:.3g} | {tf:.1f} |")
