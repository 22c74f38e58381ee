"""Summarise rocprofv3 output of a bench.py run (kernel-trace stats + PMC passes) into a
small table. Usage: python profiles/summarize.py <stats_dir> <pmc_fetch_dir> <pmc_write_dir> [traffic.json [libspecdec_hip.so [workload]]] > profiles/<name>.md
FETCH_SIZE is in KiB and, on gfx950, counts exactly half of a 16-B/lane coalesced stream
(MI355X_MICROARCH.md §HBM): the corrected figure doubles it. WRITE_SIZE is exact."""

import collections
import csv
import glob
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void sd::", "").replace("sd::", "")
    for a, b in (("(GemvArgs)", ""), ("(AttnArgs)", ""), ("(EmbedArgs)", ""), ("(PersistArgs)", "")):
        name = name.replace(a, b)
    return name[:60]


def load_trace(d):
    f = (glob.glob(d + "/*/*kernel_trace.csv") + glob.glob(d + "/*kernel_trace.csv"))[0]
    return [r for r in csv.DictReader(open(f)) if "sd::" in r["Kernel_Name"]]


def load_pmc(d):
    f = (glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"))[0]
    return [r for r in csv.DictReader(open(f)) if "sd::" in r["Kernel_Name"]]


def main():
    stats_dir, fetch_dir, write_dir = sys.argv[1:4]
    rows = load_trace(stats_dir)
    agg = collections.OrderedDict()
    for r in rows:
        key = (short(r["Kernel_Name"]), r["LDS_Block_Size"])
        agg.setdefault(key, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    pmc = {}
    for d, cname in ((fetch_dir, "FETCH_SIZE"), (write_dir, "WRITE_SIZE")):
        for r in load_pmc(d):
            if r["Counter_Name"] != cname:
                continue
            key = (short(r["Kernel_Name"]), r["LDS_Block_Size"])
            pmc.setdefault((key, cname), []).append(float(r["Counter_Value"]))
    total = sum(sum(v) for v in agg.values())
    if len(sys.argv) > 4:   # machine-readable per-launch HBM traffic (bench.py's roofline.traffic)
        import json

        tr = {}
        for key in agg:
            f, w = pmc.get((key, "FETCH_SIZE")), pmc.get((key, "WRITE_SIZE"))
            if f and w:
                rd, wr = sum(f) / len(f) * 1024 * 2, sum(w) / len(w) * 1024
                tr[key[0].split("(")[0]] = {"hbm_bytes_per_launch": rd + wr, "read_bytes_x2_corrected": rd, "write_bytes": wr,
                                            "launches_counted": len(f),
                                            # average duration of the same kernel in the UN-profiled kernel trace of the bench's graph (us)
                                            "avg_us_in_graph": sum(agg[key]) / len(agg[key]) / 1e3, "launches_traced": len(agg[key])}
        if len(sys.argv) > 5:   # stamp: content hash of the library the counters were taken with + the workload
            import hashlib

            tr["_meta"] = {"lib_sha256": hashlib.sha256(open(sys.argv[5], "rb").read()).hexdigest(),
                           "workload": sys.argv[6] if len(sys.argv) > 6 else None}
        with open(sys.argv[4], "w") as fo:
            json.dump(tr, fo, indent=1)
    print("| kernel | LDS B | launches | avg us | total ms | % | FETCH KiB/launch (raw) | HBM read MB/launch (x2 corrected) | WRITE KiB/launch |")
    print("|---|---|---|---|---|---|---|---|---|")
    for key, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        f = pmc.get((key, "FETCH_SIZE"))
        w = pmc.get((key, "WRITE_SIZE"))
        fa = sum(f) / len(f) if f else None
        wa = sum(w) / len(w) if w else None
        print("| %s | %s | %d | %.2f | %.3f | %.1f | %s | %s | %s |" % (
            key[0], key[1], len(v), sum(v) / len(v) / 1e3, sum(v) / 1e6, 100.0 * sum(v) / total,
            "%.0f" % fa if fa is not None else "-", "%.2f" % (fa * 1024 * 2 / 1e6) if fa is not None else "-",
            "%.0f" % wa if wa is not None else "-"))


if __name__ == "__main__":
    main()
