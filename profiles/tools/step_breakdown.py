"""Per-step busy/gap breakdown from a rocprofv3 kernel trace of bench.py.
Usage: python profiles/tools/step_breakdown.py <kernel_trace.csv | results.db> > profiles/<name>.md
A step is delimited by consecutive verify_tail_kernel (or, sampling steps, accept_kernel) launches; the table shows, per kernel kind,
launches per step, the average duration and the average idle gap before the launch (end of the
previous kernel on the device -> start of this one), taken over the steady-state steps."""

import collections
import csv
import sys


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void sd::", "").replace("sd::", "")
    for a in ("(sd::GemvArgs)", "(sd::AttnArgs)", "(sd::EmbedArgs)", "(sd::PersistArgs)", "(PersistArgs)"):
        n = n.replace(a, "")
    return n.split("(")[0][:64]


def main():
    if sys.argv[1].endswith(".db"):      # rocprofv3's default rocpd output
        import sqlite3

        q = sqlite3.connect(sys.argv[1]).execute("select name, start, end from kernels where name like '%sd::%' order by start")
        ev = [(short(n), int(s), int(e)) for n, s, e in q]
    else:
        rows = [r for r in csv.DictReader(open(sys.argv[1])) if "sd::" in r["Kernel_Name"]]
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        ev = [(short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
    cuts = [i for i, e in enumerate(ev) if e[0].startswith(("verify_tail_kernel", "accept_kernel"))]
    steps = [(cuts[i] + 1, cuts[i + 1] + 1) for i in range(len(cuts) - 1)]
    steps = steps[len(steps) // 2:]          # steady state: second half
    per = collections.OrderedDict()
    wall = busy = gap = 0
    for a, b in steps:
        seg = ev[a:b]
        prev_end = ev[a - 1][2]
        wall += seg[-1][2] - prev_end
        for name, s, e in seg:
            d = per.setdefault(name, [0, 0, 0])
            d[0] += 1
            d[1] += e - s
            d[2] += max(0, s - prev_end)
            busy += e - s
            gap += max(0, s - prev_end)
            prev_end = max(prev_end, e)
    n = len(steps)
    print(f"steady-state steps: {n}; per step: wall {wall / n / 1e3:.1f} us, kernels busy {busy / n / 1e3:.1f} us, "
          f"idle between kernels {gap / n / 1e3:.1f} us\n")
    print("| kernel | launches/step | avg us | avg gap before us | busy us/step | gap us/step |")
    print("|---|---|---|---|---|---|")
    for name, (c, d, g) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print(f"| {name} | {c / n:.1f} | {d / c / 1e3:.2f} | {g / c / 1e3:.2f} | {d / n / 1e3:.1f} | {g / n / 1e3:.1f} |")


if __name__ == "__main__":
    main()
