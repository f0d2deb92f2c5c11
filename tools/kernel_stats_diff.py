"""Per-kernel ms/step of two rocprofv3 --kernel-trace --stats csv files side by side (kernels that moved by > 0.05 ms/step).
usage: python tools/kernel_stats_diff.py <a_kernel_stats.csv> <b_kernel_stats.csv> <steps profiled (incl. warm-up)>"""
import csv
import sys


def load(p):
    d = {}
    for r in csv.DictReader(open(p)):
        n = r["Name"].replace("void ", "").replace("nvq::", "").replace("(anonymous namespace)::", "")
        n = n.split("(")[0][:52]
        c, avg, tot = d.get(n, (0, 0.0, 0.0))
        d[n] = (c + int(r["Calls"]), float(r["AverageNs"]) / 1e3, tot + float(r["TotalDurationNs"]) / 1e6)
    return d


a, b, steps = load(sys.argv[1]), load(sys.argv[2]), float(sys.argv[3])
print("total ms/step", sum(v[2] for v in a.values()) / steps, sum(v[2] for v in b.values()) / steps)
for n in sorted(set(a) | set(b), key=lambda n: -(b.get(n, (0, 0, 0))[2] + a.get(n, (0, 0, 0))[2])):
    va, vb = a.get(n, (0, 0, 0)), b.get(n, (0, 0, 0))
    if abs(va[2] - vb[2]) / steps > 0.05:
        print(f"{n:54s} {va[0]:4d} {va[1]:8.1f}us {va[2] / steps:7.2f} | {vb[0]:4d} {vb[1]:8.1f}us {vb[2] / steps:7.2f}")
