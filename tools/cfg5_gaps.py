"""Idle gaps of the GPU in the cfg5 step from a rocprofv3 --kernel-trace csv: for the last full steps, the kernels after which the
device sat idle for more than 5 us and for how long.  usage: python tools/cfg5_gaps.py <p_kernel_trace.csv>"""
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
rows = rows[-3000:]                                   # the steady-state tail of the run
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
print(f"{len(rows)} launches over {span / 1e6:.2f} ms: busy {busy / 1e6:.2f} ms ({busy / span:.0%})")
gaps = {}
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    g = s1 - e0
    if g > 5000:
        k = (n0.split("(")[0][-40:], n1.split("(")[0][-40:])
        c, t = gaps.get(k, (0, 0))
        gaps[k] = (c + 1, t + g)
tot = sum(t for _, t in gaps.values())
print(f"gaps > 5 us: {tot / 1e6:.2f} ms in all")
for k, (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"{t / c / 1e3:8.1f} us x {c:4d}   after {k[0]}  ->  {k[1]}")
