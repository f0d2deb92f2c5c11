"""Print a rocprofv3 *_kernel_stats.csv as ms per step.  usage: kstats.py <csv> <steps_in_trace>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total GPU kernel time: {tot / 1e6 / steps:.2f} ms/step")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print("%6.2f%% %8.3f ms/step %5.0f calls/step avg %9.1f us  %s" % (
        float(r["Percentage"]), float(r["TotalDurationNs"]) / 1e6 / steps, int(r["Calls"]) / steps,
        float(r["AverageNs"]) / 1e3, r["Name"][:80]))
