#!/usr/bin/env python3
"""Print a rocprofv3 kernel_stats.csv as us per bench step: tools/kstats.py <csv> [steps incl. warm-up] [rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 33
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("kernel time per step: %.3f ms" % (tot / steps / 1e6))
for r in rows[:top]:
    print(f"{r['Name'][:72]:72s} {int(r['Calls']):5d} {float(r['TotalDurationNs'])/steps/1e3:8.1f} us/step  avg {float(r['AverageNs'])/1e3:7.1f} us {float(r['Percentage']):5.1f}%")
