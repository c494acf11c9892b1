#!/usr/bin/env python3
"""Print name (short), calls, avg us of a rocprofv3 kernel_stats.csv, optionally filtered."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for r in rows:
    name = r["Name"].split("(")[0].replace("void ", "")
    if pat in name:
        print(f"{name:45s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs'])/1e3:9.2f} us  total {float(r['TotalDurationNs'])/1e6:8.3f} ms")
