#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc passes: walks the given directories for
*counter_collection.csv and prints counter,kernel,dispatches,avg_value (CSV).
usage: pmc_summary.py <dir> [<dir> ...] [--kernel substring]"""
import csv
import glob
import os
import sys
from collections import defaultdict

dirs = [a for a in sys.argv[1:] if not a.startswith("--")]
want = None
if "--kernel" in sys.argv:
    want = sys.argv[sys.argv.index("--kernel") + 1]
    dirs = [d for d in dirs if d != want]
acc = defaultdict(lambda: [0, 0.0])
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name") or row.get("kernel_name") or ""
            c = row.get("Counter_Name") or row.get("counter_name") or ""
            v = float(row.get("Counter_Value") or row.get("counter_value") or 0.0)
            if want and want not in k:
                continue
            a = acc[(c, k)]
            a[0] += 1
            a[1] += v
w = csv.writer(sys.stdout)
w.writerow(["counter", "kernel", "dispatches", "avg_counter_value"])
for (c, k), (n, s) in sorted(acc.items()):
    w.writerow([c, k, n, "%.6g" % (s / n)])
