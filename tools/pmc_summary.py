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
        # durations of the same dispatches, when the pass ran with --kernel-trace
        dur = {}
        for t in glob.glob(os.path.join(os.path.dirname(f), "*kernel_trace.csv")):
            for tr in csv.DictReader(open(t)):
                try:
                    dur[tr["Dispatch_Id"]] = float(tr["End_Timestamp"]) - float(tr["Start_Timestamp"])
                except (KeyError, ValueError):
                    pass
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name") or row.get("kernel_name") or ""
            c = row.get("Counter_Name") or row.get("counter_name") or ""
            v = float(row.get("Counter_Value") or row.get("counter_value") or 0.0)
            if want and want not in k:
                continue
            a = acc[(c, k)]
            a[0] += 1
            a[1] += v
            if c == "GRBM_GUI_ACTIVE":
                # the clock the kernel held under this pass: active cycles (summed over the 8 XCDs) per ns
                t0, t1 = row.get("Start_Timestamp"), row.get("End_Timestamp")
                ns = float(t1) - float(t0) if t0 and t1 else dur.get(row.get("Dispatch_Id"), 0.0)
                if ns > 0.0:
                    g = acc[("SHADER_CLOCK_GHZ(GRBM_GUI_ACTIVE/8/duration)", k)]
                    g[0] += 1
                    g[1] += v / 8.0 / ns
w = csv.writer(sys.stdout)
w.writerow(["counter", "kernel", "dispatches", "avg_counter_value"])
for (c, k), (n, s) in sorted(acc.items()):
    w.writerow([c, k, n, "%.6g" % (s / n)])
