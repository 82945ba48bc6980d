#!/usr/bin/env python
"""print per-kernel averages from a rocprofv3 --stats output directory: python tools/kstats.py DIR [filter]"""
import csv, glob, sys
d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if flt in r["Name"] and float(r["Percentage"]) > 0.3:
            print("%-58s calls %5s avg %9.1f us  %5.1f%%" % (r["Name"][:58], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
