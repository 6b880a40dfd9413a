#!/usr/bin/env python3
"""Prints the top rows of a rocprofv3 kernel_stats.csv with the kernel names cut short.  Usage: kstats.py DIR [N]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
for r in list(csv.DictReader(open(f)))[:n]:
    print(f'{r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]:60s} calls {r["Calls"]:>7s} avg {float(r["AverageNs"])/1e3:9.2f} us  {r["Percentage"]}%')
