"""usage: kstats_show.py <kernel_stats.csv> [substring ...]: rows of a rocprofv3 kernel_stats.csv whose name has one of the substrings"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
keys = sys.argv[2:]
for r in rows:
    if not keys or any(k in r["Name"] for k in keys):
        print("%-78s %5s %9.1f us %6.2f%%" % (r["Name"].replace("(anonymous namespace)::", "")[:78], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
