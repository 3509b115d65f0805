#!/usr/bin/env python3
"""compact timeline of the LAST `count` kernel / copy records of a rocprofv3 --kernel-trace --memory-copy-trace csv directory"""
import csv
import glob
import sys
d, count = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        name = name[name.find("akp") if "akp" in name else 0:][:42]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Stream_Id", r.get("Queue_Id", "?"))))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")[:24], r.get("Stream_Id", "?")))
rows.sort()
rows = rows[-count:]
t0 = rows[0][0] if rows else 0
for a, b, k, s in rows:
    print("%9.3f ms  +%7.3f ms  %-44s stream %s" % ((a - t0) / 1e6, (b - a) / 1e6, k, s))
