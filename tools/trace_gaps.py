"""Prints kernel durations and the idle gaps between consecutive kernels from a rocprofv3 --kernel-trace CSV."""
import csv
import glob
import sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]))
rows.sort()
rows = rows[len(rows) // 2:][:24]
prev_end = None
for s, e, n in rows:
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"{n:40s} dur {(e - s) / 1e3:9.1f} us   gap before {gap:9.1f} us")
    prev_end = e
