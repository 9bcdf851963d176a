"""Prints, from a rocprofv3 --kernel-trace CSV, the last pool-kernel launches with start/end relative to the previous one."""
import csv
import glob
import sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "poolkernel" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
prev = None
for s, e, q, st in rows[-16:]:
    print(f"queue {q} stream {st}: dur {(e - s) / 1e3:8.1f} us; starts {((s - prev[1]) / 1e3 if prev else 0):8.1f} us after the previous kernel ENDED")
    prev = (s, e)
