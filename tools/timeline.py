"""Kernel timeline of a rocprofv3 --kernel-trace run (csv): start / end of every dispatch relative to the first, in ms.
usage: timeline.py <dir with *kernel_trace.csv> [last N rows]"""
import csv, glob, os, sys
files = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
rows = list(csv.DictReader(open(files[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows)
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    name = r["Kernel_Name"]
    name = name[:name.index("(")] if "(" in name else name
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    print(f"{s:10.3f} {e:10.3f} {e - s:9.3f} ms  q{r.get('Queue_Id', '?')} grid {r.get('Grid_Size', '?'):>8s} {name[:70]}")
