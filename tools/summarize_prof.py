"""Summarises rocprofv3 CSV output (kernel trace stats + PMC passes) of tools/profile_gpu.sh into one text file."""
import csv, glob, os, sys, collections
root = sys.argv[1]
out = []
for f in sorted(glob.glob(os.path.join(root, "kt", "**", "*kernel_stats.csv"), recursive=True)):
    out.append(f"# kernel stats ({os.path.relpath(f, root)})")
    out.append(open(f).read().strip())
for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            k = (row.get("Kernel_Name", "?")[:60], row.get("Counter_Name", "?"))
            acc[k][0] += float(row.get("Counter_Value", 0) or 0)
            acc[k][1] += 1
        out.append(f"# pmc {os.path.basename(d)} ({os.path.relpath(f, root)}): mean per dispatch")
        for (kn, cn), (v, n) in sorted(acc.items()):
            out.append(f"{kn:60s} {cn:32s} {v / n:20.1f}  (n={n})")
print("\n".join(out))
