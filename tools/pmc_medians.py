"""per-kernel median of every counter found in gpurun_out/<prefix>*/.../*counter_collection.csv"""
import csv, glob, os, statistics, sys
root, prefix = sys.argv[1], sys.argv[2]
acc = {}
for f in glob.glob(os.path.join(root, prefix + "*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f, newline="")):
        acc.setdefault(row["Kernel_Name"], {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
for k in sorted(acc):
    if "dr3" not in k:
        continue
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("   %-32s n=%-4d median %.6g" % (c, len(v), statistics.median(v)))
