"""last launches of a rocprofv3 kernel trace, as a timeline (usage: trace_tail.py <kernel_trace.csv> [n])"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = [r for r in rows if 'upload' not in r['Kernel_Name']]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
t0 = None
for r in rows[-n:]:
    k = r['Kernel_Name']; k = k[:k.index('(')] if '(' in k else k; k = k.split('::')[-1][:30]
    if t0 is None: t0 = int(r['Start_Timestamp'])
    print("%-32s start %8.1f dur %7.1f us grid %sx%sx%s" % (k, (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z']))
