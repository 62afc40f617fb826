# per-launch tracker durations over a long timed region (is the start of the timed region slower than its steady state?)
set -e
R=$GRAFT_REPO_ROOT
( cd $R && python3 -c "import __graft_entry__ as g; g.build()" > /dev/null )
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_decay -o d -- python3 $R/bench.py --steps 200 --warmup 2 --no-cpu --no-frontend --no-host-api --no-warp > $R/gpurun_out/prof_decay.log 2>&1
python3 - <<'PY'
import csv, os, glob
R = os.environ["GRAFT_REPO_ROOT"]
f = glob.glob(R + "/gpurun_out/prof_decay/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "lk_trackg_kernel<15" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
t = [int(r["Start_Timestamp"]) / 1e6 for r in rows]
print(len(d), "tracker launches")
for i in range(0, len(d), 8):
    print("%4d  t=%9.2f ms  %s" % (i, t[i] - t[0], " ".join("%6.1f" % x for x in d[i:i + 8])))
PY
