# rocprofv3 kernel stats of the detector alone (tools/fast_quick.py)
set -e
R=$GRAFT_REPO_ROOT
( cd $R && python3 -c "import __graft_entry__ as g; g.build()" > /dev/null )
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_fast -o fast -- python3 $R/tools/fast_quick.py > $R/gpurun_out/prof_fast.log 2>&1
cat $R/gpurun_out/prof_fast.log | tail -3
python3 - <<'PY'
import csv, os
R=os.environ["GRAFT_REPO_ROOT"]
for root, d, f in os.walk(R + "/gpurun_out/prof_fast"):
    for n in f:
        if n.endswith("kernel_stats.csv"):
            for r in csv.DictReader(open(os.path.join(root, n))):
                print("%-60s calls %4s avg %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
