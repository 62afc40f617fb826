# FETCH_SIZE / WRITE_SIZE / TA of the tracker at the headline launch shape (tools/lk_time.py): separate passes
set -e
R=$GRAFT_REPO_ROOT
( cd $R && python3 -c "import __graft_entry__ as g; g.build()" > /dev/null )
cd /tmp && export TMPDIR=/tmp
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TA_BUSY_avg TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_lkq_$n -o p -- python3 $R/tools/lk_time.py > $R/gpurun_out/pmc_lkq_$n.log 2>&1
done
python3 $R/tools/pmc_medians.py $R/gpurun_out pmc_lkq_ | grep -A12 lk_trackg
