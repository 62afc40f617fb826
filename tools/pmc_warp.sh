# counters of the warp legs (tools/warp_quick.py): issue / wait / cache / traffic, one rocprofv3 --pmc pass per counter group
# (tools/pmc_generic.sh: FETCH_SIZE and WRITE_SIZE have passes of their own, a failed pass fails the script)
set -e
bash $GRAFT_REPO_ROOT/tools/pmc_generic.sh warpq tools/warp_quick.py
