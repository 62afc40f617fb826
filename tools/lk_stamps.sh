set -e
cd $GRAFT_REPO_ROOT
touch 3dr_amd/csrc/lk_kernels.hip
make -C 3dr_amd/csrc EXTRA="-DDR3_DBG_STAMPS" > /dev/null 2>&1
python tools/lk_stamps.py 2>&1 | grep cycles
