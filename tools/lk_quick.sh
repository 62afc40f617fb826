set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_lk.py tests/test_golden.py -m gpu -q --timeout 240 -p no:cacheprovider -x 2>&1 | tail -2
python tools/lk_sweep.py 2>&1 | grep -E "levels 4"
touch 3dr_amd/csrc/lk_kernels.hip
make -C 3dr_amd/csrc EXTRA="-DDR3_DBG_STAMPS" > /dev/null 2>&1
python tools/lk_stamps.py 2>&1 | grep cycles
