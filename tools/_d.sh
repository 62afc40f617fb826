set -e
timeout -k 10 300 python -m pytest tests/test_gpu_strides.py tests/test_harness.py -x -q -m gpu 2>&1 | tail -2
timeout -k 10 200 python tools/host_api_probe.py 2>&1 | grep -v amdgpu.ids | head -4
timeout -k 10 600 python -m pytest tests/test_gpu_lk.py -x -q -m gpu 2>&1 | tail -2
