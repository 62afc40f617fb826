python -m pytest tests/test_gpu_stitch.py tests/test_golden.py tests/test_gpu_warp.py tests/test_harness.py -x -q -m gpu > gpurun_out/r4_stitch_tests.log 2>&1; tail -15 gpurun_out/r4_stitch_tests.log
python tools/warp_quick.py > gpurun_out/r4_warp_quick_strips.log 2>&1; tail -2 gpurun_out/r4_warp_quick_strips.log
DR3HIP_LIB=3dr_amd/variants/libdr3hip_nostrips.so python tools/warp_quick.py > gpurun_out/r4_warp_quick_nostrips.log 2>&1; tail -2 gpurun_out/r4_warp_quick_nostrips.log
