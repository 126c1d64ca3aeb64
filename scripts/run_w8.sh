cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_conv.py -q -x -k "8_wave" > gpurun_out/w8_test.log 2>&1; tail -3 gpurun_out/w8_test.log
timeout 300 python scripts/wino8_micro.py > gpurun_out/w8_micro.log 2>&1; cat gpurun_out/w8_micro.log
