cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_conv.py -q -x -k "pointwise or library" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_image_encoder.py tests/test_gpu_depth_model.py -q -x 2>&1 | tail -3
for m in lib pw lib pw; do SR_CONV1X1_GEMM=$m timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m', round(d['value'],1), round(d['ms_per_step'],2))"; done
for m in lib pw; do echo "== $m"; SR_CONV1X1_GEMM=$m timeout 300 python scripts/effnet_micro.py 2>&1 | grep -v amdgpu | grep "stage\|whole\|block1"; done
