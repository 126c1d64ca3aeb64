cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_image_encoder.py tests/test_gpu_e2e_full_size.py tests/test_gpu_depth_model.py tests/test_gpu_matching_encoder.py -x -q 2>&1 | tail -3
for i in 1 2 3; do
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step', round(d['value'],1), round(d['ms_per_step'],2))"; done
timeout 300 python scripts/layer_table.py 8 2>&1 | grep -v amdgpu | head -8
