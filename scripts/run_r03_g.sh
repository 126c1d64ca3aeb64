cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_mlp_volume.py tests/test_gpu_e2e_full_size.py tests/test_gpu_depth_model.py tests/test_gpu_conv.py tests/test_gpu_image_encoder.py -x -q 2>&1 | tail -3
for lib in "" $GRAFT_REPO_ROOT/simplerecon_amd/alt/libsr_head.so "" $GRAFT_REPO_ROOT/simplerecon_amd/alt/libsr_head.so; do
SR_HIP_LIBRARY=$lib timeout 300 python bench.py --workload hero_cfg3_volume --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('volume lib=$lib', round(d['value'],1), round(d['ms_per_step'],3))"
SR_HIP_LIBRARY=$lib timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step lib=$lib', round(d['value'],1), round(d['ms_per_step'],2))"; done
