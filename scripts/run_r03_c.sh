cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_mlp_volume.py tests/test_gpu_depth_model.py tests/test_gpu_e2e_full_size.py -x -q 2>&1 | tail -3
for v in 0 1; do for wl in hero_cfg3_volume hero_cfg5_volume; do
SR_MLP_VEC_STORE=$v timeout 300 python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('vec=$v $wl', round(d['value'],1), round(d['ms_per_step'],3))"; done; done
for v in 0 1; do SR_MLP_VEC_STORE=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('vec=$v step', round(d['value'],1), round(d['ms_per_step'],3))"; done
