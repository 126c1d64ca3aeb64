cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export SR_MICRO_MODES=0 SR_MICRO_SHAPES=0,1,3,5
timeout 300 python scripts/wino8_micro.py 2>&1 | grep -v amdgpu | cut -c1-80
timeout 900 python -m pytest tests/test_gpu_conv.py -q -x 2>&1 | tail -2
for i in 1 2; do timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hero_cfg3', round(d['value'],1), round(d['ms_per_step'],2))"; done
