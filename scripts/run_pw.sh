cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(echo "== direct kernel (sr_pw_kernel), forced plans: channel tiles per wave x K split across the waves"; python scripts/pw_sweep.py 2>&1 | grep -v amdgpu
 echo "== LDS-tiled kernel (sr_pw_tiled_kernel), forced plans: tile config (0: 64x128, 1: 128x160, 2: 128x64, 3: 64x64) x K split across workgroups"; SR_SWEEP_TILED=1 python scripts/pw_sweep.py 2>&1 | grep -v amdgpu) > gpurun_out/pw_plan_sweep.txt
for m in 0 auto 0 auto; do SR_PW_TILED=$m timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tiled=$m', round(d['value'],1), round(d['ms_per_step'],2))"; done
