cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_conv.py -x -q 2>&1 | tail -3
echo "=== regv1 (product lib)"; timeout 300 python scripts/wino_ab.py 2>&1 | grep -v amdgpu
echo "=== regv0"; SR_HIP_LIBRARY=$GRAFT_REPO_ROOT/simplerecon_amd/alt/libsr_regv0.so timeout 300 python scripts/wino_ab.py 2>&1 | grep -v amdgpu
for lib in "" $GRAFT_REPO_ROOT/simplerecon_amd/alt/libsr_regv0.so; do for x in 0 1; do
SR_HIP_LIBRARY=$lib SR_WINO_XCD=$x timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=$lib xcd=$x', round(d['value'],1), round(d['ms_per_step'],2))"; done; done
