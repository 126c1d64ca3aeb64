cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_conv.py -x -q 2>&1 | tail -3
echo "=== pipe1 (product lib)"; SR_AB_VALUES=1 timeout 300 python scripts/wino_ab.py 2>&1 | grep -v amdgpu
echo "=== pipe0"; SR_AB_VALUES=1 SR_HIP_LIBRARY=$GRAFT_REPO_ROOT/simplerecon_amd/alt/libsr_pipe0.so timeout 300 python scripts/wino_ab.py 2>&1 | grep -v amdgpu
for lib in "" $GRAFT_REPO_ROOT/simplerecon_amd/alt/libsr_pipe0.so "" $GRAFT_REPO_ROOT/simplerecon_amd/alt/libsr_pipe0.so; do
SR_HIP_LIBRARY=$lib timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=$lib', round(d['value'],1), round(d['ms_per_step'],2))"; done
echo "=== layer table pipe1"; timeout 300 python scripts/layer_table.py 8 2>&1 | grep -v amdgpu | head -12
