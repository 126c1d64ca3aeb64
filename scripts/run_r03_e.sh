cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_image_encoder.py -x -q 2>&1 | tail -3
for lib in "" $GRAFT_REPO_ROOT/simplerecon_amd/alt/libsr_pipe0.so "" $GRAFT_REPO_ROOT/simplerecon_amd/alt/libsr_pipe0.so; do
SR_HIP_LIBRARY=$lib timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=$lib', round(d['value'],1), round(d['ms_per_step'],2))"; done
echo "=== layer table"; timeout 300 python scripts/layer_table.py 8 2>&1 | grep -v amdgpu | head -12
echo "=== trace 64->64"; SR_AB_VALUES=1 SR_MICRO_SHAPES=0 SR_HIP_LIBRARY=$GRAFT_REPO_ROOT/simplerecon_amd/alt/libsr_trace.so SR_WINO_TRACE_FILE=$GRAFT_REPO_ROOT/gpurun_out/wino_trace.bin SR_WINO_TRACE_LAUNCH=10 timeout 300 python scripts/wino_ab.py 2>&1 | grep -v amdgpu
python scripts/wino_trace.py gpurun_out/wino_trace.bin
