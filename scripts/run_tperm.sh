cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export SR_MICRO_SHAPES=0,1,3,5,6 SR_MICRO_MODES=0,2
echo "permuted (default lib)"; timeout 200 python scripts/wino8_micro.py 2>&1 | grep -v amdgpu
echo "linear"; SR_HIP_LIBRARY=$GRAFT_REPO_ROOT/simplerecon_amd/abl/lib_tlinear.so timeout 200 python scripts/wino8_micro.py 2>&1 | grep -v amdgpu
echo "permuted again"; timeout 200 python scripts/wino8_micro.py 2>&1 | grep -v amdgpu
timeout 600 python -m pytest tests/test_gpu_conv.py -q -x 2>&1 | tail -2
