cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export SR_MICRO_MODES=0 SR_MICRO_SHAPES=0,1,3,5,6
for r in 1 2; do
echo "default"; timeout 300 python scripts/wino8_micro.py 2>&1 | grep -v amdgpu | cut -c1-80
echo "zero-C peeled"; SR_HIP_LIBRARY=$GRAFT_REPO_ROOT/simplerecon_amd/abl/lib_zeroc.so timeout 300 python scripts/wino8_micro.py 2>&1 | grep -v amdgpu | cut -c1-80
done
