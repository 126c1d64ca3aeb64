cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export SR_MICRO_SHAPES=${SHAPES:-0,2} SR_MICRO_MODES=${MODES:-2}
echo "default lib"; timeout 100 python scripts/wino8_micro.py 2>&1 | grep -v amdgpu
for f in simplerecon_amd/abl/*.so; do echo "$f"; SR_HIP_LIBRARY=$GRAFT_REPO_ROOT/$f timeout 100 python scripts/wino8_micro.py 2>&1 | grep -v amdgpu; done
