cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export SR_MICRO_SHAPES=0,1 SR_MICRO_MODES=0,2
for st in 0 8000 30000 60000 0; do echo "STAGGER_CU=$st"; SR_WINO_STAGGER_CU=$st timeout 200 python scripts/wino8_micro.py 2>&1 | grep -v amdgpu; done
