cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for x in 0 1; do echo "=== XCD=$x"; SR_WINO_XCD=$x timeout 300 python scripts/layer_table.py 8 2>&1 | grep -v amdgpu | head -14; done
