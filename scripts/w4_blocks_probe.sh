#!/bin/bash
# wave-specialised F(4x4) kernel with a capped grid: per-round time vs the number of workgroups on the chip
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export SR_MICRO_SHAPES=${SR_MICRO_SHAPES:-0,1,5}
for rep in 1 2; do
echo "== 256 (product)"; python scripts/wino4_micro.py 2>&1 | grep -v amdgpu | sed 's/w2:.*w4_ws:/w4_ws:/'
for n in 224 192 128 64; do echo "== $n workgroups"; SR_HIP_LIBRARY=$R/simplerecon_amd/alt/libsr_mb$n.so python scripts/wino4_micro.py 2>&1 | grep -v amdgpu | sed 's/w2:.*w4_ws:/w4_ws:/'; done
done
