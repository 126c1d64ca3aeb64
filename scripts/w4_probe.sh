#!/bin/bash
# F(4x4) wave-specialised kernel probes: ablation builds + the s_memtime trace on the 64 -> 64 full-resolution layer.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; export TMPDIR=/tmp; O=$R/gpurun_out; mkdir -p $O
export SR_MICRO_SHAPES=${SR_MICRO_SHAPES:-0,1}
echo "== product"; python scripts/wino4_micro.py 2>&1 | tail -3
for lib in $(ls simplerecon_amd/alt/libsr_w4abl*.so 2>/dev/null); do
  echo "== $lib"; SR_HIP_LIBRARY=$R/$lib python scripts/wino4_micro.py 2>&1 | tail -3
done
if [ -f simplerecon_amd/alt/libsr_w4trace.so ]; then
  echo "== trace"; SR_MICRO_SHAPES=0 SR_HIP_LIBRARY=$R/simplerecon_amd/alt/libsr_w4trace.so SR_W4_TRACE_LAUNCH=${SR_W4_TRACE_LAUNCH:-30} python scripts/wino4_micro.py 2>&1 | python scripts/w4_trace_fmt.py
fi
