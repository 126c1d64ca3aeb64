#!/bin/bash
# cache-policy variants of the wave-specialised F(4x4) kernel's residual / output / patch streams: isolated layers, then the step
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export SR_MICRO_SHAPES=${SR_MICRO_SHAPES:-0,1,5}
LIBS=${@:-product ntres ntout ntro ntall scout ntsc}
for lib in $LIBS; do
  path=$R/simplerecon_amd/alt/libsr_$lib.so; [ "$lib" = product ] && path=""
  echo "== $lib"; SR_HIP_LIBRARY=$path python scripts/wino4_micro.py 2>&1 | grep -v amdgpu | sed 's/w2:.*w4_ws:/w4_ws:/'
done
for rep in 1 2; do for lib in $LIBS; do
  path=$R/simplerecon_amd/alt/libsr_$lib.so; [ "$lib" = product ] && path=""
  SR_HIP_LIBRARY=$path timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', round(d['value'],1), round(d['ms_per_step'],3))"
done; done
