# A/B of libraries on the Winograd layer shapes + the whole step:  scripts/run_ab.sh lib1 lib2 ...  ("product" = in-tree)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
LIBS=${@:-product}
for lib in $LIBS; do
  path=simplerecon_amd/alt/libsr_$lib.so; [ "$lib" = product ] && path=""
  echo "== library: $lib"
  SR_HIP_LIBRARY=$path SR_MICRO_SHAPES=${SR_MICRO_SHAPES:-0,1,3,5,6,8} SR_AB_VAR=SR_WINO_XCD SR_AB_VALUES=1 timeout 300 python scripts/wino_ab.py 2>&1 | grep -v "amdgpu\|library:"
done
for rep in 1 2; do for lib in $LIBS; do
  path=simplerecon_amd/alt/libsr_$lib.so; [ "$lib" = product ] && path=""
  SR_HIP_LIBRARY=$path timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', round(d['value'],1), round(d['ms_per_step'],2))"
done; done
