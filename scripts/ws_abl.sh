cd $GRAFT_REPO_ROOT
for lib in product abl1 abl2 abl3 abl4 abl8 abl16 abl31; do
  path=simplerecon_amd/alt/libsr_$lib.so; [ "$lib" = product ] && path=""
  echo "== lib=$lib"
  SR_HIP_LIBRARY=$path SR_MICRO_SHAPES=0,1,3 timeout 200 python scripts/wino_split_check.py 2>&1 | grep "^(" | cut -c1-27,108-130
done
