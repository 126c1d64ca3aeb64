cd $GRAFT_REPO_ROOT
for lib in w4trace w4trace128 w4trace64; do for sh in 0 1; do
echo "== $lib shape $sh"; SR_MICRO_SHAPES=$sh SR_HIP_LIBRARY=$GRAFT_REPO_ROOT/simplerecon_amd/alt/libsr_$lib.so SR_W4_TRACE_LAUNCH=12 python scripts/wino4_micro.py 2>&1 | grep "W4CLOCK\|w4_ws" | sed 's/w2:.*w4_ws:/w4_ws:/'
done; done
