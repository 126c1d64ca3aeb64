cd $GRAFT_REPO_ROOT
for v in 2 4; do
  echo "== SR_POOL_BW=$v"
  SR_POOL_BW=$v timeout 200 python scripts/menc_micro.py 2>&1 | grep "maxpool\|whole"
done
timeout 300 python -m pytest tests/test_gpu_matching_encoder.py -q -x 2>&1 | tail -2
