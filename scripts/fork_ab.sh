# A/B of the decoder's branch-fork threshold (SR_DECODER_FORK_REGIONS) on the whole step, alternating on one box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for rep in 1 2; do for v in ${@:-1300 0 400 700 5000}; do
  SR_DECODER_FORK_REGIONS=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fork_regions=$v', round(d['value'],1), round(d['ms_per_step'],3))"
done; done
