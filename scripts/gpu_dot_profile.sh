#!/bin/bash
# Dot-product sweep artefacts: bench lines (B=1 cfg2, B=8), rocprofv3 kernel stats of the same commands, PMC passes
# (SQ / LDS / TA / L2 + FETCH_SIZE / WRITE_SIZE in separate passes) for the LDS-staged kernel and the L1-gather kernel.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; export TMPDIR=/tmp; O=$R/gpurun_out; mkdir -p $O
for wl in dot_cfg2 dot_b8; do
  timeout 300 python bench.py --workload $wl --steps 50 --warmup 5 > $O/bench_$wl.json 2> $O/bench_$wl.err
done
SR_DOT_LDS=0 timeout 300 python bench.py --workload dot_b8 --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_dot_b8_l1gather.json 2> $O/bench_dot_b8_l1gather.err
cd /tmp
for wl in dot_cfg2 dot_b8; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$wl -o $wl -- python $R/bench.py --workload $wl --steps 20 --warmup 2 --no-cpu-baseline --no-roofline > $O/prof_$wl.log 2>&1
done
cd $R
SR_MICRO_B=8 scripts/pmc_dot.sh ldsB8 > $O/pmc_dot_ldsB8.log 2>&1
SR_MICRO_B=1 scripts/pmc_dot.sh ldsB1 > $O/pmc_dot_ldsB1.log 2>&1
SR_DOT_LDS=0 SR_MICRO_B=8 scripts/pmc_dot.sh l1qB8 > $O/pmc_dot_l1qB8.log 2>&1
cat $O/bench_dot_cfg2.json $O/bench_dot_b8.json $O/bench_dot_b8_l1gather.json; cat $O/pmc_dot_ldsB8.log $O/pmc_dot_ldsB1.log $O/pmc_dot_l1qB8.log
