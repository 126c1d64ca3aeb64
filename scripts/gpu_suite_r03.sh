#!/bin/bash
# Round-3 GPU suite: smoke + full GPU tests + bench lines + rocprofv3 kernel stats + PMC passes.
# usage: scripts/gpu_suite_r03.sh [tests|bench|prof|pmc ...]   (default: everything)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; export TMPDIR=/tmp; O=$R/gpurun_out; mkdir -p $O
WHAT=${@:-tests bench prof pmc}
for w in $WHAT; do case $w in
tests)
  python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
  timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
  tail -2 $O/smoke.log; tail -4 $O/pytest_gpu.log ;;
bench)
  timeout 600 python bench.py --steps 10 --warmup 3 > $O/bench_hero_cfg3.json 2> $O/bench_hero_cfg3.err
  for wl in hero_b1 hero_cfg3_noprior hero_cfg3_graph hero_b1_graph hero_cfg4_stream hero_cfg5_volume; do
    timeout 400 python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_$wl.json 2> $O/bench_$wl.err
  done
  cut -c1-600 $O/bench_hero_cfg3.json; for wl in hero_b1 hero_cfg3_graph hero_b1_graph hero_cfg4_stream hero_cfg5_volume; do cut -c1-160 $O/bench_$wl.json; done ;;
prof)
  cd /tmp
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_hero_cfg3 -o hero_cfg3 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $O/prof_hero_cfg3.log 2>&1
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_hero_cfg5_volume -o hero_cfg5_volume -- python $R/bench.py --workload hero_cfg5_volume --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $O/prof_hero_cfg5.log 2>&1
  cd $R; head -12 $O/prof_hero_cfg3/hero_cfg3_kernel_stats.csv | cut -c1-150 ;;
pmc)
  cd /tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 400 rocprofv3 --pmc $c --output-format csv -d $O/pmc_${c}_hero -o hero -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/pmc_${c}_hero.log 2>&1
    timeout 400 rocprofv3 --pmc $c --output-format csv -d $O/pmc_${c}_cfg5 -o cfg5 -- python $R/bench.py --workload hero_cfg5_volume --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/pmc_${c}_cfg5.log 2>&1
  done
  timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES --output-format csv -d $O/pmc_mfma -o hero -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/pmc_mfma.log 2>&1
  cd $R; ls $O/pmc_FETCH_SIZE_hero | head -3 ;;
esac; done
