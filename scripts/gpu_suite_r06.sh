#!/bin/bash
# Round-6 GPU suite: smoke + full GPU tests + bench lines + rocprofv3 kernel stats + PMC passes.
# usage: scripts/gpu_suite_r06.sh [tests|bench|fenced|prof|pmc|pmcdot ...]   (default: everything but `fenced`)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; export TMPDIR=/tmp; O=$R/gpurun_out; mkdir -p $O
WHAT=${@:-tests bench prof pmc pmcdot}
for w in $WHAT; do case $w in
tests)
  python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
  timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
  tail -2 $O/smoke.log; tail -4 $O/pytest_gpu.log ;;
bench)
  timeout 600 python bench.py --steps 20 --warmup 3 > $O/bench_hero_cfg3.json 2> $O/bench_hero_cfg3.err
  for wl in hero_b1 hero_cfg3_noprior hero_cfg3_graph hero_b1_graph hero_cfg4_stream hero_cfg5_volume hero_cfg5 dot_cfg2 dot_b8; do
    timeout 400 python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_$wl.json 2> $O/bench_$wl.err
  done
  for try in 1 2 3; do   # (the RCCL world-1 line came back empty once in a while right behind the other bench processes: retry)
    timeout 600 python bench.py --gpus 1 --force-collective --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_hero_cfg3_rccl_world1.json 2> $O/bench_rccl.err
    [ -s $O/bench_hero_cfg3_rccl_world1.json ] && break; sleep 5
  done
  cut -c1-400 $O/bench_hero_cfg3.json; for wl in hero_b1 hero_cfg3_noprior hero_cfg3_graph hero_b1_graph hero_cfg4_stream hero_cfg5_volume hero_cfg5 dot_cfg2 dot_b8 hero_cfg3_rccl_world1; do cut -c1-190 $O/bench_$wl.json; done ;;
fenced)   # the split-precision experiments: never the headline
  for wl in hero_cfg3_bf16x3 hero_cfg3_f16x3 hero_cfg3_bf16x3_convs hero_cfg3_f16x3_convs hero_b1_graph_f16x3_convs; do
    timeout 400 python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_$wl.json 2> $O/bench_$wl.err
    cut -c1-190 $O/bench_$wl.json
  done
  timeout 300 python scripts/mlp_split_check.py > $O/mlp_split_check.txt 2>&1
  timeout 300 python scripts/wino_split_check.py > $O/wino_split_check.txt 2>&1
  cd /tmp
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_hero_cfg3_f16x3_convs -o f16x3 -- python $R/bench.py --workload hero_cfg3_f16x3_convs --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $O/prof_f16x3.log 2>&1
  cd $R; head -8 $O/prof_hero_cfg3_f16x3_convs/f16x3_kernel_stats.csv | cut -c1-150 ;;
prof)
  cd /tmp
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_hero_cfg3 -o hero_cfg3 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $O/prof_hero_cfg3.log 2>&1
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_hero_b1_graph -o hero_b1_graph -- python $R/bench.py --workload hero_b1_graph --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $O/prof_hero_b1_graph.log 2>&1
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_hero_cfg5 -o hero_cfg5 -- python $R/bench.py --workload hero_cfg5 --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > $O/prof_hero_cfg5.log 2>&1
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_effnet -o effnet -- python $R/scripts/effnet_micro.py 8 > $O/prof_effnet.log 2>&1
  cd $R; head -12 $O/prof_hero_cfg3/hero_cfg3_kernel_stats.csv | cut -c1-150 ;;
pmc)
  cd /tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 400 rocprofv3 --pmc $c --output-format csv -d $O/pmc_${c}_hero -o hero -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/pmc_${c}_hero.log 2>&1
    timeout 400 rocprofv3 --pmc $c --output-format csv -d $O/pmc_${c}_cfg5 -o cfg5 -- python $R/bench.py --workload hero_cfg5_volume --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/pmc_${c}_cfg5.log 2>&1
  done
  timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES --output-format csv -d $O/pmc_mfma -o hero -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/pmc_mfma.log 2>&1
  cd $R; ls $O/pmc_FETCH_SIZE_hero | head -3 ;;
pmcdot)
  rm -rf $O/pmc_ldsB8? $O/pmc_ldsB1?
  SR_MICRO_B=8 bash scripts/pmc_dot.sh ldsB8 > $O/pmc_dot_b8.log 2>&1
  SR_MICRO_B=1 bash scripts/pmc_dot.sh ldsB1 > $O/pmc_dot_b1.log 2>&1
  tail -3 $O/pmc_dot_b8.log | cut -c1-300 ;;
esac; done
