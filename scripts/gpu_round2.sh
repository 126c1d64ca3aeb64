#!/bin/bash
# tests + hero bench + kernel-trace profile
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; export TMPDIR=/tmp; O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 600 python bench.py --steps 10 --warmup 2 > $O/bench_hero_cfg3.json 2> $O/bench_hero_cfg3.err
timeout 300 python bench.py --workload hero_b1 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_hero_b1.json 2> $O/bench_hero_b1.err
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_hero_cfg3 -o hero_cfg3 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $O/prof_hero_cfg3.log 2>&1
cd $R; tail -5 $O/pytest_gpu.log; cat $O/bench_hero_cfg3.json $O/bench_hero_b1.json; tail -3 $O/bench_hero_cfg3.err; find $O/prof_hero_cfg3 -name "*stats*" | head
