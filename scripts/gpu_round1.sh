#!/bin/bash
# First GPU contact: smoke, GPU parity tests, bench lines, kernel-trace profile.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; export TMPDIR=/tmp; O=$R/gpurun_out; mkdir -p $O
rocm-smi --showproductname 2>/dev/null | head -8 > $O/gpu.txt; nproc >> $O/gpu.txt; lscpu | grep "Model name" >> $O/gpu.txt
python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 300 python bench.py --steps 50 --warmup 5 > $O/bench_dot_cfg2.json 2> $O/bench_dot_cfg2.err
timeout 300 python bench.py --workload dot_b8 --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_dot_b8.json 2> $O/bench_dot_b8.err
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_dot_cfg2 -o dot_cfg2 -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $O/prof_dot_cfg2.log 2>&1
cd $R; tail -3 $O/smoke.log; tail -5 $O/pytest_gpu.log; cat $O/bench_dot_cfg2.json $O/bench_dot_b8.json; ls $O/prof_dot_cfg2 | head
