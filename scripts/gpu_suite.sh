#!/bin/bash
# full GPU suite + headline benches + kernel-trace profiles + PMC passes (HBM traffic, MFMA utilisation);
# afterwards, in the build container: python scripts/pmc_traffic.py r01; python scripts/pmc_mfma.py r01; copy bench_*.json / *_kernel_stats.csv to profiles/
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; export TMPDIR=/tmp; O=$R/gpurun_out; mkdir -p $O
python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 600 python bench.py --steps 10 --warmup 3 > $O/bench_hero_cfg3.json 2> $O/bench_hero_cfg3.err
timeout 300 python bench.py --workload hero_b1 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_hero_b1.json 2> $O/bench_hero_b1.err
timeout 300 python bench.py --workload hero_cfg3_noprior --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_hero_cfg3_noprior.json 2> $O/bench_hero_cfg3_noprior.err
timeout 300 python bench.py --workload tsdf_fuse --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_tsdf_fuse.json 2> $O/bench_tsdf_fuse.err
timeout 300 python bench.py --workload dot_cfg2 --steps 50 --warmup 5 > $O/bench_dot_cfg2.json 2> $O/bench_dot_cfg2.err
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_hero_cfg3 -o hero_cfg3 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $O/prof_hero_cfg3.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_dot_cfg2 -o dot_cfg2 -- python $R/bench.py --workload dot_cfg2 --steps 20 --warmup 2 --no-cpu-baseline --no-roofline > $O/prof_dot_cfg2.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o hero -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o hero -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/pmc_write.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_dot -o dot -- python $R/bench.py --workload dot_cfg2 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/pmc_fetch_dot.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_dot -o dot -- python $R/bench.py --workload dot_cfg2 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/pmc_write_dot.log 2>&1
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES --output-format csv -d $O/pmc_mfma -o hero -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/pmc_mfma.log 2>&1
cd $R; tail -2 $O/smoke.log; tail -3 $O/pytest_gpu.log; cat $O/bench_hero_cfg3.json; ls $O/pmc_fetch $O/pmc_write | head
