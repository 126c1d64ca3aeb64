cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
for m in 0 1; do SR_WINO8=$m timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_w8_$m.json 2> gpurun_out/bench_w8_$m.err; cut -c1-120 gpurun_out/bench_w8_$m.json; done
