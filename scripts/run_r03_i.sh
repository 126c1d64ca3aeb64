cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python scripts/train_step_micro.py 2 7 0 2>&1 | grep -v amdgpu
timeout 600 python scripts/train_step_micro.py 2 7 1 2>&1 | grep -v amdgpu
timeout 600 python scripts/train_step_micro.py 8 7 0 2>&1 | grep -v amdgpu
cd /tmp; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_train_r03 -o train -- python $GRAFT_REPO_ROOT/scripts/train_step_micro.py 8 7 0 > $GRAFT_REPO_ROOT/gpurun_out/prof_train_r03.log 2>&1
cd $GRAFT_REPO_ROOT; tail -2 gpurun_out/prof_train_r03.log; find gpurun_out/prof_train_r03 -name "*kernel_stats.csv" | head -2; f=$(find gpurun_out/prof_train_r03 -name "*kernel_stats.csv" | head -1); head -25 $f | cut -c1-160
find gpurun_out/prof_train_r03 -name "*kernel_trace.csv" -delete
