cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python scripts/effnet_micro.py 8 2>&1 | grep -v amdgpu | tail -12
cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_effnet_r03 -o effnet -- python $GRAFT_REPO_ROOT/scripts/effnet_micro.py 8 > $GRAFT_REPO_ROOT/gpurun_out/prof_effnet_r03.log 2>&1
cd $GRAFT_REPO_ROOT; f=$(find gpurun_out/prof_effnet_r03 -name "*kernel_stats.csv" | head -1); head -40 $f | cut -c1-150
find gpurun_out/prof_effnet_r03 -name "*kernel_trace.csv" -delete
