cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export SR_WINO8=2 SR_MICRO_MODES=2 SR_MICRO_SHAPES=${SHAPES:-0} SR_WINO_TRACE_FILE=$GRAFT_REPO_ROOT/gpurun_out/wino8_trace.bin SR_WINO_TRACE_LAUNCH=10
timeout 300 python scripts/wino8_micro.py > gpurun_out/w8_trace_micro.log 2>&1; cat gpurun_out/w8_trace_micro.log
python scripts/wino8_trace.py gpurun_out/wino8_trace.bin
