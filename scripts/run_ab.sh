cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for m in 0 1; do echo "SPLIT=$m"; SR_ENCODER_SPLIT=$m timeout 300 python scripts/effnet_micro.py 2>&1 | grep -v amdgpu | grep "whole"; done
timeout 900 python -m pytest tests/test_gpu_image_encoder.py tests/test_gpu_graph.py tests/test_gpu_depth_model.py -q -x 2>&1 | tail -2
for m in 0 1 0 1; do SR_ENCODER_SPLIT=$m timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('SPLIT=$m', round(d['value'],1), round(d['ms_per_step'],2))"; done
