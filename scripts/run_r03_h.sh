cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ddp.py tests/test_gpu_encoder_training.py -q -k "ddp or image_prior" 2>&1 | tail -8
