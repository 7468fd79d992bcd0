mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_gpu_parity_full.py -m gpu -x -q -k "kitti_default_r1200_n64 and fp32" 2>&1 | tail -2
