mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity_full.py -m gpu -x -q -k "(kitti_default_r1200_n64 and bf16) or (kitti_c5_r32_n512 and bf16) or (bf_c4 and bf16)" 2>&1 | tail -2
