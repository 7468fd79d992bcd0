mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity_full.py tests/test_gpu_graph.py tests/test_gpu_loss_side.py -m gpu -x -q -k "(kitti_c2_r1200_n128 and bf16) or graphed_step_replays or channels_last or source_loss" 2>&1 | tail -3
python bench.py --gpus 1 --steps 20 --warmup 5 --headline-only 2>/dev/null | tail -1 | cut -c1-200
