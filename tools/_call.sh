mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_render.py tests/test_gpu_graph.py tests/test_gpu_dp.py tests/test_gpu_loss_side.py -m gpu -x -q 2>&1 | tail -2
export SRF_COMMIT=e1da126
bash tools/profile_round.sh r04_n > gpurun_out/r04_n_profile_round.log 2>&1
tail -1 gpurun_out/r04_n_profile_round.log | cut -c1-260
