mkdir -p gpurun_out
R=$(pwd); O=$R/gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
tail -3 $O/pytest_gpu.log
