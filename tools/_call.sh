mkdir -p gpurun_out
R=$(pwd); O=$R/gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
tail -3 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
