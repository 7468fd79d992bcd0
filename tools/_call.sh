mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_stages.py -q -x -k "mlp_backward" 2>&1 | grep -v "rel L2:" | tail -60 ) > gpurun_out/r04_b_stage_tests.log 2>&1
( timeout 300 python tools/wide_probe.py 153600 10 2>&1 | tail -12 ) > gpurun_out/r04_b_wide_probe.log 2>&1
( timeout 300 python tools/ab_step.py "renderer.LINOUT_SCRATCH=True,False" 3 40 2>&1 | tail -3 ) > gpurun_out/r04_b_ab_scratch.log 2>&1
( timeout 300 python tools/ab_step.py "cfg.bwd_kernel='wide','wide_staged'" 3 40 2>&1 | tail -3 ) > gpurun_out/r04_b_ab_bwd.log 2>&1
cat gpurun_out/r04_b_stage_tests.log gpurun_out/r04_b_wide_probe.log gpurun_out/r04_b_ab_scratch.log gpurun_out/r04_b_ab_bwd.log | tail -100
