mkdir -p gpurun_out
python bench.py --gpus 1 --force-dist --graph off --steps 200 --warmup 20 --headline-only --device-warm-steps 50 > gpurun_out/fdq_default.out 2>/dev/null
