mkdir -p gpurun_out/final
SLK_PANEL_SPLIT=2 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/final/bench_a.json 2>/dev/null
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --streams 3,1 > gpurun_out/final/bench_b.json 2>/dev/null
