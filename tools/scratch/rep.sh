set -o pipefail
mkdir -p gpurun_out/final
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 &&
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final/bench_d.json 2> gpurun_out/final/bench_d.err && echo bench-ok
