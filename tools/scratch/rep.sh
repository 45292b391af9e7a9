set -o pipefail
SLK_INNER=256 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "order_and_factor or panel_step or lookahead or large_cases or cfg5" 2>&1 | tail -2 &&
for i in 1 2; do for v in 0 256; do
SLK_INNER=$v python bench.py --config cfg3 --steps 8 --warmup 4 --no-configs --no-extras --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg3 inner=$v', j['value'], j['ms_per_step'])"
SLK_INNER=$v python bench.py --steps 20 --warmup 5 --no-configs --no-extras --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline inner=$v', j['value'], j['ms_per_step'])"
done; done
