for b in 8 12 16 24; do
SLK_LOCAL_BATCH=$b python bench.py --config cfg2 --steps 20 --warmup 6 --no-configs --no-extras --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 batch=$b', j['value'], j['ms_per_step'])"
SLK_LOCAL_BATCH=$b python bench.py --config cfg4 --steps 8 --warmup 4 --no-configs --no-extras --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg4 batch=$b', j['value'], j['ms_per_step'])"
done
