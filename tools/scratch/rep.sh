set -o pipefail
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 &&
for c in cfg2 cfg3 cfg4; do python tools/micro_rank_of_n.py 8 --config $c 2>/dev/null | grep "N=" | cut -c1-45,150-; done &&
python tools/micro_rank_of_n.py 4 8 2>/dev/null | grep "N=" | cut -c1-45,150- &&
python tools/micro_rank_of_n.py 2 4 --config cfg2 2>/dev/null | grep "N=" | cut -c1-45,150-
