set -o pipefail
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 && python tools/scratch/lat_one.py 2>/dev/null | tail -4 | tr '\n' ' '
