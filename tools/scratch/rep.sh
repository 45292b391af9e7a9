for w in 5 15 30; do WARMUP=$w python tools/micro_rank_of_n.py 8 --config cfg3 2>/dev/null | grep "N=" | cut -c1-45,150-; done
for w in 5 15; do WARMUP=$w python tools/micro_rank_of_n.py 8 --config cfg4 2>/dev/null | grep "N=" | cut -c1-45,150-; done
for w in 5 15; do WARMUP=$w python tools/micro_rank_of_n.py 2 2>/dev/null | grep "N=" | cut -c1-45,150-; done
