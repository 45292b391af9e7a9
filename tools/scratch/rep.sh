python tools/micro_rank_of_n.py 2 4 8 2>/dev/null | grep "N=" | cut -c1-45,150-
for c in cfg3 cfg4; do python tools/micro_rank_of_n.py 8 --config $c 2>/dev/null | grep "N=" | cut -c1-45,150-; done
