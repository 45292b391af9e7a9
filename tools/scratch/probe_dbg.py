import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from sleekit_amd import _device as dev
g = dev.queue_groups()
print([len(x) for x in g], [[s.stream_id for s in x] for x in g])
