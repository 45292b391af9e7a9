#!/usr/bin/env python3
"""Micro-benchmark (GPU box): per-kernel times of the factorisation at several n, for each form of the panel step
(panel_split = 3: the chain of round 4 (default); 2: the panel kernel, one launch per panel; 1: two launches per panel).
    python tools/micro_factor.py [n ...]        FORMS=3,2,1  BATCH=1  LOOKAHEAD=0"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from sleekit_amd import _lib, engine
forms = [int(x) for x in os.environ.get("FORMS", "3,2,1").split(",")]
batch = int(os.environ.get("BATCH", "1"))
ahead = bool(int(os.environ.get("LOOKAHEAD", "0")))
for n in [int(x) for x in sys.argv[1:]] or [768, 4096]:
    rng = np.random.default_rng(n)
    A = rng.standard_normal((n, n + 8)).astype(np.float32)
    H = torch.from_numpy((A @ A.T / n + 0.1 * np.eye(n)).astype(np.float32)).cuda()
    Hs = [H.clone() for _ in range(batch)]

    def run():
        if batch > 1:
            return engine.factorize_batch(Hs, n, 0.01, _lib.ORDER_DIAG)
        return engine.factorize(H, n, 0.01, _lib.ORDER_DIAG, lookahead=ahead)

    for form in forms:
        with _lib.option("panel_split", form):
            for rep in range(3):
                run()
            torch.cuda.synchronize()
            t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
            t0.record()
            for rep in range(5):
                run()
            t1.record(); torch.cuda.synchronize()
            _lib.lib.slk_profile_reset(); _lib.lib.slk_profile_enable(1)
            for rep in range(5):
                run()
            torch.cuda.synchronize()
            _lib.lib.slk_profile_enable(0)
            print(f"n={n} batch={batch} form={form} lookahead={int(ahead)}: whole factorize {t0.elapsed_time(t1)/5*1e3:.1f} us (unprofiled)", flush=True)
            for k in _lib.profile_report():
                print(f"   {k['kernel']:<18s} {k['launches']//5:4d}/call  avg {1e3*k['total_ms']/k['launches']:8.2f} us  total/call {k['total_ms']/5:8.3f} ms")
            _lib.lib.slk_profile_reset()
