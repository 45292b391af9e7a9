"""Helper of test_gpu_parity.py::test_exchange_over_rccl_single_rank (run as a script, one process).

A one-rank RCCL process group on cuda:0 with sleekit_amd.dist.always_exchange set: every layer's factor is
packed, all-gathered by RCCL on the comm stream and unpacked before its loop, exactly as on N > 1 ranks;
the shards must equal the plain single-GPU path bit for bit.
"""

import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from sleekit_amd import codebook, engine, synth  # noqa: E402
from sleekit_amd import dist as sdist  # noqa: E402


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", sys.argv[1] if len(sys.argv) > 1 else "29631")
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
    cb = codebook.UniformCodebook(8, -1, 1)
    layers = []
    for i, (R, n) in enumerate(((48, 192), (64, 320), (33, 192), (16, 512))):  # mixed widths: padded payloads
        lay = synth.make_layer(R, n, 77 + i)
        layers.append({k: torch.from_numpy(lay[k]).to(device) for k in ("W", "H", "scale")})
    plain = []
    for lay in layers:
        res = engine.quantize_layer(lay["W"], lay["H"], cb, lay["scale"])
        plain.append((res.Q.cpu().numpy(), res.idx.cpu().numpy()))
    sdist.always_exchange = True
    for overlap in ((2, 2), False):
        backend = sdist.HipBackend(cb, "diag", 0.01, 0, with_error=True, overlap=overlap)
        for join in (True, False):
            shards = sdist.quantize_stream(layers, backend, join=join)
            torch.cuda.synchronize()
            for (Q, idx), sh in zip(plain, shards):
                assert np.array_equal(sh["idx"].cpu().numpy(), idx)
                assert np.array_equal(sh["Q"].cpu().numpy(), Q)
                assert int(sh["info"].item()) == 0
    # the batched-round route (what N > 1 ranks run) over the same RCCL exchange: rounds of one layer, shards of 128-row multiples
    layers2 = []
    for i, (R, n) in enumerate(((256, 512), (128, 320), (256, 512))):
        lay = synth.make_layer(R, n, 177 + i)
        layers2.append({k: torch.from_numpy(lay[k]).to(device) for k in ("W", "H", "scale")})
    layers2[1]["H"] = layers2[1]["H"].clone()
    layers2[1]["H"][0, 3] += 0.5  # asymmetric: the verdict travels with the payload
    for moves in (0, 4):
        backend = sdist.HipBackend(cb, "diag", 0.01, moves, with_error=True, overlap=(2, 2))
        backend.min_batch = 1
        rounds = []
        run_round = backend.run_round
        backend.run_round = lambda *a: (rounds.append(1), run_round(*a))[1]
        shards = sdist.quantize_stream(layers2, backend)
        torch.cuda.synchronize()
        assert len(rounds) == len(layers2)
        for lay, sh in zip(layers2, shards):
            res = engine.quantize_layer(lay["W"], lay["H"], cb, lay["scale"], nb_ls_moves=moves)
            err = engine.row_errors(lay["W"], res.Q, lay["H"])
            assert np.array_equal(sh["idx"].cpu().numpy(), res.idx.cpu().numpy())
            assert np.array_equal(sh["Q"].cpu().numpy(), res.Q.cpu().numpy())
            np.testing.assert_allclose(sh["row_err"].cpu().numpy(), err.cpu().numpy(), rtol=1e-5)
    t = torch.ones(1, device=device)
    dist.all_reduce(t)
    dist.barrier()
    dist.destroy_process_group()
    print("RCCL_SINGLE_RANK_OK")


if __name__ == "__main__":
    main()
