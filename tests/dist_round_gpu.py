"""Helper of test_gpu_parity.py::test_batched_rounds_two_ranks_one_gpu (launched by torch.distributed.run, 2 ranks).

Both ranks share cuda:0 and talk over gloo (RCCL refuses two ranks on one GPU): the row shards of a round's
layers go through the batched loop (HipBackend.run_round; also with local-search moves) and must equal, bit for bit, the rows of the
unsharded single-GPU result.  The stream is in MODEL order (shapes alternate): sleekit_amd.dist buckets it by shape, pads the ragged
shards to whole tiles, and only the lone last layer takes the layer-by-layer route.
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
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    dist.init_process_group("gloo")
    rank, size = dist.get_rank(), dist.get_world_size()
    cb = codebook.UniformCodebook(8, -1, 1)
    # a model-order stream: shapes alternate like the layers of a transformer block; (100, 192) shards are ragged (50 rows)
    shapes = [(256, 512), (512, 1024), (100, 192), (256, 512), (512, 1024), (100, 192), (256, 320)]
    layers = []
    for i, (R, n) in enumerate(shapes):
        lay = synth.make_layer(R, n, 300 + i)
        layers.append({k: torch.from_numpy(lay[k]).to(device) for k in ("W", "H", "scale")})
    layers[1]["H"] = layers[1]["H"].clone()
    layers[1]["H"][2, 5] += 0.125  # not symmetric: that layer's error takes the float32 kernel inside the batch
    calls = {"round": 0, "rows": 0}
    for overlap, moves, on_fs in (((2, 2), 0, True), (False, 0, True), ((2, 2), 6, True), ((2, 1), 0, False)):
        backend = sdist.HipBackend(cb, "diag", 0.01, moves, with_error=True, overlap=overlap)
        backend.rounds_on_factor_streams = on_fs  # (False: the rounds on a loop stream of their own, bench.py from 4 ranks up)
        run_round, run_rows = backend.run_round, backend.run_rows
        backend.run_round = lambda *a: (calls.__setitem__("round", calls["round"] + 1), run_round(*a))[1]
        backend.run_rows = lambda *a: (calls.__setitem__("rows", calls["rows"] + 1), run_rows(*a))[1]
        shards = sdist.quantize_stream(layers, backend)
        torch.cuda.synchronize()
        for lay, sh in zip(layers, shards):
            lo, hi = sh["rows"]
            assert (lo, hi) == sdist.row_range(lay["W"].shape[0], rank, size)
            res = engine.quantize_layer(lay["W"], lay["H"], cb, lay["scale"], nb_ls_moves=moves)
            err = engine.row_errors(lay["W"], res.Q, lay["H"])
            assert np.array_equal(sh["Q"].cpu().numpy(), res.Q[lo:hi].cpu().numpy())
            assert np.array_equal(sh["idx"].cpu().numpy(), res.idx[lo:hi].cpu().numpy())
            np.testing.assert_allclose(sh["row_err"].cpu().numpy(), err[lo:hi].cpu().numpy(), rtol=1e-5)
            assert int(sh["info"].item()) == 0
    # bucketed by shape: rounds (0, 3), (1, 4) and the ragged (2, 5) -- padded to whole tiles -- batched; (6) alone
    assert calls["round"] == 4 * 3 and calls["rows"] == 4 * 1, calls
    # GROUPS of rounds (small layers): 7 layers of one shape on 2 ranks = 4 rounds; the backend wants up to 64 such layers in a
    # loop batch, so they go as ONE group -- each rank's layers through one batched factorisation, one all-gather with a slot
    # per round, one stacked loop over all seven -- and every shard still equals the unsharded layer's rows bit for bit
    small = []
    for i in range(7):
        lay = synth.make_layer(100, 192, 900 + i)
        small.append({k: torch.from_numpy(lay[k]).to(device) for k in ("W", "H", "scale")})
    small.append(layers[0])  # a different shape closes the group
    for moves in (0, 4):
        backend = sdist.HipBackend(cb, "diag", 0.01, moves, with_error=True)
        seen = {"round": [], "many": []}
        run_round, many = backend.run_round, backend.factorize_many
        backend.run_round = lambda members, *a: (seen["round"].append(len(members)), run_round(members, *a))[1]
        backend.factorize_many = lambda ls: (seen["many"].append(len(ls)), many(ls))[1]
        shards = sdist.quantize_stream(small, backend)
        torch.cuda.synchronize()
        assert seen["round"] == [7] and seen["many"] == [4 - rank], (rank, seen)  # ranks 0 / 1 own 4 / 3 of the seven
        for lay, sh in zip(small, shards):
            lo, hi = sh["rows"]
            res = engine.quantize_layer(lay["W"], lay["H"], cb, lay["scale"], nb_ls_moves=moves)
            err = engine.row_errors(lay["W"], res.Q, lay["H"])
            assert np.array_equal(sh["Q"].cpu().numpy(), res.Q[lo:hi].cpu().numpy())
            assert np.array_equal(sh["idx"].cpu().numpy(), res.idx[lo:hi].cpu().numpy())
            np.testing.assert_allclose(sh["row_err"].cpu().numpy(), err[lo:hi].cpu().numpy(), rtol=1e-5)
            assert int(sh["info"].item()) == 0
    # WIDE layers with few rows (3200 columns: not "small"; 100-row shards, ragged): they join groups too -- four of them on two
    # ranks are two rounds, ONE group: each rank factors its two in one launch chain, one all-gather, one stacked loop
    wide = []
    for i in range(4):
        lay = synth.make_layer(200, 3200, 950 + i)
        wide.append({k: torch.from_numpy(lay[k]).to(device) for k in ("W", "H", "scale")})
    backend = sdist.HipBackend(cb, "diag", 0.01, 0, with_error=True)
    assert not backend.wants_local_batch(wide[0]) and backend.group_limit(wide[0], 100) >= 4
    seen = {"round": [], "many": []}
    run_round, many = backend.run_round, backend.factorize_many
    backend.run_round = lambda members, *a: (seen["round"].append(len(members)), run_round(members, *a))[1]
    backend.factorize_many = lambda ls: (seen["many"].append(len(ls)), many(ls))[1]
    shards = sdist.quantize_stream(wide, backend)
    torch.cuda.synchronize()
    assert seen["round"] == [4] and seen["many"] == [2], (rank, seen)
    for lay, sh in zip(wide, shards):
        lo, hi = sh["rows"]
        res = engine.quantize_layer(lay["W"], lay["H"], cb, lay["scale"])
        err = engine.row_errors(lay["W"], res.Q, lay["H"])
        assert np.array_equal(sh["Q"].cpu().numpy(), res.Q[lo:hi].cpu().numpy())
        assert np.array_equal(sh["idx"].cpu().numpy(), res.idx[lo:hi].cpu().numpy())
        np.testing.assert_allclose(sh["row_err"].cpu().numpy(), err[lo:hi].cpu().numpy(), rtol=1e-5)
        assert int(sh["info"].item()) == 0
    # an indefinite Hessian in a batched round: its root's status word travels in the packed factor, and EVERY rank
    # raises LinAlgError naming the layer (reference: np.linalg.cholesky, sleekit/obq.py:49-50)
    bad = [dict(lay) for lay in layers]
    bad[4]["H"] = bad[4]["H"].clone()
    bad[4]["H"][11, 11] = -3.0
    backend = sdist.HipBackend(cb, "diag", 0.01, 0, with_error=True)
    try:
        sdist.quantize_stream(bad, backend)
    except np.linalg.LinAlgError as exc:
        assert "layer 4 (512 x 1024)" in str(exc), str(exc)
    else:
        raise AssertionError(f"rank {rank}: no LinAlgError for the indefinite Hessian")
    dist.barrier()
    dist.destroy_process_group()
    print(f"DIST_ROUND_OK rank {rank}", flush=True)


if __name__ == "__main__":
    main()
