"""The row-shard stream with the REAL HIP backend, two ranks sharing the one GPU (gloo transport).

RCCL refuses two ranks on one device, so the collective here is gloo on device tensors; the
scheduling, streams, events and kernels are exactly what bench.py runs under RCCL.  Each rank's
shard must equal the corresponding rows of the single-process result bit for bit.
"""

import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _worker(rank, size, port, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=size)
    try:
        from sleekit_amd import codebook, synth
        from sleekit_amd import dist as sdist

        layers = _layers(torch.device("cuda", 0))
        be = sdist.HipBackend(codebook.UniformCodebook(8, -1, 1), "diag", 0.01, 0, with_error=True)
        shards = sdist.quantize_stream(layers, be)
        torch.cuda.synchronize()
        errs = [float(sdist.layer_error(s["row_err"], layers[i]["W"].shape[0])) for i, s in enumerate(shards)]
        q.put((rank, [(s["rows"], s["idx"].cpu().numpy(), s["Q"].cpu().numpy(), int(s["info"].item())) for s in shards], errs))
    finally:
        dist.destroy_process_group()


def _layers(device):
    from sleekit_amd import synth

    out = []
    for i, (R, n) in enumerate([(96, 172), (130, 256), (64, 96), (256, 768), (50, 64)]):
        L = synth.make_layer(R, n, 3000 + i)
        out.append({k: torch.from_numpy(L[k]).to(device) for k in ("W", "H", "scale")})
    return out


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(300)
def test_two_ranks_one_gpu_match_single_process():
    import torch.multiprocessing as mp

    from sleekit_amd import codebook
    from sleekit_amd import dist as sdist

    dev = torch.device("cuda", 0)
    layers = _layers(dev)
    cb = codebook.UniformCodebook(8, -1, 1)
    single = sdist.quantize_stream(layers, sdist.HipBackend(cb, "diag", 0.01, 0, with_error=True))
    plain = sdist.quantize_stream(layers, sdist.HipBackend(cb, "diag", 0.01, 0, with_error=True, overlap=False))
    torch.cuda.synchronize()
    for a, b in zip(single, plain):  # stream overlap changes nothing
        assert torch.equal(a["idx"], b["idx"]) and torch.equal(a["Q"], b["Q"]) and torch.equal(a["row_err"], b["row_err"])

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=240) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for l, layer in enumerate(layers):
        R = layer["W"].shape[0]
        (lo0, hi0), idx0, q0, info0 = got[0][1][l]
        (lo1, hi1), idx1, q1, info1 = got[1][1][l]
        assert (lo0, hi1) == (0, R) and hi0 == lo1 and info0 == 0 and info1 == 0
        assert np.array_equal(np.concatenate([idx0, idx1]), single[l]["idx"].cpu().numpy())
        assert np.array_equal(np.concatenate([q0, q1]), single[l]["Q"].cpu().numpy())
        want = float(single[l]["row_err"].double().sum() / R)
        assert abs(got[0][2][l] - want) <= 1e-6 * abs(want)


@pytest.mark.parametrize("moves", [0, 5])
def test_small_layers_in_batched_rounds_on_one_rank(moves):
    """One rank, a model-order stream of small layers (shapes alternate, one shape ragged): sleekit_amd.dist takes them in
    rounds of one shape, factored by slk_*_batch and looped as one stack -- every layer bit-equal to its own
    engine.quantize_layer, the factors bit-equal to engine.factorize."""
    from sleekit_amd import _lib, codebook, engine, synth
    from sleekit_amd import dist as sdist

    dev = torch.device("cuda", 0)
    cb = codebook.UniformCodebook(8, -1, 1)
    shapes = [(256, 192), (100, 320), (128, 1100), (256, 192), (100, 320), (256, 192), (64, 2048), (100, 320), (48, 172), (48, 172),
              (1100, 1600), (1100, 1600), (1100, 1600)]  # the last three: wide with few rows -> factored in one launch chain, looped as a stack
    layers = []
    for i, (R, n) in enumerate(shapes):
        L = synth.make_layer(R, n, 3100 + i)
        layers.append({k: torch.from_numpy(L[k]).to(dev) for k in ("W", "H", "scale")})
    be = sdist.HipBackend(cb, "diag", 0.01, moves, with_error=True)
    calls = {"local": 0, "stacked": 0}
    run_local, run_stacked = be.run_round_local, be.run_round_stacked
    be.run_round_local = lambda members: (calls.__setitem__("local", calls["local"] + 1), run_local(members))[1]
    be.run_round_stacked = lambda members, facs: (calls.__setitem__("stacked", calls["stacked"] + 1), run_stacked(members, facs))[1]
    shards = sdist.quantize_stream(layers, be)
    torch.cuda.synchronize()
    assert calls["local"] == 3  # (256, 192) x 3, (100, 320) x 3 and (48, 172) x 2 (172 columns: padded to 192 inside the
    #                              factorisation, 48 rows: padded to a tile); 1100 and 2048 columns alone
    assert calls["stacked"] == 1
    for lay, sh in zip(layers, shards):
        res = engine.quantize_layer(lay["W"], lay["H"], cb, lay["scale"], nb_ls_moves=moves)
        err = engine.row_errors(lay["W"], res.Q, lay["H"])
        assert torch.equal(sh["Q"], res.Q) and torch.equal(sh["idx"], res.idx) and int(sh["info"].item()) == 0
        np.testing.assert_allclose(sh["row_err"].cpu().numpy(), err.cpu().numpy(), rtol=1e-5)
    # the batched factorisation alone, against the single-layer one: same order, same U bit for bit, not-PD reported per layer
    Hs = [layers[i]["H"] for i in (0, 3, 5)]
    Hs[1] = Hs[1].clone()
    Hs[1][7, 7] = -1.0
    order, U, info = engine.factorize_batch(Hs, 192, 0.01, _lib.ORDER_DIAG)
    for b, H in enumerate(Hs):
        o1, u1, i1 = engine.factorize(H, 192, 0.01, _lib.ORDER_DIAG)
        assert torch.equal(order[b], o1) and int(info[b].item()) == int(i1.item())
        if b != 1:
            assert torch.equal(U[b], u1) and int(i1.item()) == 0
    assert int(info[1].item()) > 0


def test_indefinite_hessian_in_a_batched_round_raises_naming_the_layer():
    """One indefinite H among a round of 8 small layers (one rank: factored by slk_chol_inverse_upper_batch, looped as one
    stack): quantize_stream raises numpy.linalg.LinAlgError like the reference's np.linalg.cholesky (sleekit/obq.py:49-50),
    naming the layer; lazily (bench.py's mode, join=False) the same at raise_pending()."""
    from sleekit_amd import _device as sdev
    from sleekit_amd import codebook, synth
    from sleekit_amd import dist as sdist

    dev = torch.device("cuda", 0)
    cb = codebook.UniformCodebook(8, -1, 1)
    layers = []
    for i in range(8):
        L = synth.make_layer(128, 192, 3300 + i)
        layers.append({k: torch.from_numpy(L[k]).to(dev) for k in ("W", "H", "scale")})
    layers[5]["H"] = layers[5]["H"].clone()
    layers[5]["H"][17, 17] = -1.0
    be = sdist.HipBackend(cb, "diag", 0.01, 0, with_error=True)
    calls = []
    run_local = be.run_round_local
    be.run_round_local = lambda members: (calls.append(len(members)), run_local(members))[1]
    with pytest.raises(np.linalg.LinAlgError, match=r"layer 5 \(128 x 192\)"):
        sdist.quantize_stream(layers, be)
    assert calls == [8]  # one batched round
    # wide layers (factor streams + loop stream route), the bad one second
    wide = []
    for i in range(3):
        L = synth.make_layer(64, 1600, 3320 + i)
        wide.append({k: torch.from_numpy(L[k]).to(dev) for k in ("W", "H", "scale")})
    wide[1]["H"] = wide[1]["H"].clone()
    wide[1]["H"][900, 900] = -2.0
    be2 = sdist.HipBackend(cb, "diag", 0.01, 0, with_error=True)
    be2.local_batch = 1  # layer by layer
    with pytest.raises(np.linalg.LinAlgError, match=r"layer 1 \(64 x 1600\)"):
        sdist.quantize_stream(wide, be2)
    # deferred: nothing raised by the call, everything at raise_pending()
    sdev.raise_pending()
    try:
        sdev.lazy_errors = True
        sdist.quantize_stream(layers, be, join=False)
        torch.cuda.synchronize()
        with pytest.raises(np.linalg.LinAlgError, match=r"layer 5 "):
            sdev.raise_pending()
        sdist.quantize_stream([layers[i] for i in (0, 1, 2, 3)], be, join=False)
        torch.cuda.synchronize()
        sdev.raise_pending()  # a clean stream leaves nothing behind
    finally:
        sdev.lazy_errors = False
        sdev._pending_info.clear()
