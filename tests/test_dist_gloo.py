"""World-size-2 gloo test of the row-shard / rotating-root / broadcast scheduling (no GPU).

sleekit_amd.dist is backend-agnostic: here a stand-in backend makes a layer's "factor" a
deterministic function of the layer and its "row result" a function of (factor, rows), so
the test can check on CPU that
  * every rank receives every layer's factor from the right root (l mod G),
  * the row ranges tile [0, R) exactly once for ragged R,
  * a rank's shard equals the single-process result restricted to its rows.
"""

import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sleekit_amd import dist as sdist


class FakeBackend:
    def __init__(self):
        self.factored = []

    def payload_words(self, n):
        return 1 + n + n * n

    def alloc_payload(self, words, device):
        return torch.empty(words, dtype=torch.int64)

    def pack(self, factor, words):
        order, U, info = factor
        body = torch.cat([info.long(), order, U.reshape(-1).view(torch.int64)])
        return torch.cat([body, torch.zeros(words - body.numel(), dtype=torch.int64)])

    def unpack(self, payload, n):
        return payload[1 : 1 + n].clone(), payload[1 + n : 1 + n + n * n].view(torch.float64).reshape(n, n).clone(), payload[:1].int()

    def factorize(self, layer):
        n = layer["H"].shape[0]
        self.factored.append(int(layer["id"]))
        order = torch.argsort(-layer["H"].diagonal().double(), stable=True)
        U = torch.triu(layer["H"].double() + layer["id"])
        return order, U, torch.full((1,), int(layer.get("bad", 0)), dtype=torch.int32)  # status word: 0 = positive definite

    def run_rows(self, layer, lo, hi, factor):
        order, U, _ = factor
        W = layer["W"][lo:hi]
        Q = (W[:, order].double() @ U).float()
        return dict(Q=Q, idx=None, row_err=Q.square().sum(dim=1), rows=(lo, hi))


class FakeBatchBackend(FakeBackend):
    """The same stand-in with the batched-round interface of HipBackend (run_round / can_batch)."""

    def __init__(self):
        super().__init__()
        self.rounds = []

    def can_batch(self, round_layers, lo, hi):
        return len(round_layers) >= 2 and hi > lo and all(l["W"].shape == round_layers[0]["W"].shape for l in round_layers)

    def run_round(self, round_layers, lo, hi, payloads):
        self.rounds.append([int(l["id"]) for l in round_layers])
        out = []
        for layer, payload in zip(round_layers, payloads):
            factor = self.unpack(payload, layer["H"].shape[0])
            shard = self.run_rows(layer, lo, hi, factor)
            shard["info"] = factor[2]
            out.append(shard)
        return out


def make_equal_layers():
    g = torch.Generator().manual_seed(11)
    layers = []
    for i, (R, n) in enumerate([(8, 6), (8, 6), (8, 6), (8, 6), (5, 4), (5, 4), (8, 6)]):
        A = torch.randn(n, n, generator=g)
        layers.append(dict(id=torch.tensor(i), W=torch.randn(R, n, generator=g), H=(A @ A.T).float()))
    return layers


def make_layers():
    g = torch.Generator().manual_seed(7)
    layers = []
    for i, (R, n) in enumerate([(10, 6), (7, 5), (9, 4), (5, 8), (11, 3)]):
        A = torch.randn(n, n, generator=g)
        layers.append(dict(id=torch.tensor(i), W=torch.randn(R, n, generator=g), H=(A @ A.T).float()))
    return layers


def _worker(rank, size, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=size)
    try:
        layers = make_layers()
        be = FakeBackend()
        shards = sdist.quantize_stream(layers, be)
        errs = [float(sdist.layer_error(s["row_err"], layers[i]["W"].shape[0])) for i, s in enumerate(shards)]
        # by value (NumPy): a torch tensor travels as a shared-memory handle that dies with this process
        q.put((rank, be.factored, [(s["rows"], s["Q"].numpy().copy()) for s in shards], errs))
    finally:
        dist.destroy_process_group()


def _worker_batched(rank, size, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=size)
    try:
        layers = make_equal_layers()
        be = FakeBatchBackend()
        shards = sdist.quantize_stream(layers, be)
        q.put((rank, be.rounds, [(s["rows"], s["Q"].numpy().copy(), int(s["info"].item())) for s in shards]))
    finally:
        dist.destroy_process_group()


def make_model_order_layers():
    """A miniature of BASELINE's cfg2 stream: three 'blocks' of {4 x (d x d), 4d x d, d x 4d} in model order."""
    g = torch.Generator().manual_seed(23)
    layers = []
    for block in range(3):
        for R, n in [(6, 6)] * 4 + [(12, 6), (6, 12)]:
            A = torch.randn(n, n, generator=g)
            layers.append(dict(id=torch.tensor(len(layers)), W=torch.randn(R, n, generator=g), H=(A @ A.T).float()))
    return layers


def _worker_model_order(rank, size, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=size)
    try:
        layers = make_model_order_layers()
        be = FakeBatchBackend()
        shards = sdist.quantize_stream(layers, be)
        q.put((rank, be.rounds, be.factored, [(s["rows"], s["Q"].numpy().copy(), int(s["info"].item())) for s in shards]))
    finally:
        dist.destroy_process_group()


class StatusBackend(FakeBatchBackend):
    """+ the hook quantize_stream hands every layer's factorisation status to (HipBackend.note_statuses)."""

    def __init__(self):
        super().__init__()
        self.statuses = None

    def note_statuses(self, infos, layers, defer=False):
        self.statuses = [None if i is None else int(i.item()) for i in infos]


class GroupingBackend(FakeBatchBackend):
    """+ the hooks that let quantize_stream take rounds of small layers in GROUPS (HipBackend.group_limit / factorize_many)."""

    def __init__(self, limit):
        super().__init__()
        self.limit, self.batches = limit, []

    def group_limit(self, layer, rows):
        return self.limit

    def factorize_many(self, layers):
        self.batches.append([int(lay["id"]) for lay in layers])
        return [FakeBackend.factorize(self, lay) for lay in layers]


def _worker_grouped(rank, size, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=size)
    try:
        layers = make_model_order_layers()
        be = GroupingBackend(6)
        shards = sdist.quantize_stream(layers, be)
        q.put((rank, be.rounds, be.batches, [(s["rows"], s["Q"].numpy().copy(), int(s["info"].item())) for s in shards]))
    finally:
        dist.destroy_process_group()


def _worker_verify(rank, size, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=size)
    try:
        q.put((rank, sdist.verify_exchange(make_equal_layers(), FakeBatchBackend(), max_rounds=3)))
    finally:
        dist.destroy_process_group()


def make_layers_with_a_bad_one():
    layers = make_equal_layers()
    layers[3]["bad"] = 5  # its root (rank 1 of 2) reports a failing pivot: the word travels in the packed factor
    layers[6]["bad"] = 2  # the lone last layer goes layer by layer (root: rank 0)
    return layers


def _worker_status(rank, size, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=size)
    try:
        be = StatusBackend()
        sdist.quantize_stream(make_layers_with_a_bad_one(), be)
        q.put((rank, be.statuses))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_row_ranges_tile_exactly():
    for R in (1, 2, 7, 64, 4096, 4097):
        for size in (1, 2, 3, 8):
            spans = [sdist.row_range(R, r, size) for r in range(size)]
            assert spans[0][0] == 0 and spans[-1][1] == R
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert [sdist.factor_root(l, 8) for l in range(10)] == [0, 1, 2, 3, 4, 5, 6, 7, 0, 1]


@pytest.mark.timeout(120)
def test_stream_over_gloo_world2():
    single = sdist.quantize_stream(make_layers(), FakeBackend())  # not initialised: world of one
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=90) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    layers = make_layers()
    assert got[0][1] == [0, 2, 4] and got[1][1] == [1, 3]  # rotating roots
    for l, layer in enumerate(layers):
        R = layer["W"].shape[0]
        (lo0, hi0), q0 = got[0][2][l]
        (lo1, hi1), q1 = got[1][2][l]
        assert (lo0, hi1) == (0, R) and hi0 == lo1
        assert torch.equal(torch.cat([torch.from_numpy(q0), torch.from_numpy(q1)]), single[l]["Q"])  # shards == the unsharded result
        want = float(single[l]["row_err"].double().sum() / R)
        assert abs(got[0][3][l] - want) < 1e-9 * abs(want) and got[0][3][l] == got[1][3][l]


@pytest.mark.timeout(120)
def test_batched_rounds_over_gloo_world2():
    """Rounds whose layers share a shape go through backend.run_round (one call per round, every member's payload
    from the round's all-gather); the lone last layer goes layer by layer.  Shards == rows of the unsharded result."""
    single = sdist.quantize_stream(make_equal_layers(), FakeBackend())
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_batched, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=90) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    for rank in range(2):
        assert got[rank][1] == [[0, 1], [2, 3], [4, 5]]  # three batched rounds; layer 6 is a round of one
    for l, layer in enumerate(make_equal_layers()):
        R = layer["W"].shape[0]
        (lo0, hi0), q0, i0 = got[0][2][l]
        (lo1, hi1), q1, i1 = got[1][2][l]
        assert (lo0, hi1) == (0, R) and hi0 == lo1 and i0 == 0 and i1 == 0
        assert torch.equal(torch.cat([torch.from_numpy(q0), torch.from_numpy(q1)]), single[l]["Q"])


def test_plan_rounds_buckets_a_model_order_stream():
    """BASELINE cfg2-cfg4 streams at 8 ranks: every round is of one shape, at most 8 layers, and the roots are balanced."""
    def stream(block, blocks):
        return [dict(W=torch.empty(R, n, device="meta"), H=torch.empty(n, n, device="meta")) for _ in range(blocks) for R, n in block]

    cfg2 = stream([(768, 768)] * 4 + [(3072, 768), (768, 3072)], 12)
    cfg4 = stream([(3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096)], 24)
    for layers, n_rounds in ((cfg2, 6 + 2 + 2), (cfg4, 12)):
        rounds, root = sdist.plan_rounds(layers, 8)
        assert len(rounds) == n_rounds and sorted(l for r in rounds for l in r) == list(range(len(layers)))
        for members in rounds:
            assert len(members) <= 8 and len({tuple(layers[l]["W"].shape) for l in members}) == 1
            assert len({root[l] for l in members}) == len(members)  # one layer per rank and round
        per_rank = [sum(1 for x in root if x == r) for r in range(8)]
        assert max(per_rank) - min(per_rank) <= 1, per_rank
    # a single rank keeps the stream's own order, one layer per round
    rounds, root = sdist.plan_rounds(cfg2[:7], 1)
    assert rounds == [[i] for i in range(7)] and root == [0] * 7


@pytest.mark.timeout(120)
def test_model_order_stream_over_gloo_world2():
    """A stream in MODEL order (shapes alternate) on two ranks: bucketed into rounds of one shape, every round batched
    through run_round, roots balanced, shards == rows of the unsharded result."""
    single = sdist.quantize_stream(make_model_order_layers(), FakeBackend())
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_model_order, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=90) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    square = [l for l in range(18) if l % 6 < 4]
    want_rounds = [square[i:i + 2] for i in range(0, 12, 2)] + [[4, 10], [16], [5, 11], [17]]
    for rank in range(2):
        assert got[rank][1] == [r for r in want_rounds if len(r) == 2]  # the lone leftovers go layer by layer
        assert len(got[rank][2]) == 9  # 18 factorisations, 9 each
    assert sorted(got[0][2] + got[1][2]) == list(range(18))
    for l, layer in enumerate(make_model_order_layers()):
        R = layer["W"].shape[0]
        (lo0, hi0), q0, i0 = got[0][3][l]
        (lo1, hi1), q1, i1 = got[1][3][l]
        assert (lo0, hi1) == (0, R) and hi0 == lo1 and i0 == 0 and i1 == 0
        assert torch.equal(torch.cat([torch.from_numpy(q0), torch.from_numpy(q1)]), single[l]["Q"])


@pytest.mark.timeout(120)
def test_factor_status_reaches_every_rank_over_gloo_world2():
    """A factorisation that failed on its root (reference: LinAlgError, sleekit/obq.py:49-50) is reported for THAT layer on
    every rank, through the batched-round route and the layer-by-layer one."""
    be = StatusBackend()
    sdist.quantize_stream(make_layers_with_a_bad_one(), be)
    assert be.statuses == [0, 0, 0, 5, 0, 0, 2]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_status, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=90) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert got[0][1] == [0, 0, 0, 5, 0, 0, 2] and got[1][1] == [0, 0, 0, 5, 0, 0, 2]


@pytest.mark.timeout(120)
def test_rounds_of_small_layers_in_groups_over_gloo_world2():
    """A backend that wants up to 6 layers per loop batch: the six rounds of square layers of the model-order stream go in
    two groups of three rounds (ONE all-gather with three slots per rank, ONE run_round over six layers, each rank's three
    layers of a group through factorize_many), the other shapes' two-round groups likewise; shards == the unsharded result."""
    single = sdist.quantize_stream(make_model_order_layers(), FakeBackend())
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_grouped, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=90) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    square = [l for l in range(18) if l % 6 < 4]
    for rank in range(2):
        # groups: squares 6 + 6; (12, 6): rounds [4, 10] + [16] -> one group of 3 layers; (6, 12) likewise
        assert got[rank][1] == [square[:6], square[6:], [4, 10, 16], [5, 11, 17]], got[rank][1]
        # batched factorisations: three layers of each square group; of the 3-layer groups a rank owns two layers or one
        assert [len(b) for b in got[rank][2]][:2] == [3, 3] and all(len(b) == 2 for b in got[rank][2][2:]), got[rank][2]
    for l, layer in enumerate(make_model_order_layers()):
        R = layer["W"].shape[0]
        (lo0, hi0), q0, i0 = got[0][3][l]
        (lo1, hi1), q1, i1 = got[1][3][l]
        assert (lo0, hi1) == (0, R) and hi0 == lo1 and i0 == 0 and i1 == 0
        assert torch.equal(torch.cat([torch.from_numpy(q0), torch.from_numpy(q1)]), single[l]["Q"])


@pytest.mark.timeout(120)
def test_exchange_self_check_over_gloo_world2():
    """dist.verify_exchange (what `bench.py --gpus N` prints under `rccl`): every rank's copy of every payload carries its
    root's two checksums, compared across the ranks."""
    assert sdist.verify_exchange(make_equal_layers(), FakeBatchBackend())["rounds_checked"] == 0  # one rank: nothing crosses
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_verify, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=90) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    for rank in range(2):
        res = got[rank][1]
        assert res["agree"] is True and res["rounds_checked"] == 3 and res["world_size"] == 2 and res["backend"] == "gloo"
        assert res["payload_bytes"] == 8 * (1 + 6 + 36) or res["payload_bytes"] == 8 * (1 + 4 + 16)  # the last round checked


def test_stack_views_reuses_a_batched_result():
    """dist._stack_views: slices of one batched result stack as a VIEW of it (no copy); anything else through torch.stack."""
    from sleekit_amd.dist import _stack_views

    x = torch.arange(24.0).reshape(3, 2, 4)
    v = _stack_views([x[0], x[1], x[2]])
    assert v.data_ptr() == x.data_ptr() and torch.equal(v, x)
    v = _stack_views([x[1], x[2]])
    assert v.data_ptr() == x[1].data_ptr() and torch.equal(v, x[1:])
    v = _stack_views([x[0], x[2]])  # not neighbours
    assert v.data_ptr() != x.data_ptr() and torch.equal(v, torch.stack([x[0], x[2]]))
    y = torch.arange(8.0).reshape(2, 4)
    assert torch.equal(_stack_views([x[0], y]), torch.stack([x[0], y]))  # another allocation
    assert torch.equal(_stack_views([x[:, :, 1], x[:, :, 2]]), torch.stack([x[:, :, 1], x[:, :, 2]]))  # not contiguous



class RowsGroupingBackend(GroupingBackend):
    """A group limit that depends on the shard height the way HipBackend.group_limit's does (a step at a padding boundary,
    nothing for a rank without rows): ranks whose shards differ by one row must still agree on the groups."""

    def __init__(self):
        super().__init__(0)

    def group_limit(self, layer, rows):
        return 0 if rows <= 0 else (2 if rows > 3 else 6)


def make_ragged_small_layers():
    """Six layers of 7 rows (4 + 3 on two ranks: either side of the stand-in's step) and six of ONE row (rank 1 has none)."""
    g = torch.Generator().manual_seed(31)
    layers = []
    for R, n in [(7, 5)] * 6 + [(1, 4)] * 6:
        A = torch.randn(n, n, generator=g)
        layers.append(dict(id=torch.tensor(len(layers)), W=torch.randn(R, n, generator=g), H=(A @ A.T).float()))
    return layers


def _worker_ragged_groups(rank, size, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=size)
    try:
        layers = make_ragged_small_layers()
        be = RowsGroupingBackend()
        shards = sdist.quantize_stream(layers, be)
        q.put((rank, be.rounds, [(s["rows"], s["Q"].numpy().copy(), int(s["info"].item())) for s in shards]))
    finally:
        dist.destroy_process_group()


def test_group_plan_is_the_same_on_every_rank():
    """dist._group_rounds with the REAL HipBackend.group_limit: shard heights that differ by one row across the ranks
    (R % size != 0), on either side of the 128-row padding or of group_wide_rows, and ranks without any row (R < size) --
    every rank must come to the same groups, or the ranks issue different all-gathers."""
    be = sdist.HipBackend.__new__(sdist.HipBackend)  # the limits only: no device, no streams
    from sleekit_amd import engine

    be.engine, be.act_order = engine, "diag"
    for R, n, size in ((1030, 768, 8), (4, 768, 8), (2050, 768, 8), (258, 768, 2), (2049, 4096, 8), (1025, 1024, 8), (96, 768, 8)):
        layers = [dict(W=torch.empty(R, n, device="meta"), H=torch.empty(n, n, device="meta")) for _ in range(40)]
        rounds, _ = sdist.plan_rounds(layers, size)
        plans = [sdist._group_rounds(rounds, layers, be, rank, size) for rank in range(size)]
        assert all(p == plans[0] for p in plans), (R, n, size, [len(p) for p in plans])
        assert sorted(g for grp in plans[0] for g in grp) == list(range(len(rounds)))


@pytest.mark.timeout(120)
def test_ragged_shards_agree_on_groups_over_gloo_world2():
    """R % size != 0 with a limit that steps between the two shard heights, and R < size (rank 1 holds no row of the
    one-row layers): both ranks form the same groups -- the run completes instead of hanging in mismatched all-gathers --
    and the shards are the rows of the unsharded result."""
    single = sdist.quantize_stream(make_ragged_small_layers(), FakeBackend())
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_ragged_groups, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=90) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    # 7-row layers: the tallest shard (4 rows) sets the limit of 2 for BOTH ranks -> three groups of one round each
    assert got[0][1][:3] == [[0, 1], [2, 3], [4, 5]] and got[1][1][:3] == got[0][1][:3]
    # one-row layers: limit 6 on both ranks (rank 1 asks with rank 0's height, not its own zero) -> rank 0 runs them as one batch
    assert got[0][1][3:] == [[6, 7, 8, 9, 10, 11]]
    for l, layer in enumerate(make_ragged_small_layers()):
        R = layer["W"].shape[0]
        (lo0, hi0), q0, i0 = got[0][2][l]
        (lo1, hi1), q1, i1 = got[1][2][l]
        assert (lo0, hi1) == (0, R) and hi0 == lo1 and i0 == 0 and i1 == 0
        assert torch.equal(torch.cat([torch.from_numpy(q0), torch.from_numpy(q1)]), single[l]["Q"])
