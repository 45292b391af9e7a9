"""Drop-in for `sleekit.obq`: the GPTQ/OBQ core on MI355X.

Same names, arguments, defaults and error behaviour as the reference module
(sleekit/obq.py); every function runs on the GPU through libsleekit_amd.so.  Arguments
may be NumPy arrays (results come back as new NumPy arrays, like the reference) or torch
tensors already on the GPU (results stay there; nothing synchronises).

Quantizers are `UniformCodebook` or the general `Codebook` (sleekit_amd.codebook); arbitrary Python
callables have no device form and raise NotImplementedError.
"""

import numpy as np
import torch

from . import _device as dev
from . import _lib
from . import engine
from .codebook import UniformCodebook


def random_psd_matrix(size, rank, damp=0.0):
    """Random PSD test matrix (sleekit/obq.py:4-11). Host-side test-data helper, not on the path."""
    A = np.random.randn(size, rank).astype(np.float32)
    H = A @ A.T
    dampval = damp * np.linalg.norm(H, ord=2, axis=1)
    return H + dampval * np.eye(size)


def remove_input_bias(H, input_bias):
    """H - m m^T (sleekit/obq.py:14-25)."""
    assert H.ndim == 2
    assert input_bias.ndim == 1
    assert H.shape[0] == H.shape[1]
    assert H.shape[0] == input_bias.shape[0]
    Hd, md = dev.to_device(H), dev.to_device(input_bias)
    out = torch.empty_like(Hd)
    _lib.check(_lib.lib.slk_hessian_strip_mean(dev.ptr(Hd), dev.ptr(md), Hd.shape[0], dev.ptr(out), dev.stream_handle()))
    return dev.like_input(out, H)


def remove_dead_values(H, W):
    """In place (sleekit/obq.py:28-35): dead inputs get the mean diagonal, their weights zero."""
    assert H.ndim == 2 and W.ndim == 2 and H.shape[0] == H.shape[1] == W.shape[1]
    Hd, Wd = dev.to_device(H), dev.to_device(W)
    for given, used in ((H, Hd), (W, Wd)):
        if dev.is_device_tensor(given) and used.data_ptr() != given.data_ptr():
            raise ValueError("remove_dead_values works in place: pass contiguous float32 device tensors")
    R, n = Wd.shape
    ws, ws_bytes = dev.workspace(R, n)
    _lib.check(_lib.lib.slk_hessian_patch_dead(dev.ptr(Hd), dev.ptr(Wd), R, n, dev.ptr(ws), ws_bytes, dev.stream_handle()))
    if not dev.is_device_tensor(H):
        H[...] = dev.like_input(Hd, H)
    if not dev.is_device_tensor(W):
        W[...] = dev.like_input(Wd, W)


def compute_hessian_chol(H):
    """Upper-triangular float64 U with U^T U = H^-1 (sleekit/obq.py:38-55).

    Raises numpy.linalg.LinAlgError when H is not positive definite.
    """
    assert H.ndim == 2 and H.shape[0] == H.shape[1]
    n = H.shape[0]
    Md = dev.to_device(H, torch.float64)
    ld = _lib.lib.slk_factor_ld(n)
    A = torch.empty(ld * ld, dtype=torch.float64, device=Md.device)
    U = torch.empty((n, n), dtype=torch.float64, device=Md.device)
    info = torch.empty(1, dtype=torch.int32, device=Md.device)
    ws, ws_bytes = dev.workspace(0, n)
    s = dev.stream_handle()
    _lib.check(_lib.lib.slk_factor_load(dev.ptr(Md), n, dev.ptr(A), s))
    _lib.check(_lib.lib.slk_chol_inverse_upper(dev.ptr(A), n, dev.ptr(U), dev.ptr(info), dev.ptr(ws), ws_bytes, s))
    dev.note_info(info, "compute_hessian_chol")
    return dev.like_input(U, H)


def compute_hessian_order(W, H, quantizer, act_order):
    """Column order (sleekit/obq.py:58-86). `H` is the damped Hessian; ties are broken by index."""
    mode = engine.order_mode_code(act_order)
    n = W.shape[1]
    if mode == _lib.ORDER_NONE:
        out = torch.arange(n, dtype=torch.int64, device=dev.require_gpu())
        return dev.like_input(out, W)
    Wd, Hd = dev.to_device(W), dev.to_device(H)
    miss = None
    if mode == _lib.ORDER_KEYS:
        miss = engine.order_keys(Hd, n, 0.0, act_order)
    elif mode >= _lib.ORDER_ERR:
        miss = engine.column_miss(Wd, engine.require_uniform(quantizer), mode == _lib.ORDER_SQERR)
    # damp = 0: the caller's H already carries its damping
    order, _, _ = engine.factorize_order_only(Hd, n, mode, miss)
    return dev.like_input(order, W)


def channelwise_error(W, Q, H):
    """Per-row (W-Q) H (W-Q)^T (sleekit/obq.py:89-95)."""
    Wd, Qd, Hd = dev.to_device(W), dev.to_device(Q), dev.to_device(H)
    lead = Wd.shape[:-1]
    out = engine.row_errors(Wd.reshape(-1, Wd.shape[-1]), Qd.reshape(-1, Qd.shape[-1]), Hd)
    return dev.like_input(out.reshape(lead), W)


def quantization_error(W, Q, H):
    """Mean over rows of channelwise_error (sleekit/obq.py:98-103)."""
    rows = channelwise_error(W, Q, H)
    if isinstance(rows, torch.Tensor):
        return rows.mean()
    return rows.mean()


def _quantize_opt_block(Q, E, Hinv, quantizer, min_block_size, num_blocks):
    """In place on Q and E (sleekit/obq.py:121-137): the blocked loop in the given column order."""
    cb_abi = engine.require_uniform(quantizer)
    Qd, Ud = dev.to_device(Q), dev.to_device(Hinv, torch.float64)
    out, _, Eo = engine.run_loop(Qd, None, None, Ud, cb_abi, min_block_size, num_blocks, want_idx=False, want_E=True)
    if dev.is_device_tensor(Q):
        Q.copy_(out)
    else:
        Q[...] = dev.like_input(out, Q)
    if dev.is_device_tensor(E):
        E.copy_(Eo)
    else:
        E[...] = dev.like_input(Eo, E)


def _quantize_opt_core(Q, E, Hinv, quantizer):
    """In place (sleekit/obq.py:106-118): the unblocked loop = one leaf as wide as the matrix."""
    _quantize_opt_block(Q, E, Hinv, quantizer, max(int(Q.shape[1]), 1), 1)


def quantize_opt(W, H, quantizer, act_order="diag", damp=0.01, nb_ls_moves=0, min_block_size=32, num_blocks=8):
    """GPTQ-style quantization of one layer (sleekit/obq.py:169-217).

    Returns codebook VALUES (float32) shaped like W.
    """
    assert W.ndim == 2
    assert H.ndim == 2
    assert H.shape[0] == H.shape[1]
    assert H.shape[0] == W.shape[1]
    assert min_block_size >= 1
    res = engine.quantize_layer(
        dev.to_device(W), dev.to_device(H), quantizer, None, act_order, damp, nb_ls_moves, min_block_size, num_blocks,
        want_idx=False,
    )
    return dev.like_input(res.Q, W)


def compute_gain(W, Q, H, candidates):
    """Error decrease of moving each weight alone to its candidate (sleekit/obq.py:220-231)."""
    Wd, Qd, Hd, Cd = dev.to_device(W), dev.to_device(Q), dev.to_device(H), dev.to_device(candidates)
    _, G = engine.row_errors(Wd, Qd, Hd, want_G=True)  # G = (W - Q) @ H = -(delta @ H)
    D = Cd - Qd
    gain = (-(D * D)) * Hd.diagonal() + (2.0 * G) * D
    return dev.like_input(gain, W)


def quantize_local_search(W, Q, H, quantizer, nb_moves):
    """Best-first local search (sleekit/obq.py:349-358); returns Q itself when nb_moves == 0."""
    if nb_moves == 0:
        return Q
    cb_abi = engine.require_uniform(quantizer)
    Wd, Hd = dev.to_device(W), dev.to_device(H)
    Qd = dev.to_device(Q).clone()
    engine.local_search(Wd, Qd, Hd, cb_abi, nb_moves)
    return dev.like_input(Qd, W)


class LocalSearchQuantizer:
    """Stateful search with the reference's interface (sleekit/obq.py:234-346).

    The gains live on the device between calls (slk_local_search, gains_mode): the constructor builds them from
    (W - Q) H like obq.py:259-262, `do_move()` loads them, makes one move per row and stores them -- the reference's
    incremental updates, so k calls of `do_move()` are `quantize_local_search(..., k)` bit for bit and `gain_up` /
    `gain_down` are the arrays the reference would hold.  `err` is recomputed from Q when asked (the reference
    carries it by subtracting the gains of the moves).
    """

    def __init__(self, W, Q, H, quantizer):
        assert W.ndim == 2
        assert H.ndim == 2
        assert H.shape[0] == H.shape[1]
        assert H.shape[0] == W.shape[1]
        assert Q.shape == W.shape
        self._like = W
        self._cb = engine.require_uniform(quantizer)
        self.quantizer = quantizer
        self._W, self._H = dev.to_device(W), dev.to_device(H)
        self._Q = dev.to_device(Q).clone()
        R, n = self._W.shape
        self._gains = torch.empty((R, 2, n), dtype=torch.float32, device=self._W.device)
        engine.local_search(self._W, self._Q, self._H, self._cb, 0, gains=self._gains, gains_mode=1)

    @property
    def nchannels(self):
        return self._W.shape[0]

    @property
    def W(self):
        return dev.like_input(self._W, self._like)

    @property
    def H(self):
        return dev.like_input(self._H, self._like)

    @property
    def Q(self):
        return dev.like_input(self._Q, self._like)

    @property
    def err(self):
        return dev.like_input(engine.row_errors(self._W, self._Q, self._H), self._like)

    @property
    def Q_up(self):
        return dev.like_input(self.quantizer.quantize_up(self._Q), self._like)

    @property
    def Q_down(self):
        return dev.like_input(self.quantizer.quantize_down(self._Q), self._like)

    @property
    def gain_up(self):
        return dev.like_input(self._gains[:, 0].contiguous(), self._like)

    @property
    def gain_down(self):
        return dev.like_input(self._gains[:, 1].contiguous(), self._like)

    def do_move(self):
        engine.local_search(self._W, self._Q, self._H, self._cb, 1, gains=self._gains, gains_mode=2)
