"""hipGraph replay of the per-layer pipeline for launch-bound shapes.

A 4096 x 4096 layer keeps the GPU busy for milliseconds per call, but a 768 x 768 one (OPT-125M's attention
projections) is ~100 launches of a few microseconds each: the Python thread that enqueues them becomes the
bottleneck (0.33 ms per layer whatever the number of streams).  Every entry point of libsleekit_amd only enqueues
on the caller's stream -- no allocation, no synchronisation -- so the whole layer (order + factor + loop + error)
can be captured ONCE into a hipGraph and replayed with a single launch.

    g = GraphedLayer(R, n, UniformCodebook(8, -1, 1))      # captures on first use
    res = g(W, H, scale)                                   # device tensors in; g.Q, g.idx, g.row_err out (static)

The graph owns static input and output buffers (a captured kernel's pointers are frozen); __call__ copies the
inputs in unless they already are those buffers.
"""

import torch

from . import _device as dev
from . import engine


class GraphedLayer:
    def __init__(self, R, n, quantizer, act_order="diag", damp=0.01, scaled=True, with_error=True, device=None, inputs=None):
        """inputs = (W, H, scale): resident device tensors the graph shall read IN PLACE (they must hold valid data
        when the graph is captured: the warm-up run uses them); default: buffers of its own, filled by __call__."""
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        self.args = (quantizer, act_order, damp)
        self.with_error = with_error
        self._own = inputs is None
        if inputs is None:
            self.W = torch.empty((R, n), dtype=torch.float32, device=device)
            self.H = torch.empty((n, n), dtype=torch.float32, device=device)
            self.scale = torch.empty(R, dtype=torch.float32, device=device) if scaled else None
        else:
            self.W, self.H, self.scale = inputs
            assert self.W.shape == (R, n) and self.H.shape == (n, n) and self.W.is_contiguous() and self.H.is_contiguous()
        self.graph = None
        self.Q = self.idx = self.row_err = self.info = None

    def _body(self):
        quantizer, act_order, damp = self.args
        # (no look-ahead inside a capture: graphs of several layers replay side by side on streams of their own)
        res = engine.quantize_layer(self.W, self.H, quantizer, self.scale, act_order, damp, lookahead=False)
        self.Q, self.idx, self.info = res.Q, res.idx, res.info
        self.row_err = engine.row_errors(self.W, res.Q, self.H) if self.with_error else None

    def capture(self):
        """Warm-up (function attributes, workspace) and capture, both on a side stream as torch requires."""
        lazy = dev.lazy_errors
        dev.lazy_errors = True  # the status word is read by the caller after a replay, not inside the capture
        noted = len(dev._pending_info)
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                if self._own:  # something harmless to factor
                    self.W.zero_()
                    self.H.zero_()
                    self.H.diagonal().fill_(1.0)
                    if self.scale is not None:
                        self.scale.fill_(1.0)
                self._body()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            dev.release_workspaces()  # the capture must own the scratch it uses
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self._body()
            dev.release_workspaces()
        finally:
            dev.lazy_errors = lazy
            del dev._pending_info[noted:]  # (the status words registered by the warm-up and the capture itself)
        return self

    def check(self):
        """Raise what the reference raises when the last replay met a non-positive pivot (reads the status word)."""
        dev._raise_if_failed(self.info, "compute_hessian_chol")

    def __call__(self, W=None, H=None, scale=None):
        if self.graph is None:
            self.capture()
        if W is not None and W.data_ptr() != self.W.data_ptr():
            self.W.copy_(W, non_blocking=True)
        if H is not None and H.data_ptr() != self.H.data_ptr():
            self.H.copy_(H, non_blocking=True)
        if scale is not None and self.scale is not None and scale.data_ptr() != self.scale.data_ptr():
            self.scale.copy_(scale, non_blocking=True)
        self.graph.replay()
        return self
