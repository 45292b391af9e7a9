"""Device plumbing: torch-ROCm owns memory and streams, the HIP library does the work.

Functions of the public modules accept NumPy arrays (the reference's calling convention:
arrays in, new arrays out) or torch tensors already resident on the GPU (nothing leaves
HBM, nothing synchronises).
"""

import numpy as np
import torch

from . import _lib

_workspaces = {}

# When True, data-dependent failures (a non-positive pivot in the factorisation) are not
# checked after each call -- that check is a device->host read and a stream sync.  The
# pending status words can be checked later with `raise_pending()`.
lazy_errors = False
_pending_info = []


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError(
            "sleekit_amd needs an AMD GPU (built for gfx950 / MI355X) visible to torch; there is no CPU fallback"
        )
    return torch.device("cuda", torch.cuda.current_device())


def stream_handle():
    return torch.cuda.current_stream().cuda_stream


def is_device_tensor(x):
    return isinstance(x, torch.Tensor) and x.is_cuda


def to_device(x, dtype=torch.float32):
    """NumPy array / CPU tensor / device tensor -> contiguous device tensor of `dtype`."""
    dev = require_gpu()
    if isinstance(x, torch.Tensor):
        return x.to(device=dev, dtype=dtype).contiguous()
    a = np.ascontiguousarray(x)
    return torch.from_numpy(a).to(device=dev, dtype=dtype).contiguous()


def like_input(t, template):
    """Return `t` the way `template` came in: NumPy for NumPy, device tensor for device tensor."""
    if is_device_tensor(template):
        return t
    if isinstance(template, torch.Tensor):
        return t.cpu()
    return t.cpu().numpy()


def ptr(t):
    return 0 if t is None else t.data_ptr()


def workspace(R, n, batch=1):
    """(tensor, bytes): grow-only scratch per (device, stream) sized by slk_workspace_bytes(_batch)."""
    dev = require_gpu()
    if batch > 1:
        need = int(_lib.lib.slk_workspace_bytes_batch(int(batch), int(R), int(n)))
    else:
        need = int(_lib.lib.slk_workspace_bytes(int(R), int(n)))
    key = (dev.index, stream_handle())
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        _workspaces[key] = ws
    return ws, ws.numel()


def release_workspaces():
    _workspaces.clear()


def note_info(info, what):
    """Register the status word of a factorisation; raise now unless lazy_errors is set."""
    if lazy_errors:
        _pending_info.append((info, what))
        return
    _raise_if_failed(info, what)


def _raise_if_failed(info, what):
    code = int(info.item())
    if code != 0:
        raise np.linalg.LinAlgError(f"{what}: Matrix is not positive definite (pivot {code - 1})")


def raise_pending():
    pending, _pending_info[:] = list(_pending_info), []
    for info, what in pending:
        _raise_if_failed(info, what)


_queue_groups = {}


def queue_groups(n_probe=12):
    """Streams grouped by the hardware queue they landed on: [[stream, ...], ...], the default stream's group last.

    HIP multiplexes every stream of a process onto a few in-order hardware queues (GPU_MAX_HW_QUEUES, 4 by default);
    which streams share one depends on what the runtime created before (torch's pool, RCCL's own streams).  Two
    streams in one queue do not overlap, whatever the code says -- the factorisation of the next round queued behind
    this round's loops costs a factor 1.5 at N = 8 -- so the pipeline picks its streams by queue.  Probe: a ~2 ms kernel
    on stream i, a one-element kernel on every other stream; those that finish after it share its queue.
    """
    dev = require_gpu()
    if dev.index in _queue_groups:
        return _queue_groups[dev.index]
    streams = [torch.cuda.Stream(dev) for _ in range(n_probe)]
    default = torch.cuda.default_stream(dev)
    everyone = streams + [default]
    sink = torch.zeros(4096, dtype=torch.float64, device=dev)
    tick = torch.zeros(len(everyone), 8, dtype=torch.float32, device=dev)
    for j, st in enumerate(everyone):  # first use of every stream (queues are bound lazily)
        with torch.cuda.stream(st):
            tick[j].add_(1.0)
    torch.cuda.synchronize(dev)
    group_of = {}
    groups = []
    for i, si in enumerate(everyone):
        if i in group_of:
            continue
        members = [i]
        group_of[i] = len(groups)
        done = torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(si):
            _lib.check(_lib.lib.slk_probe_mfma_f64(sink.data_ptr(), 256, 20000, si.cuda_stream))
            done.record(si)
        marks = {}
        for j, sj in enumerate(everyone):
            if j in group_of:
                continue
            e = torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(sj):
                tick[j].add_(1.0)
                e.record(sj)
            marks[j] = e
        torch.cuda.synchronize(dev)
        for j, e in marks.items():
            if done.elapsed_time(e) > -0.2:  # finished after (or within 0.2 ms before the end of) the long kernel
                members.append(j)
                group_of[j] = len(groups)
        groups.append(members)
    out = [[everyone[j] for j in g if everyone[j] is not default] for g in groups if len(everyone) - 1 not in g]
    last = [[everyone[j] for j in g if everyone[j] is not default] for g in groups if len(everyone) - 1 in g]
    out = [g for g in out if g] + [g for g in last if g]
    _queue_groups[dev.index] = out
    return out
