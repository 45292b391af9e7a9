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
PENDING_LIMIT = 1 << 16  # status words kept for raise_pending() (each keeps a 4-byte device tensor alive)


_gpu_seen = False
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def require_gpu():
    global _gpu_seen
    if not _gpu_seen:  # (asked once: torch.cuda.is_available() re-reads the environment on every call)
        if not torch.cuda.is_available():
            raise RuntimeError(
                "sleekit_amd needs an AMD GPU (built for gfx950 / MI355X) visible to torch; there is no CPU fallback"
            )
        _gpu_seen = True
    return torch.device("cuda", torch.cuda.current_device())


def stream_handle():
    """The current HIP stream of the current device, as the integer the C-ABI takes (called before every launch)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
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


def scratch(nbytes, tag="scratch"):
    """(tensor, bytes): grow-only scratch of at least `nbytes` per (device, stream, tag)."""
    dev = require_gpu()
    key = (dev.index, stream_handle(), tag)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
        _workspaces[key] = ws
    return ws, ws.numel()


def release_workspaces():
    """Drop the scratch tensors and the library's helper streams / events (waits for the helpers first)."""
    _workspaces.clear()
    _lib.check(_lib.lib.slk_release_helpers())


def note_info(info, what, defer=False):
    """Register the status word of a factorisation; raise now unless lazy_errors is set (or `defer`: the word is
    still in flight on another stream -- check with raise_pending() after synchronising)."""
    if lazy_errors or defer:
        _pending_info.append((info, what))
        if len(_pending_info) > PENDING_LIMIT:  # a loop that never calls raise_pending(): bounded, and it hears about it
            import warnings

            del _pending_info[: len(_pending_info) - PENDING_LIMIT]
            warnings.warn(f"sleekit_amd: more than {PENDING_LIMIT} unchecked factorisation statuses; the oldest are dropped "
                          "-- call sleekit_amd._device.raise_pending() after synchronising", RuntimeWarning, stacklevel=2)
        return
    _raise_if_failed(info, what)


HANDOFF_TIMEOUT = 0x7FFFFFFF  # SLK_INFO_HANDOFF_TIMEOUT: a workgroup of the factorisation's chain gave up waiting for another


def raise_not_pd(code, what):
    """numpy.linalg.LinAlgError, what np.linalg.cholesky raises in the reference (sleekit/obq.py:49-50)."""
    if code == HANDOFF_TIMEOUT:
        raise RuntimeError(f"{what}: the factorisation's workgroups lost one another (hand-off timed out after 2 s); results are void")
    raise np.linalg.LinAlgError(f"{what}: Matrix is not positive definite (pivot {code - 1})")


def _raise_if_failed(info, what):
    code = int(info.item())
    if code != 0:
        raise_not_pd(code, what)


def raise_pending():
    """Check every status word registered since the last call (one device -> host read for all of them)."""
    pending, _pending_info[:] = list(_pending_info), []
    if not pending:
        return
    codes = torch.cat([info.reshape(-1)[:1] for info, _ in pending]).cpu().tolist()
    for code, (_, what) in zip(codes, pending):
        if code != 0:
            raise_not_pd(code, what)

