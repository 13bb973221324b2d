"""Process-wide default HIP context (one per process/device, created lazily on first use)."""
import os

import torch

from . import ops

_CTX = {}


def default_context(device=None):
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0")) if torch.cuda.is_available() and torch.cuda.device_count() > 1 else 0
        device = min(device, max(torch.cuda.device_count() - 1, 0))
    if device not in _CTX:
        _CTX[device] = ops.Context(device)
    ctx = _CTX[device]
    cur = torch.cuda.current_stream(device)
    if cur.cuda_stream != ctx.stream.cuda_stream:
        ctx.use_stream(cur)
    return ctx
