"""Raw SDR samples -> frames (the RTL-SDR/HPS glue of the reference's README.md:5).

An RTL-SDR delivers unsigned 8-bit interleaved (I, Q) samples; a frame of the classifier is 128 such pairs
(256 bytes).  `frames_from_iq_u8` turns a device-resident byte buffer into the (n,2,128) float32 tensor every
`predict` entry point takes, on the device (mdc_iq_u8_to_frames), so the host never touches the samples.
"""
from __future__ import annotations

import numpy as np

from . import _cabi

DEFAULT_SCALE = 1.0 / 127.5
HOP_FRAME = 128


def window_count(nbytes: int, hop: int = HOP_FRAME) -> int:
    """Windows of 128 (I,Q) pairs, `hop` pairs apart, that fit a capture of nbytes bytes.  hop = 128 (disjoint frames)
    keeps the strict rule of the frame format: a trailing partial frame is an error."""
    if hop < 1:
        raise ValueError("hop must be >= 1 sample pair")
    if hop == HOP_FRAME:
        if nbytes % 256:
            raise ValueError(f"{nbytes} bytes is not a whole number of 256-byte frames")
        return nbytes // 256
    if nbytes % 2:
        raise ValueError(f"{nbytes} bytes is not a whole number of (I,Q) pairs")
    pairs = nbytes // 2
    return 0 if pairs < 128 else (pairs - 128) // hop + 1


def frames_from_iq_u8(iq, scale: float = DEFAULT_SCALE, device=None, hop: int = HOP_FRAME):
    """iq: uint8 tensor/array of interleaved bytes (I0,Q0,I1,Q1,...).  Returns float32 (n,2,128) on the device:
    row 0 = (I - 127.5)*scale, row 1 = (Q - 127.5)*scale; window i starts at pair i*hop (hop = 128: disjoint
    256-byte frames, a trailing partial frame is an error; smaller hops: overlapping windows, mdc_iq_u8_windows)."""
    import torch
    t = iq if isinstance(iq, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(iq, dtype=np.uint8)))
    if t.dtype != torch.uint8:
        raise TypeError(f"iq must be uint8, got {t.dtype}")
    if not t.is_cuda:
        t = t.to(device if device is not None else "cuda:0")
    t = t.contiguous().view(-1)
    if t.data_ptr() % 2:
        t = t.clone()       # a view starting at an odd byte of a larger buffer: the ABI wants whole (I,Q) pairs
    n = window_count(t.numel(), hop)
    x = torch.empty((n, 2, 128), dtype=torch.float32, device=t.device)
    with torch.cuda.device(t.device):
        _cabi.check(_cabi.lib().mdc_iq_u8_windows(t.data_ptr() if n else None, n, int(hop), float(scale), x.data_ptr() if n else None,
                                                  torch.cuda.current_stream(t.device).cuda_stream))
    return x
