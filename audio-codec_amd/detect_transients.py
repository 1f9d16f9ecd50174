"""detect_transients.py mirror (coder/detect_transients.py:5-23) -- the block switching caller (SURVEY.md
section 8f-2).  The detector runs on the GPU: pacx_transient_detect_f64 (k_transient_f64) for a block of
signed fractions, as here; the encode path itself decides on the int16 hops of a whole stream at once
(pacx_transient_flags, k_transient + k_stream_flags -- engine.Encoder.transient_flags)."""
import numpy as np

from . import context


def parTransientDetect(block, thresh=4.5, axis=1):
    """Peak-to-average detector; block [nCh, n] (axis=1) or [n, nCh] (axis=0).  Returns 0 where the mean is
    exactly zero, else True / False, as the reference does."""
    import torch
    block = np.asarray(block, dtype=np.float64)
    if block.ndim != 2:
        raise ValueError("block must be two-dimensional")
    if axis == 0:
        block = block.T
    elif axis != 1:
        raise ValueError("axis must be 0 or 1")
    enc = context.any_encoder()
    r = int(enc.transient_detect(torch.as_tensor(np.ascontiguousarray(block)[None], device=enc.device), thresh)[0].item())
    return 0 if r == 2 else bool(r)


def hop_transients(frac_hops, thresh=4.5):
    """frac_hops: [n_hops, nCh, hop] signed fractions.  Returns bool[n_hops]: parTransientDetect of
    (hop || zeros), which is what the driver feeds it (coder/pacfile.py:728-732: the look-ahead half is always
    zeros) -- all hops in one launch."""
    import torch
    frac_hops = np.asarray(frac_hops, dtype=np.float64)
    n_hops, n_ch, hop = frac_hops.shape
    enc = context.any_encoder()
    look = np.zeros((n_hops, n_ch, 2 * hop))
    look[:, :, :hop] = frac_hops
    r = enc.transient_detect(torch.as_tensor(look, device=enc.device), thresh).cpu().numpy()
    return r == 1
