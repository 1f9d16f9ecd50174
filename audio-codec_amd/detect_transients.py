"""detect_transients.py mirror (coder/detect_transients.py:5-23) -- the block
switching caller.  Host code: one reduction per hop,
not part of the five accelerated modules."""
import numpy as np


def parTransientDetect(block, thresh=4.5, axis=1):
    """Peak-to-average detector on a [nCh, n] block."""
    mag = np.abs(block)
    peak = np.max(mag, axis=axis)
    upto = np.argmax(mag, axis=axis) + 500
    cols = np.arange(min(max(upto), np.shape(block)[axis]))
    avg = np.mean(np.abs(np.take(block, cols, axis=axis))) if len(cols) else np.mean(mag)
    if np.any(avg == 0):
        return 0
    return bool(np.any(peak / avg > thresh))


def hop_transients(frac_hops):
    """frac_hops: [n_hops, nCh, hop] signed fractions.  Returns bool[n_hops]:
    parTransientDetect of (hop || zeros), which is what the driver feeds it
    (coder/pacfile.py:728-732: the look-ahead half is always zeros)."""
    n_hops, n_ch, hop = frac_hops.shape
    out = np.zeros(n_hops, dtype=bool)
    pad = np.zeros((n_ch, hop))
    for h in range(n_hops):
        out[h] = bool(parTransientDetect(np.concatenate((frac_hops[h], pad), axis=1)))
    return out
