"""mdct.py mirror (coder/mdct.py): MDCTslow / MDCT / IMDCT with the reference's signatures, computed on the
GPU.  The codec's block sizes (a = b = 1024 or 128) run on the FFT kernels of the encode and decode paths
(k_mdct_long / k_mdct_short, k_imdct_long / k_imdct_short); any other split a + b -- the reference's own
self-test uses a = b = 4 and 6 (coder/mdct.py:86-107) -- on k_mdct_direct, the defining cosine sums."""
import numpy as np

from . import context
from .engine import PcmView

_FFT_SIZES = (1024, 128)


def _direct(data, a, b, inverse):
    import torch
    enc = context.any_encoder()
    n_in = (a + b) // 2 if inverse else a + b
    data = np.ascontiguousarray(data, dtype=np.float64)
    if data.shape != (n_in,):
        raise ValueError(f"data must hold {n_in} values")
    return enc.mdct_direct(torch.as_tensor(data, device=enc.device), a, b, inverse)[0].cpu().numpy()


def MDCTslow(data, a, b, isInverse=False):
    """coder/mdct.py:14-40: the defining sums (2/N on the forward transform, 2 on the inverse)."""
    return _direct(data, a, b, isInverse)


def MDCT(data, a, b, isInverse=False):
    """coder/mdct.py:43-69.  Forward: an already-windowed block of a+b samples -> (a+b)/2 lines;
    isInverse: (a+b)/2 lines -> a+b samples (unwindowed)."""
    import torch
    if a != b or a not in _FFT_SIZES:
        return _direct(data, a, b, isInverse)
    enc = context.any_encoder()
    data = np.ascontiguousarray(data, dtype=np.float64)
    if isInverse:
        if data.shape != (a,):
            raise ValueError("data must hold (a+b)/2 lines")
        if a == 1024:
            return enc.imdct(torch.as_tensor(data, device=enc.device).view(1, 1024))[0].cpu().numpy()
        rows = np.zeros((8, 128))
        rows[0] = data                              # sub-block 0 sits at samples 448..703 of the 2048-sample row
        return enc.imdct(torch.as_tensor(rows, device=enc.device).view(1, 1024), short=True)[0, 448:448 + 256].cpu().numpy()
    if data.shape != (a + b,):
        raise ValueError("data must hold a+b samples")
    if a == 1024:
        pcm = PcmView.frames(torch.as_tensor(data, device=enc.device).view(1, 1, 2048))
        return enc.mdct(pcm, prewindowed=True)[0].cpu().numpy()
    frame = np.zeros(2048)
    frame[448:448 + 256] = data
    pcm = PcmView.frames(torch.as_tensor(frame, device=enc.device).view(1, 1, 2048))
    return enc.mdct(pcm, short=True, prewindowed=True)[0, 0].cpu().numpy()


def IMDCT(data, a, b):
    """coder/mdct.py:73-77."""
    return MDCT(data, a, b, isInverse=True)
