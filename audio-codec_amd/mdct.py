"""mdct.py mirror (coder/mdct.py:43-69): forward MDCT on the GPU for the block
sizes the codec uses (a = b = 1024 or 128)."""
import numpy as np

from . import context
from .engine import PcmView


def MDCT(data, a, b, isInverse=False):
    """Forward MDCT of an already-windowed block; returns (a+b)/2 lines."""
    import torch
    if isInverse:
        raise NotImplementedError("IMDCT belongs to the decode path (not accelerated yet)")
    if a != b or a not in (1024, 128):
        raise NotImplementedError("GPU MDCT kernels exist for a = b = 1024 and a = b = 128")
    data = np.ascontiguousarray(data, dtype=np.float64)
    if data.shape != (a + b,):
        raise ValueError("data must hold a+b samples")
    enc = context.any_encoder()
    if a == 1024:
        pcm = PcmView.frames(torch.as_tensor(data, device=enc.device).view(1, 1, 2048))
        return enc.mdct(pcm, prewindowed=True)[0].cpu().numpy()
    frame = np.zeros(2048)
    frame[448:448 + 256] = data
    pcm = PcmView.frames(torch.as_tensor(frame, device=enc.device).view(1, 1, 2048))
    return enc.mdct(pcm, short=True, prewindowed=True)[0, 0].cpu().numpy()
