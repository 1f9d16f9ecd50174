"""quantize.py mirror (coder/quantize.py): same names and defaults; the codes
are computed on the GPU (csrc/pacx_exact.h via k_misc.hip)."""
import numpy as np

from . import context


def _dev(x):
    import torch
    enc = context.any_encoder()
    return enc, torch.as_tensor(np.ascontiguousarray(x, dtype=np.float64), device=enc.device)


def vQuantizeUniform(aNumVec, nBits):
    """coder/quantize.py:61-78."""
    enc, x = _dev(np.atleast_1d(aNumVec))
    return enc.quantize_uniform(x, nBits).cpu().numpy().astype(int)


def QuantizeUniform(aNum, nBits):
    """coder/quantize.py:14-36."""
    if nBits <= 0:
        return 0
    return int(vQuantizeUniform(np.array([aNum]), nBits)[0])


def ScaleFactor(aNum, nScaleBits=3, nMantBits=5):
    """coder/quantize.py:99-125."""
    enc, x = _dev(np.array([aNum]))
    return int(enc.scale_factor(x, nScaleBits, nMantBits).cpu().numpy()[0])


def vMantissa(aNumVec, scale, nScaleBits=3, nMantBits=5):
    """coder/quantize.py:229-250."""
    enc, x = _dev(np.atleast_1d(aNumVec))
    return enc.mantissa(x, int(scale), nScaleBits, nMantBits).cpu().numpy().astype(int)
