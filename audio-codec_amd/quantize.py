"""quantize.py mirror (coder/quantize.py): every public function of the module, same names and defaults;
the codes and values are computed on the GPU (csrc/pacx_exact.h through k_quant_elem / k_dequant_elem)."""
import numpy as np

from . import context


def _dev(x):
    import torch
    enc = context.any_encoder()
    return enc, torch.as_tensor(np.ascontiguousarray(x, dtype=np.float64), device=enc.device)


def _dev_codes(c):
    import torch
    enc = context.any_encoder()
    return enc, torch.as_tensor(np.ascontiguousarray(c, dtype=np.int64), device=enc.device)


def vQuantizeUniform(aNumVec, nBits):
    """coder/quantize.py:61-78."""
    enc, x = _dev(np.atleast_1d(aNumVec))
    return enc.quantize_uniform(x, nBits).cpu().numpy().astype(int)


def QuantizeUniform(aNum, nBits):
    """coder/quantize.py:14-36."""
    if nBits <= 0:
        return 0
    return int(vQuantizeUniform(np.array([aNum]), nBits)[0])


def vDequantizeUniform(aQuantizedNumVec, nBits):
    """coder/quantize.py:82-95."""
    enc, c = _dev_codes(np.atleast_1d(aQuantizedNumVec))
    return enc.dequantize_uniform(c, nBits).cpu().numpy()


def DequantizeUniform(aQuantizedNum, nBits):
    """coder/quantize.py:40-57."""
    if nBits <= 0:
        return 0
    return float(vDequantizeUniform(np.array([aQuantizedNum]), nBits)[0])


def ScaleFactor(aNum, nScaleBits=3, nMantBits=5):
    """coder/quantize.py:99-125."""
    enc, x = _dev(np.array([aNum]))
    return int(enc.scale_factor(x, nScaleBits, nMantBits).cpu().numpy()[0])


def MantissaFP(aNum, scale, nScaleBits=3, nMantBits=5):
    """coder/quantize.py:130-150."""
    enc, x = _dev(np.array([aNum]))
    return int(enc.mantissa_fp(x, int(scale), nScaleBits, nMantBits).cpu().numpy()[0])


def DequantizeFP(scale, mantissa, nScaleBits=3, nMantBits=5):
    """coder/quantize.py:154-175."""
    enc, c = _dev_codes(np.array([mantissa]))
    return float(enc.dequantize_fp(c, int(scale), nScaleBits, nMantBits).cpu().numpy()[0])


def vMantissa(aNumVec, scale, nScaleBits=3, nMantBits=5):
    """coder/quantize.py:229-250."""
    enc, x = _dev(np.atleast_1d(aNumVec))
    return enc.mantissa(x, int(scale), nScaleBits, nMantBits).cpu().numpy().astype(int)


def Mantissa(aNum, scale, nScaleBits=3, nMantBits=5):
    """coder/quantize.py:178-197."""
    return int(vMantissa(np.array([aNum]), scale, nScaleBits, nMantBits)[0])


def vDequantize(scale, mantissaVec, nScaleBits=3, nMantBits=5):
    """coder/quantize.py:254-274."""
    enc, c = _dev_codes(np.atleast_1d(mantissaVec))
    return enc.dequantize(c, int(scale), nScaleBits, nMantBits).cpu().numpy()


def Dequantize(scale, mantissa, nScaleBits=3, nMantBits=5):
    """coder/quantize.py:200-225."""
    return float(vDequantize(scale, np.array([mantissa]), nScaleBits, nMantBits)[0])
