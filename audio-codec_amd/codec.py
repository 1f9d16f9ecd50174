"""codec.py mirror: Encode / EncodeSingleChannel / getCorrectWindow with the
reference's signatures and return types (coder/codec.py:30-44, 225-380), the
arithmetic done by pacx_encode_batch on the GPU.  Scalar-mantissa path
(useVQ False, useSBR False)."""
import numpy as np

from . import _lib, context, window
from .engine import PcmView

SHORT = 256


def getCorrectWindow(lastTrans, curTrans, nextTrans, Nlong=2048):
    """coder/codec.py:30-44."""
    if curTrans:
        return lambda x: window.SineWindow(x)
    if lastTrans and nextTrans:
        return lambda x: window.StartStopWindow(x, Nlong, SHORT)
    if lastTrans:
        return lambda x: window.StopWindow(x, Nlong, SHORT)
    if nextTrans:
        return lambda x: window.StartWindow(x, Nlong, SHORT)
    return lambda x: window.SineWindow(x)


def unpack_long(enc, out, i, bands=None):
    """(scaleFactor int32[nBands], bitAlloc int64[nBands], mantissa int32[nMant], overallScale)
    of channel-frame i from the batch outputs (host arrays)."""
    bands = bands or enc.sfBands
    nb = bands.nBands
    ba = out["bit_alloc"][i, :nb].astype(np.int64)
    sf = out["scale_factor"][i, :nb].astype(np.int32)
    keep = np.repeat(ba != 0, bands.nLines)
    mant = out["mantissa"][i][keep].astype(np.int32)
    return sf, ba, mant, int(out["overall"][i, 0])


def unpack_short(enc, out, i, sb):
    bands = enc.sfBandsShort
    nb = bands.nBands
    ba = out["bit_alloc"][i, sb * nb:(sb + 1) * nb].astype(np.int64)
    sf = out["scale_factor"][i, sb * nb:(sb + 1) * nb].astype(np.int32)
    keep = np.repeat(ba != 0, bands.nLines)
    mant = out["mantissa"][i, sb * 128:(sb + 1) * 128][keep].astype(np.int32)
    return sf, ba, mant, int(out["overall"][i, sb])


def _to_host(out):
    return {k: v.cpu().numpy() for k, v in out.items() if v is not None and k != "flags"}


def Encode(data, codingParams, lastTrans=False, curTrans=False, nextTrans=False):
    """coder/codec.py:225-263.  data: list (nChannels) of float64 blocks of
    2*nMDCTLines samples.  Returns (scaleFactor, bitAlloc, mantissa,
    overallScaleFactor), each a list over channels."""
    import torch
    if getattr(codingParams, "useVQ", False) or getattr(codingParams, "useSBR", False):
        raise NotImplementedError("the GPU path covers the scalar-mantissa coder (useVQ/useSBR False)")
    enc = context.encoder_for_params(codingParams)
    n_ch = codingParams.nChannels
    n = 2 * codingParams.nMDCTLines
    flags = [(bool(lastTrans), bool(curTrans), bool(nextTrans))]
    blk = np.zeros((1, n_ch, 2048))
    if curTrans:
        if n != 256:
            raise ValueError("a short block is 256 samples (nMDCTLines 128)")
        for ch in range(n_ch):
            blk[0, ch, 448:448 + 256] = data[ch]
    else:
        if n != 2048:
            raise NotImplementedError("long blocks are 2048 samples (nMDCTLines 1024)")
        for ch in range(n_ch):
            blk[0, ch] = data[ch]
    pcm = PcmView.frames(torch.as_tensor(blk, device=enc.device))
    out = _to_host(enc.encode(pcm, flags))
    res = ([], [], [], [])
    for ch in range(n_ch):
        r = unpack_short(enc, out, ch, 0) if curTrans else unpack_long(enc, out, ch)
        for dst, v in zip(res, r):
            dst.append(v)
    return res


def EncodeSingleChannel(data, codingParams, lastTrans=False, curTrans=False, nextTrans=False):
    """coder/codec.py:266-380 for one channel."""
    one = type("P", (), {})()
    one.__dict__.update(codingParams.__dict__)
    one.nChannels = 1
    s, b, m, o = Encode([data], one, lastTrans, curTrans, nextTrans)
    return s[0], b[0], m[0], o[0]


def Decode(scaleFactor, bitAlloc, mantissa, overallScaleFactor, pb, codingParams,
           lastTrans=False, curTrans=False, nextTrans=False):
    """coder/codec.py:47-92 for one channel on the GPU (pacx_decode_batch): the
    windowed IMDCT output, 2*nMDCTLines samples, before overlap-and-add.
    `mantissa` is line-indexed, as PACFile.getDecodedBlock builds it."""
    import torch
    if getattr(codingParams, "useVQ", False):
        raise NotImplementedError("the GPU decoder handles scalar-mantissa streams (useVQ False)")
    enc = context.encoder_for_params(codingParams)
    n_lines = codingParams.nMDCTLines
    if n_lines not in (1024, 128) or bool(curTrans) != (n_lines == 128):
        raise NotImplementedError("long blocks have 1024 lines, short (curTrans) blocks 128")
    nb = len(bitAlloc)
    codes = enc.alloc_outputs(1)
    for k in ("overall", "scale_factor", "bit_alloc", "mantissa"):
        codes[k].zero_()
    dev = enc.device
    codes["flags"] = torch.tensor([int(bool(lastTrans)) | int(bool(curTrans)) << 1 | int(bool(nextTrans)) << 2],
                                  dtype=torch.uint8, device=dev)
    codes["overall"][0, 0] = int(overallScaleFactor)
    codes["scale_factor"][0, :nb] = torch.as_tensor(np.asarray(scaleFactor, dtype=np.int32), device=dev)
    codes["bit_alloc"][0, :nb] = torch.as_tensor(np.asarray(bitAlloc, dtype=np.int32), device=dev)
    codes["mantissa"][0, :n_lines] = torch.as_tensor(np.asarray(mantissa, dtype=np.int32)[:n_lines], device=dev)
    block = enc.decode(codes, 1, want_blocks=True, want_pcm=False)[0].cpu().numpy()
    return block[448:448 + 256].copy() if curTrans else block
