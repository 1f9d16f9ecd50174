"""codec.py mirror: Encode / Encode_SBR / EncodeSingleChannel / getCorrectWindow /
Decode with the reference's signatures and return types (coder/codec.py:30-44,
225-555), the arithmetic done on the GPU: pacx_encode_batch for scalar
mantissas, pacx_encode_vq_batch for the gain-shape coder (codingParams.useVQ),
Encode_SBR for the long blocks of an SBR file."""
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
    keep = np.repeat(ba != 0, bands.nLines)                # bands may stop short of the last line (> 48 kHz)
    mant = out["mantissa"][i][:len(keep)][keep].astype(np.int32)
    return sf, ba, mant, int(out["overall"][i, 0])


def unpack_short(enc, out, i, sb):
    bands = enc.sfBandsShort
    nb = bands.nBands
    ba = out["bit_alloc"][i, sb * nb:(sb + 1) * nb].astype(np.int64)
    sf = out["scale_factor"][i, sb * nb:(sb + 1) * nb].astype(np.int32)
    keep = np.repeat(ba != 0, bands.nLines)
    mant = out["mantissa"][i, sb * 128:sb * 128 + len(keep)][keep].astype(np.int32)
    return sf, ba, mant, int(out["overall"][i, sb])


def _to_host(out):
    return {k: v.cpu().numpy() for k, v in out.items() if v is not None and k != "flags"}


def Encode(data, codingParams, lastTrans=False, curTrans=False, nextTrans=False):
    """coder/codec.py:225-263.  data: list (nChannels) of float64 blocks of
    2*nMDCTLines samples.  Returns (scaleFactor, bitAlloc, mantissa,
    overallScaleFactor), each a list over channels.  Like the reference's, this
    function does not look at useSBR (PACFile.Encode does the routing)."""
    if getattr(codingParams, "useVQ", False):
        return _encode_vq(data, codingParams, lastTrans, curTrans, nextTrans, sbr=False)
    return _encode_scalar(data, codingParams, lastTrans, curTrans, nextTrans, sbr=False)


def _encode_scalar(data, codingParams, lastTrans, curTrans, nextTrans, sbr):
    """Scalar mantissas: EncodeSingleChannel per channel (coder/codec.py:266-380) or, sbr,
    EncodeSingleChannel_SBR's useVQ-False branch (:426-482, 529-555) -- which the reference
    can only finish while no omitted band gets bits; where it raises, so does this
    (PACX_ST_REF_RAISES, include/pacx.h).  The mantissa list holds the coded bands'
    mantissas; the reference's array is longer there (it sizes it with the one-line omitted
    bands, np.empty) and the excess is uninitialised memory."""
    import torch
    cp = codingParams
    if not sbr and cp.nMDCTLines != (128 if curTrans else 1024):
        return _encode_scalar_any_size(data, cp, lastTrans, curTrans, nextTrans)
    enc = context.encoder(cp.sampleRate, cp.targetBitsPerSample, cp.nScaleBits, cp.nMantSizeBits,
                          getattr(cp, "sfBands", None), getattr(cp, "sfBandsShort", None),
                          use_vq=False, use_sbr=bool(sbr))
    n_ch = cp.nChannels
    blk = _frame_block(data, n_ch, 2 * cp.nMDCTLines, curTrans)
    pcm = PcmView.frames(torch.as_tensor(blk, device=enc.device))
    out = _to_host(enc.encode(pcm, [(bool(lastTrans), bool(curTrans), bool(nextTrans))]))
    if int(out["status"].max()) & _lib.ST_REF_RAISES:
        raise TypeError(_lib.REF_SCALAR_SBR_ERROR)
    res = ([], [], [], [])
    for ch in range(n_ch):
        r = unpack_short(enc, out, ch, 0) if curTrans else unpack_long(enc, out, ch)
        for dst, v in zip(res, r):
            dst.append(v)
    return res


def _encode_scalar_any_size(data, cp, lastTrans, curTrans, nextTrans):
    """EncodeSingleChannel (coder/codec.py:266-380, scalar mantissas) for block lengths the batch entry points are not
    built for -- nMDCTLines = 512, say -- composed statement by statement from the module mirrors, every
    one of which runs on the GPU: window (table path), MDCT (any split a + b), ScaleFactor, CalcSMRs (k_smr_generic),
    BitAlloc (k_bitalloc_generic), vMantissa.  A function-level path: a dozen launches per band and block, not a fast one.
    The bands are codingParams.sfBands (sfBandsShort with curTrans), as in the reference."""
    from . import bitalloc, mdct, psychoac, quantize, window
    half = int(cp.nMDCTLines)
    n = 2 * half
    bands = cp.sfBandsShort if curTrans else cp.sfBands
    max_mant = min(1 << cp.nMantSizeBits, 16)
    n_eff = int(1.45 * half) if curTrans else half                     # coder/codec.py:288-299
    if lastTrans or nextTrans:
        n_eff = int(0.85 * n_eff)
    budget = cp.targetBitsPerSample * n_eff
    budget -= cp.nScaleBits * (bands.nBands + 1)
    budget -= cp.nMantSizeBits * bands.nBands
    res = ([], [], [], [])
    for ch in range(cp.nChannels):
        x = np.ascontiguousarray(data[ch], dtype=np.float64)
        if x.shape[-1] != n:
            raise ValueError("a block holds 2 * nMDCTLines samples")
        if curTrans or not (lastTrans or nextTrans):                   # getCorrectWindow, coder/codec.py:30-45
            xw = window.SineWindow(x)
        elif lastTrans and nextTrans:
            xw = window.StartStopWindow(x, n, 256)
        elif lastTrans:
            xw = window.StopWindow(x, n, 256)
        else:
            xw = window.StartWindow(x, n, 256)
        lines = np.array(mdct.MDCT(xw, half, half)[:half], dtype=np.float64)
        overall = int(quantize.ScaleFactor(float(np.max(np.abs(lines))), cp.nScaleBits))
        lines *= (1 << overall)
        smr = psychoac.CalcSMRs(x, lines, overall, cp.sampleRate, bands)
        alloc = np.asarray(bitalloc.BitAlloc(budget, max_mant, bands.nBands, bands.nLines, smr)).astype(np.int64)
        sf = np.empty(bands.nBands, dtype=np.int32)
        mant = []
        for b in range(bands.nBands):
            lo, hi = int(bands.lowerLine[b]), int(bands.upperLine[b]) + 1
            sf[b] = quantize.ScaleFactor(float(np.max(np.abs(lines[lo:hi]))), cp.nScaleBits, int(alloc[b]))
            if alloc[b]:
                mant.append(np.asarray(quantize.vMantissa(lines[lo:hi], int(sf[b]), cp.nScaleBits, int(alloc[b]))))
        mant = np.concatenate(mant).astype(np.int32) if mant else np.zeros(0, np.int32)
        for dst, v in zip(res, (sf, alloc, mant, overall)):
            dst.append(v)
    return res


def _frame_block(data, n_ch, n, curTrans):
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
    return blk


def _encode_vq(data, codingParams, lastTrans, curTrans, nextTrans, sbr):
    """useVQ flavour: (bitAlloc, indices, idx_bits, overallScale), each a list
    over channels; indices / idx_bits hold one list per coded band
    (coder/codec.py:239-246, 330-360; Encode_SBR :395-410, 493-531)."""
    import torch
    cp = codingParams
    enc = context.encoder(cp.sampleRate, cp.targetBitsPerSample, cp.nScaleBits, cp.nMantSizeBits,
                          getattr(cp, "sfBands", None), getattr(cp, "sfBandsShort", None),
                          use_vq=True, use_sbr=bool(sbr))
    n_ch = cp.nChannels
    blk = _frame_block(data, n_ch, 2 * cp.nMDCTLines, curTrans)
    pcm = PcmView.frames(torch.as_tensor(blk, device=enc.device))
    out = enc.encode_vq(pcm, [(bool(lastTrans), bool(curTrans), bool(nextTrans))], want_entries=True)
    if int(out["status"].max().item()) & _lib.ST_VQ_UNDEFINED:
        raise RuntimeError("gain-shape coder reached a case the reference cannot code either")
    ba_all = out["bit_alloc"].cpu().numpy()
    ov = out["overall"].cpu().numpy()
    ent = out["entries"].cpu().numpy()
    cnt = out["entry_count"].cpu().numpy()
    nb = (enc.sfBandsShort if curTrans else enc.sfBands).nBands
    res = ([], [], [], [])
    for ch in range(n_ch):
        ba = ba_all[ch, :nb].astype(np.int64)
        idx, bits = [], []
        for b in range(nb):
            if not ba[b]:
                continue
            n = int(cnt[ch, 0, b])
            if n > ent.shape[3]:
                raise RuntimeError("more gain-shape fields in a band than the entry buffer holds")
            words = ent[ch, 0, b, :n]
            widths = (words[:, 1] & 0xFFFFFFFF).astype(np.int64)
            if np.any(widths > 64):
                raise NotImplementedError("gain index wider than 64 bits")
            idx.append([int(np.uint64(v)) for v in words[:, 0].astype(np.uint64)])
            bits.append([int(w) for w in widths])
        for dst, v in zip(res, (ba, idx, bits, int(ov[ch, 0]))):
            dst.append(v)
    return res


def Encode_SBR(data, codingParams, lastTrans=False, curTrans=False, nextTrans=False):
    """coder/codec.py:383-423: long blocks of an SBR file.  Unlike the
    reference this does not modify codingParams.sfBands.nLines (BitAlloc_SBR's
    in-place write, coder/bitalloc.py:141-143): the handle counts the omitted
    bands as one line itself."""
    if curTrans:
        raise ValueError("Encode_SBR codes long blocks (coder/pacfile.py:639-643)")
    if not getattr(codingParams, "useVQ", False):
        return _encode_scalar(data, codingParams, lastTrans, curTrans, nextTrans, sbr=True)
    return _encode_vq(data, codingParams, lastTrans, curTrans, nextTrans, sbr=True)


def EncodeSingleChannel_SBR(data, codingParams, lastTrans=False, curTrans=False, nextTrans=False):
    """coder/codec.py:426-555 for one channel: (bitAlloc, indices, idx_bits, overallScale) with
    codingParams.useVQ, else (scaleFactor, bitAlloc, mantissa, overallScale)."""
    one = type("P", (), {})()
    one.__dict__.update(codingParams.__dict__)
    one.nChannels = 1
    r = Encode_SBR([data], one, lastTrans, curTrans, nextTrans)
    return r[0][0], r[1][0], r[2][0], r[3][0]


def EncodeSingleChannel(data, codingParams, lastTrans=False, curTrans=False, nextTrans=False):
    """coder/codec.py:266-380 for one channel."""
    one = type("P", (), {})()
    one.__dict__.update(codingParams.__dict__)
    one.nChannels = 1
    s, b, m, o = Encode([data], one, lastTrans, curTrans, nextTrans)
    return s[0], b[0], m[0], o[0]


def _decode_scalar_block(scaleFactor, bitAlloc, mantissa, overallScaleFactor, codingParams, lastTrans, curTrans,
                         nextTrans, sbr):
    """one channel-block of scalar mantissas on the GPU: pacx_decode_batch (codec.Decode) or, sbr,
    pacx_decode_sbr_batch with Decode_SBR on the block whatever its allocations"""
    import torch
    n_lines = codingParams.nMDCTLines
    if n_lines not in (1024, 128) or bool(curTrans) != (n_lines == 128):
        if sbr:
            raise NotImplementedError("Decode_SBR on the GPU: long blocks of 1024 lines")
        return _decode_scalar_any_size(scaleFactor, bitAlloc, mantissa, overallScaleFactor, codingParams, lastTrans,
                                       curTrans, nextTrans)
    enc = context.encoder_for_params(codingParams)
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
    if not sbr:
        block = enc.decode(codes, 1, want_blocks=True, want_pcm=False)[0].cpu().numpy()
        return block[448:448 + 256].copy() if curTrans else block
    extra = {}
    block = enc.decode(codes, 1, want_blocks=True, want_pcm=False, extra=extra, every_long_block=True)[0].cpu().numpy()
    if int(extra["status"][0].item()) & _lib.ST_VQ_UNDEFINED:
        raise IndexError("index 1024 is out of bounds for axis 0 with size 1024")       # coder/codec.py:173-176
    return block


def _decode_scalar_any_size(scaleFactor, bitAlloc, mantissa, overallScaleFactor, cp, lastTrans, curTrans, nextTrans):
    """codec.Decode (coder/codec.py:47-92, scalar mantissas) for block lengths the batch entry points are not built for,
    composed from the GPU-backed mirrors: vDequantize per coded band, / 2^overall, IMDCT (any split), window."""
    from . import mdct, quantize, window
    half = int(cp.nMDCTLines)
    n = 2 * half
    bands = cp.sfBandsShort if curTrans else cp.sfBands
    lines = np.zeros(half, dtype=np.float64)
    at = 0
    for b in range(bands.nBands):
        cnt = int(bands.nLines[b])
        if bitAlloc[b]:
            lines[at:at + cnt] = quantize.vDequantize(int(scaleFactor[b]), np.asarray(mantissa[at:at + cnt]), cp.nScaleBits,
                                                      int(bitAlloc[b]))
        at += cnt
    lines /= 1. * (1 << int(overallScaleFactor))
    y = np.asarray(mdct.IMDCT(lines, half, half), dtype=np.float64)
    if curTrans or not (lastTrans or nextTrans):                       # getCorrectWindow, coder/codec.py:30-45
        return window.SineWindow(y)
    if lastTrans and nextTrans:
        return window.StartStopWindow(y, n, 256)
    if lastTrans:
        return window.StopWindow(y, n, 256)
    return window.StartWindow(y, n, 256)


def Decode(scaleFactor, bitAlloc, mantissa, overallScaleFactor, pb, codingParams,
           lastTrans=False, curTrans=False, nextTrans=False):
    """coder/codec.py:47-92 for one channel on the GPU (pacx_decode_batch): the
    windowed IMDCT output, 2*nMDCTLines samples, before overlap-and-add.
    `mantissa` is line-indexed, as PACFile.getDecodedBlock builds it."""
    if getattr(codingParams, "useVQ", False):
        return _decode_vq(bitAlloc, overallScaleFactor, pb, codingParams, lastTrans, curTrans, nextTrans, sbr=False)
    return _decode_scalar_block(scaleFactor, bitAlloc, mantissa, overallScaleFactor, codingParams, lastTrans,
                                curTrans, nextTrans, sbr=False)


class _Bits:
    """MSB-first bit string under construction (the payload handed to pacx_decode_vq_batch)."""

    def __init__(self):
        self.acc, self.n = 0, 0

    def put(self, value, width):
        if width:
            self.acc = (self.acc << width) | (int(value) & ((1 << width) - 1))
            self.n += width

    def tobytes(self):
        pad = -self.n % 8
        return ((self.acc << pad)).to_bytes((self.n + pad) // 8, "big")


def _decode_vq(bitAlloc, overallScaleFactor, pb, codingParams, lastTrans, curTrans, nextTrans, sbr):
    """useVQ branch of codec.Decode (coder/codec.py:47-92) and codec.Decode_SBR (:95-222) for one
    channel.  The reference walks `pb` (its PackedBits cursor, positioned behind the
    allocations) band by band through dequantize_gain_shape; here the bits of the coded bands
    are taken from the same cursor with its own ReadBits -- exactly as many as the reference
    consumes: bitAlloc * nLines per coded band, an SBR-omitted band counting one line
    (:121-134) -- put back behind a rebuilt block header, and the block is decoded on the GPU
    (pacx_decode_vq_batch: index decoding, mid/side recombination, gains, SBR reconstruction,
    IMDCT, window).  Any object with ReadBits(nBits) serves as `pb`."""
    import torch
    cp = codingParams
    enc = context.encoder(cp.sampleRate, cp.targetBitsPerSample, cp.nScaleBits, cp.nMantSizeBits,
                          getattr(cp, "sfBands", None), getattr(cp, "sfBandsShort", None),
                          use_vq=True, use_sbr=bool(sbr))
    bands = enc.sfBandsShort if curTrans else enc.sfBands
    nb = bands.nBands
    if len(bitAlloc) != nb:
        raise ValueError("bitAlloc must hold one entry per scale factor band")
    omitted = set(int(b) for b in getattr(cp, "omittedBands", [])) if sbr else set()
    bits = _Bits()
    bits.put(int(bool(lastTrans)), 1)
    bits.put(int(bool(curTrans)), 1)
    bits.put(int(bool(nextTrans)), 1)
    n_sub = _lib.SUB if curTrans else 1
    for sub in range(n_sub):                         # a lone short block sits in sub-block 0
        first = sub == 0
        bits.put(int(overallScaleFactor) if first else 0, cp.nScaleBits)
        for b in range(nb):
            ba = int(bitAlloc[b]) if first else 0
            bits.put(ba - 1 if ba else 0, cp.nMantSizeBits)
        if first:
            for b in range(nb):
                ba = int(bitAlloc[b])
                left = ba * (1 if (b in omitted and not curTrans) else int(bands.nLines[b]))
                while left > 0:                      # the reference's own cursor advances as it would there
                    take = min(left, 24)
                    bits.put(pb.ReadBits(take), take)
                    left -= take
    raw = bits.tobytes()
    if len(raw) > enc.payload_stride:
        raise ValueError("channel-block longer than any the coder writes")
    slot = np.zeros((1, enc.payload_stride), dtype=np.uint8)
    slot[0, :len(raw)] = np.frombuffer(raw, dtype=np.uint8)
    out = enc.decode_vq(torch.as_tensor(slot, device=enc.device),
                        torch.tensor([len(raw)], dtype=torch.int32, device=enc.device), 1,
                        want_blocks=True, want_pcm=False)
    st = int(out["status"][0].item())
    if st & _lib.ST_MALFORMED:
        raise RuntimeError("Only read a partial block of coded PACFile data")
    if st & _lib.ST_VQ_UNDEFINED:
        raise RuntimeError("gain-shape block the reference's decoder fails on (PACX_ST_VQ_UNDEFINED)")
    block = out["blocks"][0].cpu().numpy()
    return block[448:448 + 256].copy() if curTrans else block


def Decode_SBR(scaleFactor, bitAlloc, mantissa, overallScaleFactor, pb, codingParams,
               lastTrans=False, curTrans=False, nextTrans=False):
    """coder/codec.py:95-222 (long block of an SBR file): dequantised lines -- an omitted band counts
    ONE line (:121-134), gain-shape coded or, with useVQ off, dequantised from the mantissa at that
    line of the line-indexed array --, spectral band replication (Gaussian-smoothed envelope,
    order-1 spline transposition, per-band scaling), IMDCT, window."""
    if curTrans:
        raise ValueError("Decode_SBR decodes long blocks (coder/pacfile.py:661-666)")
    if not getattr(codingParams, "useVQ", False):
        if not getattr(codingParams, "useSBR", False):
            raise ValueError("Decode_SBR: codingParams.useSBR is off (no omitted bands)")
        return _decode_scalar_block(scaleFactor, bitAlloc, mantissa, overallScaleFactor, codingParams, lastTrans,
                                    False, nextTrans, sbr=True)
    return _decode_vq(bitAlloc, overallScaleFactor, pb, codingParams, lastTrans, curTrans, nextTrans, sbr=True)
