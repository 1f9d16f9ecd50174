"""pacfile.py mirror (coder/pacfile.py): the .pac block API in front of the GPU
path: scalar mantissas, or the gain-shape coder with or without SBR
(codingParams.useVQ / useSBR, the reference driver's own settings).

  PACFile.WriteFileHeader / WriteDataBlock / Close / Encode keep the
  reference's signatures and write the same bytes (one GPU call per block:
  drop-in, not fast);
  encode_stream() is the batched path: every hop of a stream in one
  pacx_encode_batch + pacx_pack_batch + pacx_gather_body.
"""
from struct import pack

import numpy as np

from . import _lib, codec, context
from .audiofile import AudioFile
from .engine import PcmView
from .pcmfile import codes_to_fraction
from .psychoac import AssignMDCTLinesFromFreqLimits, ScaleFactorBands

BYTESIZE = 8
_ST_VQ_UNDEFINED = 8          # include/pacx.h
_ST_MALFORMED = 32
_PARTIAL = "Only read a partial block of coded PACFile data"     # coder/pacfile.py:203-205


def omitted_bands(sfBands, factor=2):
    """coder/sbr.py:6-9."""
    return np.where(sfBands.lowerLine >= sfBands.upperLine[-1] // factor)[0]


def header_bytes(cp):
    """coder/pacfile.py:306-333 (mutates cp.numSamples exactly as the reference does)."""
    if not cp.numSamples % cp.nMDCTLines:
        cp.numSamples += cp.nMDCTLines - cp.numSamples % cp.nMDCTLines
    cp.numSamples += cp.nMDCTLines
    out = b"PAC " + pack("<LHLLHHHH", cp.sampleRate, cp.nChannels, cp.numSamples, cp.nMDCTLines,
                         cp.nScaleBits, cp.nMantSizeBits, int(cp.useSBR), int(cp.useVQ))
    cp.sfBands = ScaleFactorBands(AssignMDCTLinesFromFreqLimits(cp.nMDCTLines, cp.sampleRate))
    cp.sfBandsShort = ScaleFactorBands(AssignMDCTLinesFromFreqLimits(128, cp.sampleRate))
    cp.omittedBands = omitted_bands(cp.sfBands) if cp.useSBR else []
    out += pack("<L", cp.sfBands.nBands)
    out += pack("<" + str(cp.sfBands.nBands) + "H", *(cp.sfBands.nLines.tolist()))
    return out


def _raise_like_reference(out):
    """Scalar mantissas in an SBR file: the reference raises TypeError on the first long block
    whose omitted band receives bits (coder/codec.py:541-546 -> coder/quantize.py:73-74); the
    kernels flag those channel-blocks (PACX_ST_REF_RAISES, include/pacx.h)."""
    if out["status"].numel() and int(out["status"].max().item()) & _lib.ST_REF_RAISES:
        raise TypeError(_lib.REF_SCALAR_SBR_ERROR)


_SBR_INDEX = ("index 1024 is out of bounds for axis 0 with size 1024 (Decode_SBR, coder/codec.py:173-176: the "
              "cut lies in the lower half of the spectrum; PACX_ST_VQ_UNDEFINED)")


def _decode_scalar(enc, cp, codes, **want):
    """Scalar-mantissa blocks through PACFile.Decode's routing (coder/pacfile.py:645-668).  In an SBR file a
    long block with a coded omitted band is Decode_SBR's (scalar branch, coder/codec.py:117-134) -- a block no
    encoder of the reference writes (it raises there, _raise_like_reference) but its reader and decoder take;
    where Decode_SBR raises IndexError (band tables of rates above 48 kHz) so does this."""
    if not getattr(cp, "useSBR", False):
        return enc.decode(codes, cp.nChannels, **want)
    extra = {}
    out = enc.decode(codes, cp.nChannels, extra=extra, **want)
    if extra["status"].numel() and int(extra["status"].max().item()) & _ST_VQ_UNDEFINED:
        raise IndexError(_SBR_INDEX)
    return out


class PACFile(AudioFile):
    tag = b"PAC "

    def WriteFileHeader(self, codingParams):
        self.fp.write(header_bytes(codingParams))
        codingParams.priorBlock = [np.zeros(codingParams.nMDCTLines, dtype=np.float64)
                                   for _ in range(codingParams.nChannels)]

    def _write_block_any_size(self, data, cp, lastTrans, curTrans, nextTrans):
        """WriteDataBlock for nMDCTLines other than 1024 (scalar mantissas): the reference's own sequence
        (coder/pacfile.py:449-610) with its Encode calls going to the function-level mirrors (codec.Encode composes
        a block of any length from GPU-backed pieces; 128-line sub-blocks take the tuned kernels) and the bit
        packing (:404-447, 552-565) restated on the host.  A correctness path: the batched entry points are built
        for the driver's 1024 lines."""
        long_n = cp.nMDCTLines
        full = [np.concatenate((cp.priorBlock[ch], data[ch])) for ch in range(cp.nChannels)]
        cp.priorBlock = data
        if not curTrans:
            parts = [[p] for p in zip(*self.Encode(full, cp, lastTrans, curTrans, nextTrans))]
            bands = cp.sfBands
        else:
            short = 128                                               # coder/pacfile.py:490
            pad = long_n // 2 - short // 2
            parts = [[] for _ in range(cp.nChannels)]
            long_bands, cp.sfBands = cp.sfBands, None                 # the 128-line blocks' handle: default long layout
            cp.nSamplesPerBlock = cp.nMDCTLines = short
            try:
                for n in range(pad, 2 * long_n - short - pad, short):
                    sub = [f[n:n + 2 * short] for f in full]
                    if any(np.all(x == 0) for x in sub):
                        return                                        # the whole hop is dropped (:530-533)
                    for ch, p in enumerate(zip(*self.Encode(sub, cp, lastTrans, curTrans, nextTrans))):
                        parts[ch].append(p)
            finally:
                cp.nSamplesPerBlock = cp.nMDCTLines = long_n
                cp.sfBands = long_bands
            bands = cp.sfBandsShort
        for ch in range(cp.nChannels):
            bits = codec._Bits()
            for f in (lastTrans, curTrans, nextTrans):
                bits.put(int(bool(f)), 1)
            n_bits = 4                                                # the size rule of :552-565
            for (sf, ba, mant, ov) in parts[ch]:
                bits.put(int(ov), cp.nScaleBits)
                n_bits += cp.nScaleBits
                at = 0
                for b in range(bands.nBands):
                    a = int(ba[b])
                    bits.put(a - 1 if a else 0, cp.nMantSizeBits)
                    bits.put(int(sf[b]), cp.nScaleBits)
                    n_bits += cp.nMantSizeBits + cp.nScaleBits
                    if a:
                        for j in range(int(bands.nLines[b])):
                            bits.put(int(mant[at + j]), a)
                        at += int(bands.nLines[b])
                        n_bits += a * int(bands.nLines[b])
            n_bytes = n_bits // 8 if n_bits % 8 == 0 else n_bits // 8 + 1
            blob = bits.tobytes()
            blob = blob[:n_bytes] + b"\0" * (n_bytes - len(blob))
            self.fp.write(pack("<L", n_bytes))
            self.fp.write(blob)

    def _read_block_any_size(self, cp):
        """ReadDataBlock for nMDCTLines other than 1024 (scalar mantissas): coder/pacfile.py:177-298 with the
        fields parsed on the host and the blocks decoded through the function-level mirrors."""
        long_n = cp.nMDCTLines
        data = []
        for ch in range(cp.nChannels):
            s = self.fp.read(4)
            if not s:
                if cp.overlapAndAdd:
                    tail, cp.overlapAndAdd = cp.overlapAndAdd, 0
                    return tail
                return None
            n = int.from_bytes(s, "little") if len(s) == 4 else -1
            blob = self.fp.read(n) if n > 0 else b""
            if n < 1 or len(blob) < n:
                raise RuntimeError(_PARTIAL)
            acc, pos = int.from_bytes(blob, "big"), [0]
            total = 8 * len(blob)

            def get(width):
                if pos[0] + width > total:
                    raise RuntimeError(_PARTIAL)
                v = (acc >> (total - pos[0] - width)) & ((1 << width) - 1) if width else 0
                pos[0] += width
                return v

            def one(cur):
                bands = cp.sfBandsShort if cur else cp.sfBands
                ov = get(cp.nScaleBits)
                ba, sf = [], []
                mant = np.zeros(cp.nMDCTLines, np.int32)
                for b in range(bands.nBands):
                    a = get(cp.nMantSizeBits)
                    a = a + 1 if a else 0
                    ba.append(a)
                    sf.append(get(cp.nScaleBits))
                    if a:
                        lo = int(bands.lowerLine[b])
                        for j in range(int(bands.nLines[b])):
                            mant[lo + j] = get(a)
                if cur:
                    cp.sfBands = None                                 # the 128-line blocks' handle: default long layout
                return self.Decode(sf, ba, mant, ov, None, cp, last, cur, nxt)

            last, cur, nxt = get(1), get(1), get(1)
            if not cur:
                block = one(False)
            else:
                short = 128
                block = np.zeros(2 * long_n)
                pad = long_n // 2 - short // 2
                long_bands = cp.sfBands
                cp.nSamplesPerBlock = cp.nMDCTLines = short
                try:
                    for k in range(pad, 2 * long_n - short - pad, short):
                        block[k:k + 2 * short] += one(True)
                finally:
                    cp.nSamplesPerBlock = cp.nMDCTLines = long_n
                    cp.sfBands = long_bands
            data.append(np.add(cp.overlapAndAdd[ch], block[:long_n]))
            cp.overlapAndAdd[ch] = block[long_n:]
        return data

    def WriteDataBlock(self, data, codingParams, lastTrans=False, curTrans=False, nextTrans=False):
        """coder/pacfile.py:449-610: prior || data per channel, encode, pack, write."""
        import torch
        cp = codingParams
        if cp.nMDCTLines != 1024 and not getattr(cp, "useVQ", False) and not getattr(cp, "useSBR", False):
            return self._write_block_any_size(data, cp, lastTrans, curTrans, nextTrans)
        enc = context.encoder_for_params(cp)
        blk = np.stack([np.concatenate((cp.priorBlock[ch], data[ch])) for ch in range(cp.nChannels)])
        cp.priorBlock = data
        pcm = PcmView.frames(torch.as_tensor(blk[None], device=enc.device))
        flags = [(bool(lastTrans), bool(curTrans), bool(nextTrans))]
        if getattr(cp, "useVQ", False):
            out = enc.encode_vq(pcm, flags)
            payload, n_bytes = out["payload"], out["n_bytes"]
        else:
            out = enc.encode(pcm, flags)
            _raise_like_reference(out)
            payload, n_bytes = enc.pack(out, cp.nChannels)
        n_bytes = n_bytes.cpu().numpy()
        if not n_bytes.any():
            return                                  # hop dropped (coder/pacfile.py:530-533)
        payload = payload.cpu().numpy()
        for ch in range(cp.nChannels):
            self.fp.write(pack("<L", int(n_bytes[ch])))
            self.fp.write(payload[ch, :n_bytes[ch]].tobytes())

    def Close(self, codingParams):
        """coder/pacfile.py:612-625: one block of zeros flushes the last hop."""
        if self.fp.mode == "wb":
            self.WriteDataBlock([np.zeros(codingParams.nMDCTLines) for _ in range(codingParams.nChannels)],
                                codingParams)
        self.fp.close()

    def Encode(self, data, codingParams, lastTrans=False, curTrans=False, nextTrans=False):
        """coder/pacfile.py:627-643."""
        if getattr(codingParams, "useSBR", False) and not curTrans:
            return codec.Encode_SBR(data, codingParams, lastTrans, curTrans, nextTrans)
        return codec.Encode(data, codingParams, lastTrans, curTrans, nextTrans)

    def Decode(self, scaleFactor, bitAlloc, mantissa, overallScaleFactor, pb, codingParams,
               lastTrans=False, curTrans=False, nextTrans=False):
        """coder/pacfile.py:645-668: Decode_SBR for a long block of an SBR file that codes an
        omitted band, codec.Decode otherwise."""
        cp = codingParams
        if getattr(cp, "useSBR", False) and not curTrans and len(getattr(cp, "omittedBands", [])) and \
                np.any(np.array(bitAlloc)[np.array(cp.omittedBands)] != 0):
            return codec.Decode_SBR(scaleFactor, bitAlloc, mantissa, overallScaleFactor, pb, cp,
                                    lastTrans, curTrans, nextTrans)
        return codec.Decode(scaleFactor, bitAlloc, mantissa, overallScaleFactor, pb, cp,
                            lastTrans, curTrans, nextTrans)

    def ReadFileHeader(self):
        """coder/pacfile.py:136-175."""
        head = self.fp.read(4 + 22 + 4)
        n_bands = int.from_bytes(head[-4:], "little")
        head += self.fp.read(2 * n_bands)
        cp, _ = parse_header(head)
        cp.omittedBands = omitted_bands(cp.sfBands) if cp.useSBR else []
        cp.overlapAndAdd = [np.zeros(cp.nMDCTLines, dtype=np.float64) for _ in range(cp.nChannels)]
        return cp

    def ReadDataBlock(self, codingParams):
        """coder/pacfile.py:231-298: one hop of every channel as signed fractions
        (overlap-and-add done); at the end of the file the pending half-block once,
        then None.  Unpacking and codec.Decode run on the GPU."""
        import torch
        cp = codingParams
        if cp.nMDCTLines != 1024 and not getattr(cp, "useVQ", False) and not getattr(cp, "useSBR", False):
            return self._read_block_any_size(cp)
        enc = context.encoder_for_params(cp)
        payloads = []
        for ch in range(cp.nChannels):
            s = self.fp.read(4)
            if not s:
                if cp.overlapAndAdd:
                    tail, cp.overlapAndAdd = cp.overlapAndAdd, 0
                    return tail
                return None
            if len(s) < 4:
                raise RuntimeError(_PARTIAL)
            n = int.from_bytes(s, "little")
            if n < 1 or n > enc.payload_stride:
                raise RuntimeError(_PARTIAL + f" (record of {n} bytes)")
            blob = self.fp.read(n)
            if len(blob) < n:
                raise RuntimeError(_PARTIAL)
            payloads.append(blob)
        slot = enc.payload_stride
        buf = np.zeros((cp.nChannels, slot), dtype=np.uint8)
        for ch, blob in enumerate(payloads):
            buf[ch, :len(blob)] = np.frombuffer(blob, dtype=np.uint8)
        sizes = torch.tensor([len(b) for b in payloads], dtype=torch.int32, device=enc.device)
        if getattr(cp, "useVQ", False):
            out = enc.decode_vq(torch.as_tensor(buf, device=enc.device), sizes, cp.nChannels,
                                want_blocks=True, want_pcm=False)
            if int(out["status"].max().item()) & _ST_MALFORMED:
                raise RuntimeError(_PARTIAL)
            blocks = out["blocks"].cpu().numpy()
        else:
            codes = enc.unpack(torch.as_tensor(buf, device=enc.device), sizes)
            if int(codes["status"].max().item()) & _ST_MALFORMED:
                raise RuntimeError(_PARTIAL)
            blocks = _decode_scalar(enc, cp, codes, want_blocks=True, want_pcm=False).cpu().numpy()
        data = []
        for ch in range(cp.nChannels):
            data.append(np.add(cp.overlapAndAdd[ch], blocks[ch][:cp.nMDCTLines]))
            cp.overlapAndAdd[ch] = blocks[ch][cp.nMDCTLines:]
        return data


def stream_flags(pcm, block_switching, hop=1024):
    """(last, cur, next) for every written hop of the driver loop
    (coder/pacfile.py:717-741) plus the Close block.  pcm: int16 [n_hops*hop, nCh].
    Detector and flag shifting run on the GPU (pacx_transient_flags), as in encode_stream."""
    n_hops = len(pcm) // hop
    flags = np.zeros((n_hops + 2, 3), dtype=np.uint8)
    if block_switching and n_hops:
        enc = context.any_encoder()
        _, packed = enc.transient_flags(device_stream(enc, np.ascontiguousarray(pcm), hop), n_hops, hop)
        packed = packed.cpu().numpy()
        flags[:, 0], flags[:, 1], flags[:, 2] = packed & 1, (packed >> 1) & 1, (packed >> 2) & 1
    return flags                                          # last row (Close) stays 0,0,0


def device_stream(enc, pcm, hop=1024):
    """Planar int16 device buffer [nCh, (n_hops+3)*hop]: zeros, the hops, the
    last hop again (the driver writes it twice), zeros (Close)."""
    import torch
    n, n_ch = pcm.shape
    n_hops = n // hop
    buf = np.zeros((n_ch, (n_hops + 3) * hop), dtype=np.int16)
    buf[:, hop:hop + n] = pcm.T
    if n_hops:
        buf[:, hop + n:2 * hop + n] = pcm[n - hop:].T
    return torch.as_tensor(buf, device=enc.device)


def _encode_stream_any_size(pcm, cp, block_switching):
    """the reference's driver loop (coder/pacfile.py:716-757) over the PACFile mirror, block by block: nMDCTLines other
    than 1024"""
    import io
    from .detect_transients import parTransientDetect
    hop, n_ch = cp.nMDCTLines, cp.nChannels
    f = PACFile("<memory>")
    f.fp = io.BytesIO()
    f.fp.mode = "wb"
    f.WriteFileHeader(cp)
    look = np.zeros((n_ch, 2 * hop))
    cur = last = False
    n_hops = len(pcm) // hop
    for h in range(n_hops + 1):
        if h < n_hops:
            data = np.stack([codes_to_fraction(pcm[h * hop:(h + 1) * hop, ch]) for ch in range(n_ch)])
            look = np.concatenate((np.copy(data), look[:, hop:]), axis=1)
            nxt = bool(parTransientDetect(look)) if block_switching else False
        else:
            nxt = False
        f.WriteDataBlock([look[ch, :hop] for ch in range(n_ch)], cp, lastTrans=last, curTrans=cur, nextTrans=nxt)
        last, cur = cur, nxt
    f.WriteDataBlock([np.zeros(hop) for _ in range(n_ch)], cp)          # Close (:612-625)
    return f.fp.getvalue()


def encode_stream(pcm, sample_rate, kbps_per_channel, block_switching=False, header_samples=None,
                  n_scale_bits=4, n_mant_size_bits=12, use_vq=False, use_sbr=False, chunk_hops=None, n_lines=1024):
    """Whole-stream batched encode -> .pac bytes identical to what the
    reference's driver (coder/pacfile.py:674-757) writes for the same PCM:
    scalar mantissas by default; use_vq (+ use_sbr) selects the gain-shape
    coder, and the driver's own settings are use_vq=True,
    use_sbr=(kbps < 128), block_switching=True (:703-705).  pcm: int16
    [n, nCh], n a multiple of 1024 (see pcmfile.wav_effective_stream for real
    WAV files)."""
    from .audiofile import CodingParams
    pcm = np.ascontiguousarray(pcm)
    hop = int(n_lines)
    assert pcm.ndim == 2 and len(pcm) % hop == 0
    cp = CodingParams()
    cp.sampleRate, cp.nChannels = int(sample_rate), pcm.shape[1]
    cp.numSamples = len(pcm) if header_samples is None else int(header_samples)
    cp.nMDCTLines = cp.nSamplesPerBlock = hop
    cp.nScaleBits, cp.nMantSizeBits = n_scale_bits, n_mant_size_bits
    cp.targetBitsPerSample = kbps_per_channel / (cp.sampleRate / 1000)
    cp.useSBR, cp.useVQ = bool(use_sbr), bool(use_vq)
    if hop != 1024:
        # function level: the driver loop block by block through the mirrors (scalar mantissas; a correctness path)
        if use_vq or use_sbr:
            raise NotImplementedError("gain-shape / SBR streams: nMDCTLines 1024")
        return _encode_stream_any_size(pcm, cp, block_switching)
    head = header_bytes(cp)
    enc = context.encoder_for_params(cp)
    if chunk_hops:
        # host memory to host memory in chunks: PCM in, kernels and bodies out overlap on three streams
        # (streaming.HostStreamEncoder); same bytes as the one-batch path below
        from .streaming import HostStreamEncoder
        hs = HostStreamEncoder(enc, pcm.shape[1], int(chunk_hops), block_switching=block_switching)
        parts = [bytes(b) for b in hs.encode(pcm)]
        if hs.reference_raises():
            raise TypeError(_lib.REF_SCALAR_SBR_ERROR)
        return head + b"".join(parts)
    planar = device_stream(enc, pcm, hop)
    view = PcmView.stream(planar, hop)
    if block_switching:
        _, flags = enc.transient_flags(planar, len(pcm) // hop, hop)     # detector + flag shifting on the GPU
    else:
        flags = None
    if use_vq:
        out = enc.encode_vq(view, flags)
        payload, n_bytes = out["payload"], out["n_bytes"]
    else:
        out = enc.encode_pack(view, flags)
        _raise_like_reference(out)
        payload, n_bytes = out["payload"], out["n_bytes"]
    body, total = enc.gather_body(payload, n_bytes)
    n = int(total.item())
    return head + body[:n].cpu().numpy().tobytes()


def parse_header(data):
    """coder/pacfile.py:142-151 -> (CodingParams, header length)."""
    from struct import unpack, calcsize
    from .audiofile import CodingParams
    if data[:4] != b"PAC ":
        raise RuntimeError("Tried to read a non-PAC file into a PACFile object")
    fmt = "<LHLLHHHH"
    (sr, n_ch, n_samples, n_lines, n_scale, n_mant_size, use_sbr, use_vq) = unpack(fmt, data[4:4 + calcsize(fmt)])
    pos = 4 + calcsize(fmt)
    n_bands = unpack("<L", data[pos:pos + 4])[0]
    n_lines_band = unpack("<" + str(n_bands) + "H", data[pos + 4:pos + 4 + 2 * n_bands])
    cp = CodingParams()
    cp.sampleRate, cp.nChannels, cp.numSamples = sr, n_ch, n_samples
    cp.nMDCTLines = cp.nSamplesPerBlock = n_lines
    cp.nScaleBits, cp.nMantSizeBits = n_scale, n_mant_size
    cp.useSBR, cp.useVQ = bool(use_sbr), bool(use_vq)
    cp.sfBands = ScaleFactorBands(n_lines_band)
    cp.sfBandsShort = ScaleFactorBands(AssignMDCTLinesFromFreqLimits(128, sr))
    cp.targetBitsPerSample = 128 / (sr / 1000)            # not used by the decoder
    return cp, pos + 4 + 2 * n_bands


def record_chain(data, pos, max_record):
    """Walks the '<L nBytes' chain of a .pac body (sequential by nature) and returns the
    payload offsets and sizes.  Sizes come from the file, so they are checked before anything
    is handed to the GPU: the reference's reader raises when a block is cut short
    (coder/pacfile.py:200-205), and a record longer than any the coder writes is corrupt."""
    offs, sizes = [], []
    end = len(data)
    while pos < end:
        if pos + 4 > end:
            raise RuntimeError(_PARTIAL)
        n = int.from_bytes(data[pos:pos + 4], "little")
        if n < 1 or n > max_record or pos + 4 + n > end:
            raise RuntimeError(_PARTIAL + f" (record of {n} bytes at offset {pos})")
        offs.append(pos + 4)
        sizes.append(n)
        pos += 4 + n
    return offs, sizes


def decode_stream(data):
    """Whole .pac (bytes; scalar, gain-shape or gain-shape + SBR) -> int16 [n, nCh], batched on the GPU: what
    the reference's decode loop (coder/pacfile.py:745-757) writes as PCM."""
    import torch
    cp, pos = parse_header(data)
    if cp.nMDCTLines != 1024 and not cp.useVQ and not cp.useSBR:
        # function level: the reference's decode loop block by block through the PACFile mirror
        import io
        from .pcmfile import fraction_to_codes
        f = PACFile("<memory>")
        f.fp = io.BytesIO(bytes(data))
        cp = f.ReadFileHeader()
        out = []
        while True:
            block = f.ReadDataBlock(cp)
            if not block:
                break
            out.append(np.stack([fraction_to_codes(x) for x in block], axis=1))
        return np.concatenate(out).astype(np.int16) if out else np.zeros((0, cp.nChannels), np.int16)
    enc = context.encoder_for_params(cp)
    offs, sizes = record_chain(data, pos, enc.payload_stride)
    if len(offs) % cp.nChannels:
        raise RuntimeError(_PARTIAL)
    body = torch.frombuffer(bytearray(data) + bytearray(8), dtype=torch.uint8).to(enc.device)
    sizes_t = torch.tensor(sizes, dtype=torch.int32, device=enc.device)
    offs_t = torch.tensor(offs, dtype=torch.int64, device=enc.device)
    if cp.useVQ:
        out = enc.decode_vq(body, sizes_t, cp.nChannels, offsets=offs_t)
        st = int(out["status"].max().item()) if len(sizes) else 0
        if st & _ST_MALFORMED:
            raise RuntimeError(_PARTIAL)
        if st & _ST_VQ_UNDEFINED:
            raise RuntimeError("stream holds a gain-shape block the reference's decoder fails on "
                               "(PACX_ST_VQ_UNDEFINED)")
        return out["pcm"].cpu().numpy()
    codes = enc.unpack(body, sizes_t, offs_t)
    if len(sizes) and int(codes["status"].max().item()) & _ST_MALFORMED:
        raise RuntimeError(_PARTIAL)
    return _decode_scalar(enc, cp, codes).cpu().numpy()
