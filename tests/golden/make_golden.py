#!/usr/bin/env python3
"""
Generates the golden vectors under tests/golden/ by RUNNING THE REFERENCE ITSELF.

Only runs in the build container (needs /root/reference).  Nothing from the
reference is copied: this script imports it, feeds it inputs and stores
inputs + outputs as .npz/.json data.  The GPU box never runs this file; it
only reads the fixtures.

    cd /tmp && MPLBACKEND=Agg python -B /root/repo/tests/golden/make_golden.py [--full]

Shims applied before use (ordinary NumPy-2 incompatibilities of the reference,
SURVEY.md section 8c): np.int/np.float aliases, writable cwd (pacfile creates
../test_debug_long_out on import), Agg backend, no bytecode.

Outputs
  tables.npz        band tables, windows, known answers of the reference's own
                    __main__ self-tests (mdct ramp, quantizer table, bitpack demo)
  stages.npz        per-stage vectors for long and short channel-frames
  excerpt_<wav>.npz 64-hop int16 excerpts of the reference's test WAVs with the
                    scalar-path .pac bytes (long-only and block-switched)
  fullfile.json     (--full) sha256 + size of whole-file scalar-path encodes
  vqfile.json/.npz  (--vq) whole-file encodes in the shipped configuration
                    (gain-shape PVQ, SBR below 128 kb/s) vs the reference's own
                    committed test_decoded_full/*.pac; per-block CRCs
  excerpt_vq_*.npz  (--vq) 24-hop excerpts in the shipped configuration
  decoded_vq_*.npz  (--vq-decoded) those excerpts through the reference's decoder
  vqwav.json        (--vq-decoded) hashes of the decoded WAVs the reference committed
  lines512.npz      (--lines512) the reference's file loop with nMDCTLines = 512, long-only and block-switched,
                    and its decoder on the result
  sbr_scalar_decode.npz (--sbr-scalar-decode) the reference's reader + Decode_SBR scalar branch on streams
                    and code sets with coded omitted bands (inputs made with the oracle, outputs the reference's)
  sbr_scalar.json   (--sbr-scalar) what the reference does with scalar mantissas + spectral band
                    replication (useVQ off, useSBR on: the branch of EncodeSingleChannel_SBR the
                    shipped driver never selects, coder/codec.py:529-555): it raises as soon as an
                    omitted band receives bits
  kbd.npz           (--kbd) window.KBDWindow tables and MDCT(KBDWindow(x)) of the six-tone
                    block, the expression at coder/bitalloc.py:161
"""
import hashlib
import io
import json
import os
import struct
import sys
import tempfile

sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")
HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

import numpy as np  # noqa: E402

np.int = int      # noqa
np.float = float  # noqa

_work = tempfile.mkdtemp(prefix="pacref_")
os.makedirs(os.path.join(_work, "run"), exist_ok=True)
os.chdir(os.path.join(_work, "run"))
sys.path.insert(0, os.path.join(REF, "coder"))

import codec        # noqa: E402
import mdct         # noqa: E402
import window       # noqa: E402
import psychoac     # noqa: E402
import bitalloc     # noqa: E402
import quantize     # noqa: E402
import bitpack      # noqa: E402
import pacfile      # noqa: E402
import pcmfile      # noqa: E402
import sbr          # noqa: E402
from audiofile import CodingParams               # noqa: E402
from detect_transients import parTransientDetect  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
SYNTH_SEED = 422


# ------------------------------------------------------------------ helpers
def read_wav(path):
    b = open(path, "rb").read()
    assert b[:4] == b"RIFF" and b[8:12] == b"WAVE"
    pos = 12
    fmt = None
    while pos < len(b):
        tag, size = b[pos:pos + 4], struct.unpack("<L", b[pos + 4:pos + 8])[0]
        if tag == b"fmt ":
            fmt = struct.unpack("<HHLLHH", b[pos + 8:pos + 24])
        if tag == b"data":
            n_ch, sr = fmt[1], fmt[2]
            pcm = np.frombuffer(b[pos + 8:pos + 8 + size], dtype="<i2")
            return sr, pcm.reshape(-1, n_ch).copy()
        pos += 8 + size
    raise RuntimeError("no data chunk")


def wav_bytes(sr, pcm):
    n, n_ch = pcm.shape
    data = pcm.astype("<i2").tobytes()
    return struct.pack("<4sL4s4sLHHLLHH4sL", b"RIFF", 36 + len(data), b"WAVE",
                       b"fmt ", 16, 1, n_ch, sr, sr * n_ch * 2, n_ch * 2, 16,
                       b"data", len(data)) + data


def ref_fraction(codes):
    """int16 -> signed fraction exactly as PCMFile.ReadDataBlock does it
    (calls the reference's own vDequantizeUniform)."""
    codes = np.asarray([int(c) for c in codes])
    signs = np.signbit(codes)
    codes[signs] *= -1
    t = quantize.vDequantizeUniform(codes, 16)
    t[signs] *= -1.
    return t


def ref_params(sr, n_ch, kbps, n_lines=1024):
    cp = CodingParams()
    cp.sampleRate, cp.nChannels = sr, n_ch
    cp.nMDCTLines = cp.nSamplesPerBlock = n_lines
    cp.nScaleBits, cp.nMantSizeBits = 4, 12
    cp.targetBitsPerSample = kbps / (sr / 1000)
    cp.useSBR, cp.useVQ = False, False
    cp.sfBands = psychoac.ScaleFactorBands(
        psychoac.AssignMDCTLinesFromFreqLimits(n_lines, sr))
    cp.sfBandsShort = psychoac.ScaleFactorBands(
        psychoac.AssignMDCTLinesFromFreqLimits(128, sr))
    cp.omittedBands = []
    return cp


def ref_encode_file(wav_path, kbps, block_switching, out_path, vq=False, sbr=None, n_lines=1024):
    """The reference's own PCMFile -> PACFile objects driven the way its
    encode_decode_test does, with the scalar mantissa path selected (or, with
    vq=True, exactly the shipped settings: useVQ, useSBR below 128 kb/s)."""
    src = pcmfile.PCMFile(wav_path)
    dst = pacfile.PACFile(out_path)
    cp = src.OpenForReading()
    cp.nMDCTLines = n_lines
    cp.nScaleBits = 4
    cp.nMantSizeBits = 12
    cp.targetBitsPerSample = kbps / (cp.sampleRate / 1000)
    cp.useSBR = bool(vq and kbps < 128) if sbr is None else bool(sbr)
    cp.useVQ = bool(vq)
    cp.nSamplesPerBlock = cp.nMDCTLines
    dst.OpenForWriting(cp)
    look = np.zeros((cp.nChannels, 2 * cp.nSamplesPerBlock))
    cur = last = False
    flags = []
    while True:
        data = src.ReadDataBlock(cp)
        if not data:
            nxt = False
        else:
            look = np.concatenate((np.copy(data), look[:, cp.nSamplesPerBlock:]),
                                  axis=1)
            nxt = parTransientDetect(look) if block_switching else False
        flags.append((int(bool(last)), int(bool(cur)), int(bool(nxt))))
        dst.WriteDataBlock(look[:, :cp.nSamplesPerBlock], cp, lastTrans=last,
                           curTrans=cur, nextTrans=nxt)
        last, cur = cur, nxt
        if not data:
            break
    src.Close(cp)
    dst.Close(cp)
    return open(out_path, "rb").read(), flags


def stage_capture(x_i16, sr, flags, n_lines, kbps=128):
    """Run every reference stage on one channel-frame given as int16."""
    last, cur, nxt = [bool(f) for f in flags]
    cp = ref_params(sr, 1, kbps)
    if cur:
        cp.nMDCTLines = cp.nSamplesPerBlock = n_lines
    bands = cp.sfBandsShort if cur else cp.sfBands
    data = ref_fraction(x_i16)
    n = 2 * n_lines
    win = codec.getCorrectWindow(last, cur, nxt, n)
    windowed = win(data)
    lines = mdct.MDCT(windowed, n_lines, n_lines)[:n_lines]
    overall = quantize.ScaleFactor(np.max(np.abs(lines)), cp.nScaleBits)
    scaled = lines * (1 << overall)
    norm = 4 / (n ** 2 * np.mean(np.hanning(n) ** 2))
    inten = norm * abs(np.fft.rfft(window.HanningWindow(data))) ** 2
    pk_f, pk_spl = psychoac.estimate_peaks(
        inten.copy(), np.fft.rfftfreq(n, d=1 / sr))
    thr = psychoac.getMaskedThreshold(data, scaled, overall, sr, bands)
    smr = psychoac.CalcSMRs(data, scaled, overall, sr, bands)
    sf, ba, mant, ov = codec.EncodeSingleChannel(data.copy(), cp, last, cur, nxt)
    assert ov == overall
    dense = np.zeros(n_lines, dtype=np.int32)
    dense[:len(mant)] = mant
    return dict(x=data, windowed=windowed, mdct=lines, overall=overall,
                inten=inten, n_peaks=len(pk_f),
                pk_f=np.array(pk_f + [0.0] * (n_lines - len(pk_f))),
                pk_spl=np.array([float(s) for s in pk_spl] +
                                [0.0] * (n_lines - len(pk_spl))),
                thr=thr, smr=smr, ba=np.asarray(ba, dtype=np.int64),
                sf=np.asarray(sf, dtype=np.int32), mant=dense, n_mant=len(mant))


def synth_stream(n_hops, n_ch=2, sr=48000, seed=SYNTH_SEED):
    """Config-2 synthetic stream: six reference tones
    (coder/psychoac.py:338-339 amplitudes/frequencies) + noise."""
    rng = np.random.default_rng(seed)
    amps = np.array([.43, .24, .15, .09, .05, .04])
    freqs = np.array([440, 550, 660, 880, 4400, 8800])
    n = np.arange(n_hops * 1024)
    out = np.zeros((n_hops * 1024, n_ch), dtype=np.int16)
    for ch in range(n_ch):
        ph = rng.uniform(0, 2 * np.pi, size=6)
        x = 0.5 * np.sum(amps[:, None] * np.cos(
            2 * np.pi * freqs[:, None] * n[None, :] / sr + ph[:, None]), axis=0)
        x = x + 0.01 * rng.standard_normal(len(n))
        out[:, ch] = np.rint(32767 * np.clip(x, -1, 1)).astype(np.int16)
    return out


# ------------------------------------------------------------------- tables
def make_tables():
    out = {}
    for n_lines in (1024, 128, 512):
        for sr in (48000, 44100):
            b = psychoac.ScaleFactorBands(
                psychoac.AssignMDCTLinesFromFreqLimits(n_lines, sr))
            key = f"bands_{n_lines}_{sr}"
            out[key + "_nLines"] = np.asarray(b.nLines, dtype=np.int64)
            out[key + "_lower"] = np.asarray(b.lowerLine, dtype=np.int64)
            out[key + "_upper"] = np.asarray(b.upperLine, dtype=np.int64)
            out[key + "_omitted"] = np.asarray(sbr.omitted_bands(b),
                                               dtype=np.int64)
    ones = np.ones(2048)
    out["win_sine_2048"] = window.SineWindow(ones)
    out["win_sine_256"] = window.SineWindow(np.ones(256))
    out["win_hann_2048"] = window.HanningWindow(ones)
    out["win_hann_256"] = window.HanningWindow(np.ones(256))
    out["win_start_2048"] = window.StartWindow(ones, 2048, 256)
    out["win_stop_2048"] = window.StopWindow(ones, 2048, 256)
    out["win_startstop_2048"] = window.StartStopWindow(ones, 2048, 256)
    for sr in (48000, 44100):
        for n_lines in (1024, 128):
            f = sr / (2 * n_lines) * (np.arange(n_lines) + 0.5)
            out[f"bark_{n_lines}_{sr}"] = psychoac.Bark(f.copy())
            out[f"thresh_{n_lines}_{sr}"] = psychoac.Thresh(f.copy())
    # the reference's own self-test inputs (coder/mdct.py:86-107)
    ramp = np.array([0, 1, 2, 3, 4, 4, 4, 4, 3, 1, -1, -3])
    ramp = np.concatenate([np.zeros(4), ramp, np.zeros(4)])
    out["mdct_ramp_in"] = ramp
    out["mdct_ramp_fast"] = mdct.MDCT(ramp, 10, 10)
    out["mdct_ramp_slow"] = mdct.MDCTslow(ramp, 10, 10)
    out["imdct_ramp_fast"] = mdct.MDCT(out["mdct_ramp_fast"], 10, 10, True)
    # quantizer table inputs (coder/quantize.py:283-286)
    q_in = np.array([-0.99, -0.39, -.08, -0.001, 0, 0.01, 0.29, 0.68, 0.99, 1.0])
    out["quant_in"] = q_in
    for bits in (8, 12):
        out[f"quant_u{bits}"] = np.array(
            [quantize.QuantizeUniform(v, bits) for v in q_in])
        out[f"quant_v{bits}"] = quantize.vQuantizeUniform(q_in, bits)
        out[f"dequant_v{bits}"] = quantize.vDequantizeUniform(
            quantize.vQuantizeUniform(q_in, bits), bits)
    out["quant_scale_3_5"] = np.array([quantize.ScaleFactor(v) for v in q_in])
    out["quant_mant_3_5"] = np.array(
        [quantize.vMantissa(np.array([v]), quantize.ScaleFactor(v))[0]
         for v in q_in])
    out["quant_deq_3_5"] = np.array(
        [quantize.vDequantize(quantize.ScaleFactor(v), quantize.vMantissa(
            np.array([v]), quantize.ScaleFactor(v)))[0] for v in q_in])
    # scale factors over a magnitude sweep at the codec's settings
    sweep = np.concatenate(([0.0, 1.0, 1.5, 0.999999], 2.0 ** -np.arange(0, 24),
                            0.75 * 2.0 ** -np.arange(0, 24)))
    out["sf_sweep_in"] = sweep
    for mb in (5, 0, 2, 7, 16):
        out[f"sf_sweep_4_{mb}"] = np.array(
            [quantize.ScaleFactor(v, 4, mb) for v in sweep])
    # bitpack demo (coder/bitpack.py:184-192)
    bp = bitpack.PackedBits()
    bp.Size(2)
    for v, w in zip((3, 5, 11, 3, 1), (4, 3, 5, 3, 1)):
        bp.WriteBits(v, w)
    out["bitpack_demo"] = np.frombuffer(bp.GetPackedData(), dtype=np.uint8)
    # all 65536 PCM codes through the reference's input conversion
    allc = np.arange(-32768, 32768)
    out["pcm_all_fraction"] = ref_fraction(allc)
    np.savez_compressed(os.path.join(HERE, "tables.npz"), **out)
    print("tables.npz", len(out), "arrays")


# ------------------------------------------------------------------- stages
def make_stages(wavs):
    longs, shorts = [], []

    def add_long(x, sr, flags, tag):
        assert len(x) == 2048
        longs.append((np.asarray(x, dtype=np.int16), sr, flags, tag))

    def add_short(x, sr, flags, tag):
        assert len(x) == 256
        shorts.append((np.asarray(x, dtype=np.int16), sr, flags, tag))

    flag_cycle = [(0, 0, 0), (0, 0, 0), (0, 0, 1), (1, 0, 0), (1, 0, 1), (0, 0, 0)]
    for name, (sr, pcm) in wavs.items():
        n_hops = -(-len(pcm) // 1024)
        pad = np.zeros((n_hops * 1024 + 1024, pcm.shape[1]), dtype=np.int16)
        pad[:len(pcm)] = pcm
        pad = np.concatenate((np.zeros((1024, pcm.shape[1]), np.int16), pad))
        # frame h = pad[h*1024 : h*1024+2048]  (hop h-1 || hop h, hop -1 = zeros)
        picks = [0, 1, 37, 38, 120, 121, 300, n_hops // 2, n_hops - 2,
                 n_hops - 1, n_hops]      # n_hops = last hop followed by zero hop
        for i, h in enumerate(picks):
            ch = i % pcm.shape[1]
            add_long(pad[h * 1024:h * 1024 + 2048, ch], sr,
                     flag_cycle[i % len(flag_cycle)], f"{name}:h{h}:c{ch}")
        # duplicated last hop (prior = last hop, new = last hop again)
        last = pad[n_hops * 1024:(n_hops + 1) * 1024, 0]
        add_long(np.concatenate((last, last)), sr, (0, 0, 0), f"{name}:dup")
        # short sub-blocks from a loud region
        loud = int(np.argmax(np.abs(pcm[:, 0].astype(np.int32))))
        base = max(0, loud - 300)
        for j in range(3):
            seg = pcm[base + j * 128: base + j * 128 + 256, j % pcm.shape[1]]
            if len(seg) == 256:
                fl = [(0, 1, 0), (1, 1, 0), (0, 1, 1)][j]
                add_short(seg, sr, fl, f"{name}:short{j}")

    # synthetic, 48 kHz
    sr = 48000
    n = np.arange(2048)
    add_long(np.zeros(2048, np.int16), sr, (0, 0, 0), "zeros")
    add_long(np.rint(32767 * np.cos(2 * np.pi * 1000 * n / sr)), sr, (0, 0, 0),
             "cos1k_fullscale")
    amps = [.43, .24, .15, .09, .05, .04]
    fr = [440, 550, 660, 880, 4400, 8800]
    six = sum(a * np.cos(2 * np.pi * f * n / sr) for a, f in zip(amps, fr))
    add_long(np.rint(32767 * six), sr, (0, 0, 0), "six_tone")
    rng = np.random.default_rng(7)
    add_long(np.rint(8000 * rng.standard_normal(2048)).clip(-32768, 32767), sr,
             (0, 0, 0), "white")
    add_long(rng.integers(-32768, 32768, 2048), sr, (0, 0, 1), "uniform_full")
    m = np.rint(3000 * rng.standard_normal(2048)).astype(np.int64)
    m[::97] = -32768
    m[5::211] = 32767
    add_long(m, sr, (0, 0, 0), "min_code")
    add_long(np.concatenate((np.zeros(1024), np.rint(32767 * six[:1024]))), sr,
             (0, 0, 0), "zero_prior")
    add_long(np.full(2048, 1234), sr, (0, 0, 0), "dc")
    add_long(np.where(n == 1500, 32767, 0), sr, (1, 0, 0), "impulse")
    add_long(np.where(n % 2 == 0, 20000, -20000), sr, (0, 0, 0), "nyquist")
    syn = synth_stream(6)
    for h in range(1, 5):
        for ch in range(2):
            add_long(syn[(h - 1) * 1024:(h + 1) * 1024, ch], sr, (0, 0, 0),
                     f"synth:h{h}:c{ch}")
    for j in range(4):
        add_short(syn[2000 + j * 128:2256 + j * 128, j % 2], sr,
                  [(0, 1, 0), (1, 1, 1), (1, 1, 0), (0, 1, 1)][j],
                  f"synth:short{j}")
    add_short(np.rint(32767 * six[:256]), sr, (0, 1, 0), "six_tone_short")
    # 96 kb/s budget variants on a few frames are covered through 'kbps'
    out = {}
    for kind, items, n_lines in (("long", longs, 1024), ("short", shorts, 128)):
        caps = []
        for (x, sr, flags, tag) in items:
            kb = 96 if tag.endswith("c1") and "h1" in tag else 128
            c = stage_capture(x, sr, flags, n_lines, kb)
            c["kbps"] = kb
            caps.append(c)
        out[f"{kind}_x_i16"] = np.stack([it[0] for it in items])
        out[f"{kind}_sr"] = np.array([it[1] for it in items])
        out[f"{kind}_flags"] = np.array([it[2] for it in items], dtype=np.uint8)
        out[f"{kind}_tag"] = np.array([it[3] for it in items])
        out[f"{kind}_kbps"] = np.array([c["kbps"] for c in caps])
        nb = max(len(c["smr"]) for c in caps)
        for key in ("x", "windowed", "mdct", "inten", "thr", "pk_f", "pk_spl",
                    "mant"):
            out[f"{kind}_{key}"] = np.stack([c[key] for c in caps])
        for key, dt in (("smr", np.float64), ("ba", np.int64), ("sf", np.int32)):
            arr = np.zeros((len(caps), nb), dtype=dt)
            for i, c in enumerate(caps):
                arr[i, :len(c[key])] = c[key]
            out[f"{kind}_{key}"] = arr
        out[f"{kind}_nbands"] = np.array([len(c["smr"]) for c in caps])
        out[f"{kind}_overall"] = np.array([c["overall"] for c in caps])
        out[f"{kind}_n_peaks"] = np.array([c["n_peaks"] for c in caps])
        out[f"{kind}_n_mant"] = np.array([c["n_mant"] for c in caps])
        print(kind, len(caps), "channel-frames")
    np.savez_compressed(os.path.join(HERE, "stages.npz"), **out)


# ----------------------------------------------------------------- excerpts
EXCERPT_HOPS = 64


def pick_start(name, pcm):
    """Start hop of the excerpt: for castanet start just before the first
    burst of transients so block switching is exercised."""
    if name != "castanet":
        return {"harpsichord": 40, "quar48_1": 100, "spmg": 200}[name]
    look = np.zeros((pcm.shape[1], 2048))
    for h in range(len(pcm) // 1024):
        data = np.array([ref_fraction(pcm[h * 1024:(h + 1) * 1024, c])
                         for c in range(pcm.shape[1])])
        look = np.concatenate((data, look[:, 1024:]), axis=1)
        if parTransientDetect(look):
            return max(0, h - 8)
    return 0


def make_excerpts(wavs):
    for name, (sr, pcm) in wavs.items():
        h0 = pick_start(name, pcm)
        ex = pcm[h0 * 1024:(h0 + EXCERPT_HOPS) * 1024 - (137 if name == "spmg" else 0)]
        path = os.path.join(_work, f"{name}_ex.wav")
        open(path, "wb").write(wav_bytes(sr, ex))
        res = {"pcm": ex, "sr": np.array(sr), "start_hop": np.array(h0)}
        for tag, bs in (("long", False), ("bs", True)):
            pac, flags = ref_encode_file(path, 128, bs,
                                         os.path.join(_work, f"{name}_{tag}.pac"))
            res[f"pac_{tag}"] = np.frombuffer(pac, dtype=np.uint8)
            res[f"flags_{tag}"] = np.array(flags, dtype=np.uint8)
            print(name, tag, len(pac), "bytes; cur-transient hops:",
                  int(np.sum(res[f"flags_{tag}"][:, 1])))
        pac96, _ = ref_encode_file(path, 96, False,
                                   os.path.join(_work, f"{name}_96.pac"))
        res["pac_long96"] = np.frombuffer(pac96, dtype=np.uint8)
        np.savez_compressed(os.path.join(HERE, f"excerpt_{name}.npz"), **res)


def ref_decode_file(pac_path, wav_path):
    """The reference's own PACFile -> PCMFile decode loop (coder/pacfile.py:745-757).
    Shim: PackedBits keeps `bytes` (NumPy-2 uint8 overflow in ReadBits)."""
    def keep_bytes(self, data):
        self.nBytes = len(data)
        self.data = bytes(data)
    bitpack.PackedBits.SetPackedData = keep_bytes
    src = pacfile.PACFile(pac_path)
    dst = pcmfile.PCMFile(wav_path)
    cp = src.OpenForReading()
    cp.bitsPerSample = 16
    dst.OpenForWriting(cp)
    while True:
        data = src.ReadDataBlock(cp)
        if not data:
            break
        dst.WriteDataBlock(data, cp)
    src.Close(cp)
    dst.Close(cp)
    raw = open(wav_path, "rb").read()
    return np.frombuffer(raw[44:], dtype="<i2").reshape(-1, cp.nChannels).copy()


def make_decoded():
    """Decode the excerpt .pac goldens with the reference's decoder."""
    for name in ["castanet", "harpsichord", "quar48_1", "spmg"]:
        ex = np.load(os.path.join(HERE, f"excerpt_{name}.npz"))
        res = {}
        for tag in ("long", "bs"):
            pac = os.path.join(_work, f"dec_{name}_{tag}.pac")
            open(pac, "wb").write(bytes(ex[f"pac_{tag}"]))
            res[f"pcm_{tag}"] = ref_decode_file(pac, os.path.join(_work, f"dec_{name}_{tag}.wav"))
            print(name, tag, res[f"pcm_{tag}"].shape)
        np.savez_compressed(os.path.join(HERE, f"decoded_{name}.npz"), **res)


def make_vq_decoded():
    """decoded_vq_<wav>.npz: the VQ excerpt goldens through the reference's own
    decoder; vqwav.json: sha256 of the PCM in the decoded WAVs the reference
    committed next to its .pac files (test_decoded_full/<wav>_<rate>.wav)."""
    sys.setrecursionlimit(12000)
    for name in ["castanet", "harpsichord", "quar48_1", "spmg"]:
        ex = np.load(os.path.join(HERE, f"excerpt_vq_{name}.npz"))
        res = {}
        for kbps in (128, 96):
            pac = os.path.join(_work, f"decvq_{name}_{kbps}.pac")
            open(pac, "wb").write(bytes(ex[f"pac_vq{kbps}"]))
            res[f"pcm_vq{kbps}"] = ref_decode_file(pac, os.path.join(_work, f"decvq_{name}_{kbps}.wav"))
            print(name, kbps, res[f"pcm_vq{kbps}"].shape)
        np.savez_compressed(os.path.join(HERE, f"decoded_vq_{name}.npz"), **res)
    rec = {}
    for name in ["castanet", "harpsichord", "quar48_1", "spmg"]:
        for kbps in (128, 96):
            raw = open(os.path.join(REF, "test_decoded_full", f"{name}_{kbps}.wav"), "rb").read()
            at = raw.index(b"data")
            n = struct.unpack("<L", raw[at + 4:at + 8])[0]
            pcm = raw[at + 8:at + 8 + n]
            rec[f"{name}:{kbps}"] = dict(pcm_sha256=hashlib.sha256(pcm).hexdigest(),
                                        n_samples=len(pcm) // 4)
    json.dump(rec, open(os.path.join(HERE, "vqwav.json"), "w"), indent=1, sort_keys=True)


def _full_one(args):
    name, bs = args
    path = os.path.join(REF, "test_signals", name + ".wav")
    out = os.path.join(_work, f"{name}_full_{int(bs)}.pac")
    pac, flags = ref_encode_file(path, 128, bs, out)
    return (name, bs, hashlib.sha256(pac).hexdigest(), len(pac),
            int(sum(f[1] for f in flags)), len(flags))


def make_full_inputs(names):
    """The PCM the reference's driver feeds the coder for each whole test WAV
    (data chunk + the bytes it reads past it, zero padded to a hop): the input
    side of fullfile.json, so the GPU box can reproduce the whole-file hashes."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import pac_oracle as po
    for n in names:
        raw = open(os.path.join(REF, "test_signals", n + ".wav"), "rb").read()
        sr, pcm, declared = po.wav_effective_stream(raw)
        np.savez_compressed(os.path.join(HERE, f"full_{n}.npz"), pcm=pcm, sr=np.array(sr),
                            declared=np.array(declared))


def _vq_one(args):
    """One whole-file run of the reference's shipped configuration (useVQ,
    useSBR below 128 kb/s, block switching) + the per-channel-block CRCs the
    GPU tests use to compare block by block."""
    import zlib
    name, kbps = args
    sys.setrecursionlimit(12000)
    path = os.path.join(REF, "test_signals", name + ".wav")
    out = os.path.join(_work, f"{name}_vq_{kbps}.pac")
    pac, flags = ref_encode_file(path, kbps, True, out, vq=True)
    committed = open(os.path.join(REF, "test_decoded_full",
                                  f"{name}_coded_{kbps}.pac"), "rb").read()
    pos = 4 + struct.calcsize("<LHLLHHHH")
    pos += 4 + 2 * struct.unpack("<L", pac[pos:pos + 4])[0]
    crcs, differs, blk = [], [], 0
    while pos < len(pac):
        n = struct.unpack("<L", pac[pos:pos + 4])[0]
        crcs.append(zlib.crc32(pac[pos:pos + 4 + n]))
        if len(committed) == len(pac) and committed[pos:pos + 4 + n] != pac[pos:pos + 4 + n]:
            differs.append(blk)
        pos += 4 + n
        blk += 1
    return (name, kbps, hashlib.sha256(pac).hexdigest(), len(pac),
            hashlib.sha256(committed).hexdigest(), len(committed),
            np.array(crcs, dtype=np.uint32), differs)


def make_vq(names):
    """vqfile.json / vqfile.npz: whole-file outputs of the reference run HERE in
    its shipped configuration, next to the hashes of the .pac files the
    reference itself committed under test_decoded_full/ (made by its author in
    2020).  Where the two differ the blocks are listed: they are the
    rounding-noise-decided DC sub-blocks of quar48_1."""
    from multiprocessing import Pool
    jobs = [(n, r) for n in names for r in (128, 96)]
    with Pool(8) as pool:
        rows = pool.map(_vq_one, jobs)
    res, crc = {}, {}
    for name, kbps, sha, size, csha, csize, crcs, differs in rows:
        res[f"{name}:{kbps}"] = dict(sha256=sha, size=size, committed_sha256=csha,
                                     committed_size=csize,
                                     blocks_differing_from_committed=differs)
        crc[f"{name}_{kbps}"] = crcs
        print(name, kbps, sha, size, "== committed" if sha == csha else f"differs in {differs}")
    json.dump(res, open(os.path.join(HERE, "vqfile.json"), "w"), indent=1,
              sort_keys=True)
    np.savez_compressed(os.path.join(HERE, "vqfile.npz"), **crc)


VQ_EXCERPT_HOPS = 24


def make_vq_excerpts(names):
    """excerpt_vq_<wav>.npz: the first 24 hops of each committed excerpt through
    the reference's shipped configuration at 128 and 96 kb/s."""
    sys.setrecursionlimit(12000)
    for name in names:
        ex = np.load(os.path.join(HERE, f"excerpt_{name}.npz"))
        pcm = ex["pcm"][:VQ_EXCERPT_HOPS * 1024]
        path = os.path.join(_work, f"{name}_vqex.wav")
        open(path, "wb").write(wav_bytes(int(ex["sr"]), pcm))
        res = {"hops": np.array(VQ_EXCERPT_HOPS)}
        for kbps in (128, 96):
            pac, flags = ref_encode_file(path, kbps, True,
                                         os.path.join(_work, f"{name}_vqex_{kbps}.pac"), vq=True)
            res[f"pac_vq{kbps}"] = np.frombuffer(pac, dtype=np.uint8)
            res[f"flags_vq{kbps}"] = np.array(flags, dtype=np.uint8)
            print(name, kbps, len(pac))
        np.savez_compressed(os.path.join(HERE, f"excerpt_vq_{name}.npz"), **res)


def make_full(names):
    from multiprocessing import Pool
    jobs = [(n, bs) for n in names for bs in (False, True)]
    with Pool(8) as pool:
        rows = pool.map(_full_one, jobs)
    res = {}
    for name, bs, sha, size, n_cur, n_blocks in rows:
        res[f"{name}:{'bs' if bs else 'long'}"] = dict(
            sha256=sha, size=size, cur_transient_hops=n_cur, hops_written=n_blocks)
        print(name, bs, sha, size)
    json.dump(res, open(os.path.join(HERE, "fullfile.json"), "w"), indent=1,
              sort_keys=True)


def make_kbd():
    """window.KBDWindow (coder/window.py:45-57) and its one use with the MDCT
    (coder/bitalloc.py:153-161: MDCT(KBDWindow(x), N//2, N//2) of the six-tone block)."""
    Fs = 48000
    freqs = np.array([440, 550, 660, 880, 4400, 8800])
    amps = np.array([0.43, 0.24, 0.15, 0.09, 0.05, 0.04])
    out = {}
    for N in (2048, 256, 1024):
        x = np.sum(np.array([amps[i] * np.cos(2 * np.pi * freqs[i] * np.arange(N) / Fs)
                             for i in range(6)]), axis=0)
        out[f"x_{N}"] = x
        out[f"kbd_{N}"] = window.KBDWindow(np.ones(N))
        out[f"kbd_x_{N}"] = window.KBDWindow(x)
        out[f"mdct_kbd_x_{N}"] = mdct.MDCT(window.KBDWindow(x), N // 2, N // 2)
    out["kbd_2048_alpha2p5"] = window.KBDWindow(np.ones(2048), alpha=2.5)
    out["kbd_512_alpha4"] = window.KBDWindow(np.ones(512))
    rng = np.random.default_rng(7)
    xs = rng.uniform(-1, 1, size=(3, 2048))
    out["rand_x"] = xs
    out["rand_kbd_x"] = np.stack([window.KBDWindow(r) for r in xs])
    np.savez_compressed(os.path.join(HERE, "kbd.npz"), **out)
    print("kbd.npz written")


def make_sbr_scalar():
    """Scalar mantissas + SBR (useVQ off, useSBR on) through the reference's own file loop.
    Records what the reference does with each case in sbr_scalar.json: the branch at
    coder/codec.py:541-548 hands a NumPy SCALAR to vMantissa, whose vQuantizeUniform then
    assigns into it (coder/quantize.py:73-74) -- a TypeError in every NumPy version as soon
    as an omitted band receives bits."""
    import traceback
    report = []
    for name, kbps, bs in (("castanet", 96, True), ("harpsichord", 96, False), ("quar48_1", 64, True),
                           ("harpsichord", 128, True), ("quar48_1", 32, False), ("castanet", 24, False)):
        ex = np.load(os.path.join(HERE, f"excerpt_{name}.npz"))
        pcm, sr = ex["pcm"][:24 * 1024], int(ex["sr"])
        wav = os.path.join(_work, f"sbrs_{name}.wav")
        open(wav, "wb").write(wav_bytes(sr, pcm))
        pac_path = os.path.join(_work, f"sbrs_{name}_{kbps}.pac")
        entry = {"excerpt": name, "hops": 24, "kbps_per_channel": kbps, "block_switching": bs,
                 "useVQ": False, "useSBR": True}
        try:
            pac, flags = ref_encode_file(wav, kbps, bs, pac_path, vq=False, sbr=True)
            entry["outcome"] = "encoded"
            entry["bytes"] = len(pac)
            entry["sha256"] = hashlib.sha256(pac).hexdigest()
            dec = ref_decode_file(pac_path, os.path.join(_work, f"sbrs_{name}_{kbps}_dec.wav"))
            entry["decoded_shape"] = list(dec.shape)
            entry["decoded_sha256"] = hashlib.sha256(np.ascontiguousarray(dec).astype("<i2").tobytes()).hexdigest()
        except Exception as e:                                   # noqa: BLE001
            tb = traceback.extract_tb(e.__traceback__)
            entry["outcome"] = "raised"
            entry["error"] = f"{type(e).__name__}: {e}"
            entry["where"] = [f"{os.path.relpath(f.filename, REF)}:{f.lineno} {f.name}" for f in tb
                              if f.filename.startswith(REF)]
        report.append(entry)
        print(entry)
    json.dump(report, open(os.path.join(HERE, "sbr_scalar.json"), "w"), indent=1)


def make_sbr_scalar_decode():
    """Decode_SBR's scalar branch (coder/codec.py:117-134 with useVQ off) and the reader rule that feeds
    it (coder/pacfile.py:203-205, 659-663).  No reference ENCODER writes a scalar-mantissa block with a coded
    omitted band (make_sbr_scalar above), so the inputs are made here: plain scalar streams of the excerpts
    (the oracle's encoder, bit-identical to the reference's) rewritten with one mantissa per omitted band
    (oracle.pac_oracle.recode_scalar_sbr_stream), and random code sets.  The OUTPUTS are the reference's:
    its PACFile reader + PCMFile writer on the streams, its codec.Decode_SBR on the code sets."""
    from oracle import pac_oracle as po
    out = {}
    cases = []
    for name, kbps, bs, drop, h0 in (("harpsichord", 96, False, False, 0), ("castanet", 192, True, True, 30),
                                     ("quar48_1", 256, True, False, 40), ("spmg", 192, False, True, 0)):
        ex = np.load(os.path.join(HERE, f"excerpt_{name}.npz"))
        pcm, sr = ex["pcm"][h0 * 1024:(h0 + 16) * 1024], int(ex["sr"])
        plain = po.encode_stream(pcm, sr, kbps, bs)
        keep = (lambda hop, ch, band: (hop + ch + band) % 3 != 0) if drop else (lambda hop, ch, band: True)
        pac = po.recode_scalar_sbr_stream(plain, keep)
        tag = f"{name}_{kbps}_{'bs' if bs else 'long'}"
        path = os.path.join(_work, f"sbrsd_{tag}.pac")
        open(path, "wb").write(pac)
        dec = ref_decode_file(path, os.path.join(_work, f"sbrsd_{tag}.wav"))
        out[f"pac_{tag}"] = np.frombuffer(pac, dtype=np.uint8)
        out[f"pcm_{tag}"] = dec.astype(np.int16)
        cases.append(tag)
        print(tag, len(pac), dec.shape)
    rng = np.random.default_rng(20261004)
    for sr in (48000, 44100, 32000):
        cp = ref_params(sr, 1, 96)
        cp.useSBR = True
        cp.omittedBands = sbr.omitted_bands(cp.sfBands)
        nb = cp.sfBands.nBands
        K = 12
        sfs = rng.integers(0, 16, (K, nb)).astype(np.int32)
        bas = rng.integers(0, 13, (K, nb)).astype(np.int32)
        bas[bas == 1] = 0
        bas[0, cp.omittedBands] = 0                      # (the caller sends this one to Decode; Decode_SBR takes it all the same)
        bas[1, :] = 0
        bas[1, cp.omittedBands[-1]] = 5
        mant = np.zeros((K, 1024), np.int32)
        for k in range(K):
            for b in range(nb):
                a = int(bas[k, b])
                if not a:
                    continue
                lo, hi = cp.sfBands.lowerLine[b], cp.sfBands.upperLine[b] + 1
                if b in cp.omittedBands:
                    mant[k, lo:hi] = rng.integers(0, 1 << a)
                else:
                    mant[k, lo:hi] = rng.integers(0, 1 << a, hi - lo)
        ov = rng.integers(0, 16, K).astype(np.int32)
        fl = rng.integers(0, 2, (K, 2)).astype(np.int32)     # lastTrans, nextTrans
        blocks = np.stack([codec.Decode_SBR(sfs[k], bas[k], mant[k], int(ov[k]), None, cp,
                                            bool(fl[k, 0]), False, bool(fl[k, 1])) for k in range(K)])
        out[f"fn_{sr}_sf"], out[f"fn_{sr}_ba"], out[f"fn_{sr}_mant"] = sfs, bas, mant
        out[f"fn_{sr}_overall"], out[f"fn_{sr}_flags"], out[f"fn_{sr}_block"] = ov, fl, blocks
        print(sr, "Decode_SBR scalar:", blocks.shape, float(np.abs(blocks).max()))
    out["stream_cases"] = np.array(cases)
    np.savez_compressed(os.path.join(HERE, "sbr_scalar_decode.npz"), **out)


def make_lines512():
    """nMDCTLines = 512: the reference's own file loop with 512-line long blocks (1024-sample windows;
    a short-coded hop is then FOUR 128-line sub-blocks, coder/pacfile.py:489-547) on 12 288-sample excerpts, long-only
    and block-switched, and its decoder on the result."""
    out = {}
    cases = []
    for name, kbps, bs, h0 in (("harpsichord", 128, False, 0), ("castanet", 128, True, 30), ("spmg", 96, True, 8)):
        ex = np.load(os.path.join(HERE, f"excerpt_{name}.npz"))
        pcm, sr = ex["pcm"][h0 * 1024:(h0 + 12) * 1024], int(ex["sr"])
        tag = f"{name}_{kbps}_{'bs' if bs else 'long'}"
        wav = os.path.join(_work, f"l512_{tag}.wav")
        open(wav, "wb").write(wav_bytes(sr, pcm))
        pac_path = os.path.join(_work, f"l512_{tag}.pac")
        pac, flags = ref_encode_file(wav, kbps, bs, pac_path, n_lines=512)
        dec = ref_decode_file(pac_path, os.path.join(_work, f"l512_{tag}_dec.wav"))
        out[f"pcm_{tag}"], out[f"sr_{tag}"] = pcm, sr
        out[f"pac_{tag}"] = np.frombuffer(pac, dtype=np.uint8)
        out[f"dec_{tag}"] = dec.astype(np.int16)
        out[f"flags_{tag}"] = np.array(flags, dtype=np.uint8)
        cases.append(tag)
        print(tag, len(pac), dec.shape, int(np.array(flags)[:, 1].sum()), "short-coded hops")
    out["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(HERE, "lines512.npz"), **out)


if __name__ == "__main__":
    if "--lines512" in sys.argv:
        make_lines512()
        sys.exit(0)
    if "--sbr-scalar-decode" in sys.argv:
        make_sbr_scalar_decode()
        sys.exit(0)
    if "--sbr-scalar" in sys.argv:
        make_sbr_scalar()
        sys.exit(0)
    if "--kbd" in sys.argv:
        make_kbd()
        sys.exit(0)
    names = ["castanet", "harpsichord", "quar48_1", "spmg"]
    wavs = {n: read_wav(os.path.join(REF, "test_signals", n + ".wav"))
            for n in names}
    if "--decoded" in sys.argv:
        make_decoded()
    elif "--vq-decoded" in sys.argv:
        make_vq_decoded()
    elif "--vq" in sys.argv:
        make_vq_excerpts(names)
        make_vq(names)
    elif "--full" in sys.argv:
        make_full(names)
        make_full_inputs(names)
    else:
        make_tables()
        make_stages(wavs)
        make_excerpts(wavs)
