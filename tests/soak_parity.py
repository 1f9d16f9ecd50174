"""Parity soak (test infrastructure, run by hand on the GPU box -- not collected by pytest):

    python3 tests/soak_parity.py [--minutes 8] [--workers 14] [--seed0 1000] [--coders vq] [--seeds a,b,c] [--out gpurun_out/soak.txt]

Random programmes (chords, noise beds, sweeps, level steps over 70 dB, clicks, gaps of digital silence,
clipping), random sample rate / bit rate / channel count, through the three stream coders of the product
(scalar, scalar + block switching, gain-shape + block switching (+ SBR below 128 kb/s)) and, for each, the
.pac bytes AND the decoded PCM against the oracle (oracle/pac_oracle.py, oracle/pac_oracle_vq.py), which runs
in worker processes on the host cores while the GPU encodes.  The suite's parity tests cover a few thousand
channel-frames of fixed material; this covers as many different ones as the time given allows and writes what
it saw.  A stream that differs is taken apart block by block (classify_scalar_mismatch, classify_vq_mismatch):
differences where the reference's own rounding noise decides -- the product's PACX_ST_GUARD flag is up, an exactly
zero line's sign bit, an input with an exactly sparse or flat spectrum -- are counted by class, anything else is a
MISMATCH, reported with the case's parameters (every case is reproducible from its seed on the same machine;
NumPy's SIMD sin/exp differ in the last place between CPU generations, so another machine draws other programmes)."""
import argparse
import hashlib
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def programme(seed, n_hops, n_ch, sr):
    rng = np.random.default_rng(seed)
    n = n_hops * 1024
    t = np.arange(n) / sr
    out = np.zeros((n, n_ch))
    common = rng.standard_normal(n) if rng.random() < 0.5 else None       # correlated channels, sometimes
    for ch in range(n_ch):
        x = np.zeros(n)
        for _ in range(int(rng.integers(0, 9))):                          # partials with slow envelopes
            f = rng.uniform(30, 0.45 * sr)
            env = np.clip(np.sin(2 * np.pi * rng.uniform(0.3, 6.0) * t + rng.uniform(0, 6.28)), 0, 1) ** 2
            x += rng.uniform(0.005, 0.3) * env * np.sin(2 * np.pi * f * t + rng.uniform(0, 6.28))
        if rng.random() < 0.4:                                            # a sweep
            f0, f1 = rng.uniform(50, 2000), rng.uniform(2000, 0.4 * sr)
            x += rng.uniform(0.02, 0.3) * np.sin(2 * np.pi * (f0 * t + (f1 - f0) * t * t / (2 * t[-1] + 1e-9)))
        kind = rng.integers(0, 4)
        noise = rng.standard_normal(n) if common is None or rng.random() < 0.5 else common
        if kind == 1:
            noise = np.convolve(noise, np.ones(8) / 8, mode="same")
        elif kind == 2:
            noise = np.diff(noise, prepend=0.0)
        elif kind == 3:
            noise = noise * (np.sin(2 * np.pi * rng.uniform(0.5, 4.0) * t + ch) > 0)
        x += noise * 10.0 ** rng.uniform(-4.5, -0.7)
        step = int(rng.integers(2, 12)) * 1024                            # level steps
        x *= 10.0 ** (-rng.uniform(0, 3.5) * ((np.arange(n) // step) % 3) / 2.0)
        for c in rng.integers(1100, n - 1100, int(rng.integers(0, 6))):   # clicks and bursts
            w = int(rng.integers(4, 200))
            x[c:c + w] += rng.choice([-1.0, 1.0]) * rng.uniform(0.2, 0.95) * (rng.standard_normal(w) if rng.random() < 0.5 else 1.0)
        if rng.random() < 0.3:                                            # digital silence
            g = int(rng.integers(0, n - 4000))
            x[g:g + int(rng.integers(1500, 4000))] = 0.0
        if rng.random() < 0.15:                                           # hot master: clipping
            x *= rng.uniform(2.0, 6.0)
        out[:, ch] = x
    return np.clip(np.rint(out * 32767), -32768, 32767).astype(np.int16)


CODERS = ["scalar", "scalar_bs", "vq"]
EXTRA_CODERS = ["scalar_sbr"]        # only with --coders: scalar mantissas in an SBR file (the reference's driver never selects it;
                                     # defined while no omitted band gets bits, TypeError otherwise: DESIGN.md section 7)


def draw_case(seed, coders=None, rates=None, channels=None):
    rng = np.random.default_rng(seed ^ 0x5EED)
    coder = CODERS[int(rng.integers(0, 3))]
    if coders and coder not in coders:                     # a restricted run keeps the other draws of the seed
        coder = coders[seed % len(coders)]
    sr = [48000, 44100, 32000, 96000][int(rng.choice(4, p=[0.55, 0.25, 0.1, 0.1]))]
    if rates:
        sr = rates[seed % len(rates)]
    n_ch = int(rng.choice([1, 2, 3], p=[0.15, 0.75, 0.1]))
    if channels:
        n_ch = channels[seed % len(channels)]
    if coder == "vq":
        kbps = int(rng.choice([48, 64, 96, 128, 192, 256]))
        n_hops = int(rng.integers(8, 25))
    elif coder == "scalar_sbr":
        kbps = int(rng.choice([24, 32, 48, 64, 96]))
        n_hops = int(rng.integers(8, 49))
    else:
        kbps = int(rng.choice([32, 64, 96, 128, 192, 320]))
        n_hops = int(rng.integers(16, 97))
    return dict(seed=seed, coder=coder, sr=sr, n_ch=n_ch, kbps=kbps, n_hops=n_hops)


def oracle_side(case):
    """worker process: NumPy only, never touches the GPU"""
    from oracle import pac_oracle as po, pac_oracle_vq as pv
    pcm = programme(case["seed"], case["n_hops"], case["n_ch"], case["sr"])
    t0 = time.time()
    vq = case["coder"] == "vq"
    try:
        pac = pv.encode_stream_vq(pcm, case["sr"], case["kbps"]) if vq else \
            po.encode_stream(pcm, case["sr"], case["kbps"], case["coder"] != "scalar", use_sbr=case["coder"] == "scalar_sbr")
    except Exception as e:                                                # the oracle follows the reference's raises
        return case, None, "raised " + type(e).__name__, time.time() - t0
    try:
        dec = pv.decode_stream_vq(pac) if vq else po.decode_stream(pac)
    except Exception as e:                                                # ... of its decoder too (SBR above 48 kHz)
        return case, pac, "raised " + type(e).__name__, time.time() - t0
    return case, pac, hashlib.sha256(np.ascontiguousarray(dec).tobytes()).hexdigest(), time.time() - t0


_GUARD_ENCODERS = {}


def _blocks(b, hdr):
    out, pos = [], hdr
    while pos < len(b):
        n = int.from_bytes(b[pos:pos + 4], "little")
        out.append(b[pos + 4:pos + 4 + n])
        pos += 4 + n
    return out


def _parse_scalar_block(po, p, b):
    br = po.BitReader(b)
    fl = [br.get(1) for _ in range(3)]
    units = []
    for _ in range(8 if fl[1] else 1):
        bands = p.sfBandsShort if fl[1] else p.sfBands
        ov = br.get(4)
        ba, sf, mant = [], [], []
        for k in range(bands.nBands):
            a = br.get(12)
            a = a + 1 if a else 0
            ba.append(a)
            sf.append(br.get(4))
            mant.append([br.get(a) for _ in range(bands.nLines[k])] if a else [])
        units.append((ov, ba, sf, mant))
    return fl, units


def degenerate(po, samples, windowed_lines=None):
    """a (sub-)block whose coding the reference's FFT rounding noise decides: at most 8 non-zero samples (impulses:
    an exactly flat spectrum, every local-maximum test a coin toss) or a quarter of its MDCT lines below 1e-12 of
    the largest (constant, period-2 / period-4 inputs: one or two bins, the rest noise)"""
    if np.count_nonzero(samples) <= 8:
        return True
    X = windowed_lines
    if X is None:
        n = len(samples) // 2
        X = po.mdct_forward(po.sine_window(2 * n) * po.pcm16_to_fraction(samples), n, n)[:n]
    top = float(np.max(np.abs(X)))
    return top == 0.0 or np.count_nonzero(np.abs(X) < 1e-12 * top) >= len(X) // 4


def classify_vq_mismatch(A, case, pcm, got, want):
    """gain-shape streams: every differing block must carry the product's PACX_ST_GUARD flag (round 3: the gain-shape
    coder raises it for split angles, band gains, pulse-search floors and ties, and lines at rounding-noise level) or
    hold a degenerate (sub-)block (see degenerate()) / an exactly-zero line, else 'REAL'"""
    from oracle import pac_oracle as po
    p = po.make_params(case["sr"], case["n_ch"], case["kbps"])
    hdr = len(po.pac_header(p, len(pcm)))
    bg, bw = _blocks(got, hdr), _blocks(want, hdr)
    if got[:hdr] != want[:hdr] or len(bg) != len(bw):
        return ["REAL: header or block count"]
    key = (case["sr"], case["kbps"], "vq")
    if key not in _GUARD_ENCODERS:                   # a handle that raises PACX_ST_GUARD (same bytes, a few per cent slower)
        _GUARD_ENCODERS[key] = A.engine.Encoder(case["sr"], case["kbps"] / (case["sr"] / 1000), use_vq=True,
                                                use_sbr=case["kbps"] < 128, guard=True)
    enc = _GUARD_ENCODERS[key]
    planar = A.pacfile.device_stream(enc, pcm)
    flags = enc.transient_flags(planar, len(pcm) // 1024, 1024)[1]
    status = enc.encode_vq(A.engine.PcmView.stream(planar, 1024), flags)["status"].cpu().numpy()
    fl = flags.cpu().numpy()
    host = planar.cpu().numpy()
    n_ch = case["n_ch"]
    n_frames = len(status) // n_ch
    kept = [f for f in range(n_frames) if not any(int(status[f * n_ch + c]) & A._lib.ST_ZERO_SUBBLOCK for c in range(n_ch))]
    if len(kept) * n_ch != len(bw):
        return [f"REAL: the oracle wrote {len(bw)} blocks, the product's status words keep {len(kept) * n_ch}"]
    classes = []
    for i, (x, y) in enumerate(zip(bg, bw)):
        if x == y:
            continue
        f, ch = kept[i // n_ch], i % n_ch
        blk = host[ch, f * 1024:(f + 2) * 1024]
        if (x[0] >> 5) != (y[0] >> 5):
            classes.append(f"REAL: block {i} flags")
        elif int(status[f * n_ch + ch]) & A._lib.ST_GUARD:
            # the product's own flag: a BitAlloc value, a split angle, a band gain, a pulse-search floor or tie, or a
            # line at rounding-noise level sits within rounding distance of its decision boundary in this channel-frame
            classes.append(f"guard: block {i} (frame {f} of {n_frames}, channel {ch})")
        elif (int(fl[f]) >> 1) & 1:
            deg = any(degenerate(po, blk[448 + 128 * s_:448 + 128 * s_ + 256]) for s_ in range(8))
            classes.append(("degenerate" if deg else "REAL") + f": block {i} (frame {f} of {n_frames}, channel {ch}, short)")
        else:
            kind = "degenerate" if degenerate(po, blk) else "REAL"
            if kind == "REAL":
                # a long block with lines that are zero in exact arithmetic: the reference's SPL(0) = 1e-8 rule makes its
                # SMRs hang on the FFT's rounding (see classify_scalar_mismatch, 'zero-line')
                X = po.mdct_forward(po.apply_window(po.pcm16_to_fraction(blk), bool(fl[f] & 1), False, bool(fl[f] & 4)), 1024, 1024)[:1024]
                if np.count_nonzero(np.abs(X) < 1e-12 * np.max(np.abs(X))):
                    kind = "zero-line"
            classes.append(kind + f": block {i} (frame {f} of {n_frames}, channel {ch})")
    return classes


def classify_decode_mismatch(case, stream, decoded):
    """The streams are equal, the decoded PCM is not: 'decode-tie' if every differing sample differs by one code and
    the oracle's own float sample sits within 1e-9 of the 16-bit quantiser's rounding boundary ((2^16 - 1)|x| + 1 an
    even integer, coder/quantize.py:73 through coder/pcmfile.py:127-134) -- the IMDCTs agree to ~1e-16 relative,
    which side of the boundary such a sample falls on is rounding noise; anything else 'REAL'."""
    from oracle import pac_oracle as po, pac_oracle_vq as pv
    floats, orig = [], po.fraction_to_pcm16

    def spy(x):
        floats.append(np.array(x, dtype=np.float64))
        return orig(x)
    po.fraction_to_pcm16 = spy
    try:
        d_o = pv.decode_stream_vq(stream) if case["coder"] == "vq" else po.decode_stream(stream)
    finally:
        po.fraction_to_pcm16 = orig
    if d_o.shape != decoded.shape:
        return ["REAL: decoded lengths differ"]
    n_ch = case["n_ch"]
    fl = np.concatenate([np.stack(floats[i:i + n_ch], axis=1) for i in range(0, len(floats), n_ch)])
    if fl.shape != d_o.shape:
        return ["REAL: could not follow the oracle's decoder"]
    diff = decoded.astype(np.int64) - d_o.astype(np.int64)
    idx = np.argwhere(diff != 0)
    t = 65535.0 * np.abs(fl[idx[:, 0], idx[:, 1]]) + 1.0
    near = np.abs(t - 2.0 * np.round(t / 2.0))
    if np.abs(diff).max() == 1 and np.all(near < 1e-9):
        return [f"decode-tie: {len(idx)} sample(s), the closest {near.min():.1e} and the farthest {near.max():.1e} from the boundary"]
    return [f"REAL: {len(idx)} samples differ by up to {int(np.abs(diff).max())}, up to {near.max():.1e} from a rounding boundary"]


def classify_scalar_mismatch(A, case, pcm, got, want):
    """Why a scalar stream differs from the oracle's, block by block.  Two classes are the reference's own
    rounding noise and not a disagreement about the algorithm:
      guard      the product's PACX_ST_GUARD flag is up for the channel-frame: a BitAlloc value or a quantiser input
                 sits within an ulp-scale margin of a rounding boundary (coder/bitalloc.py:103, coder/quantize.py:73),
                 so the last bits of the SMRs (compared to 1e-9 dB, not bit for bit) decide;
      alloc-follows-smr  the allocations differ, the flag is down, but the product's allocation IS the oracle's
                 BitAlloc of the product's own SMRs, and those lie within 1e-9 dB of the oracle's: an earlier pass of
                 the water-filling loop (not the final one, which the flag watches) sat on a rounding boundary;
      zero-line  a line whose exact value is zero (the oracle's own line is below 1e-12 of the block maximum).  Either
                 only sign bits of zero-magnitude mantissas differ there (the sign of NumPy's FFT rounding noise), or
                 the allocation differs and the oracle's SMRs, redone with those lines at exactly 0.0, are the
                 product's: the reference's SPL() gives an exactly zero intensity 1e-8 (+16 dB) and a 1e-40 one the
                 -30 dB floor, so the line's level hangs on whether the FFT's rounding left 0.0 or 1e-21;
      degenerate a quarter or more of the (sub-)block's lines are such rounding noise in the oracle's own MDCT (constant,
                 period-2 / period-4 and similar inputs: clipped stretches, +-2 LSB tones) or it holds a handful of
                 impulses (a flat spectrum): which noise bins are "peaks", and everything downstream, is the
                 reference's FFT rounding ("known limit");
    anything else is returned as 'REAL'."""
    from oracle import pac_oracle as po
    bs = case["coder"] != "scalar"
    sbr = case["coder"] == "scalar_sbr"            # long blocks: another budget and overall scale -- only the guard flag,
    p = po.make_params(case["sr"], case["n_ch"], case["kbps"])        # the zero-line and the degenerate tests apply
    hdr = len(po.pac_header(p, len(pcm)))
    if got[:hdr] != want[:hdr]:
        return ["REAL: header"]
    bg, bw = _blocks(got, hdr), _blocks(want, hdr)
    if len(bg) != len(bw):
        return [f"REAL: {len(bg)} blocks against {len(bw)}"]
    key = (case["sr"], case["kbps"], sbr)
    if key not in _GUARD_ENCODERS:
        _GUARD_ENCODERS[key] = A.engine.Encoder(case["sr"], case["kbps"] / (case["sr"] / 1000), guard=True, use_sbr=sbr)
    enc = _GUARD_ENCODERS[key]
    planar = A.pacfile.device_stream(enc, pcm)
    flags = enc.transient_flags(planar, len(pcm) // 1024, 1024)[1] if bs else None
    status = enc.encode_pack(A.engine.PcmView.stream(planar, 1024), flags)["status"].cpu().numpy()
    host = planar.cpu().numpy()
    classes = []
    n_ch = case["n_ch"]
    n_frames = len(status) // n_ch
    # the writer leaves out a short-coded hop with an all-zero sub-block in any channel (coder/pacfile.py:530-533)
    kept = [f for f in range(n_frames) if not any(int(status[f * n_ch + c]) & A._lib.ST_ZERO_SUBBLOCK for c in range(n_ch))]
    if len(kept) * n_ch != len(bw):
        return [f"REAL: the oracle wrote {len(bw)} blocks, the product's status words keep {len(kept) * n_ch}"]
    import torch
    max_mant = min(1 << p.nMantSizeBits, 16)

    def oracle_unit(blk, fx, s_):
        """the oracle's stages of one (sub-)block"""
        short = bool(fx[1])
        sub = blk[448 + 128 * s_:448 + 128 * s_ + 256] if short else blk
        st = {}
        if short:
            p.nMDCTLines = p.nSamplesPerBlock = 128
        try:
            po.encode_channel(po.pcm16_to_fraction(sub), p, bool(fx[0]), short, bool(fx[2]), stages=st)
        finally:
            p.nMDCTLines = p.nSamplesPerBlock = 1024
        return st

    for i, (x, y) in enumerate(zip(bg, bw)):
        if x == y:
            continue
        f, ch = kept[i // n_ch], i % n_ch
        where = f"block {i} (frame {f} of {n_frames}, channel {ch})"
        if int(status[f * n_ch + ch]) & A._lib.ST_GUARD:
            classes.append(f"guard: {where}")
            continue
        fx, ux = _parse_scalar_block(po, p, x)
        fy, uy = _parse_scalar_block(po, p, y)
        if fx != fy:
            classes.append(f"REAL: {where} flags {fx}/{fy}")
            continue
        short = bool(fx[1])
        bands = p.sfBandsShort if short else p.sfBands
        nb = bands.nBands
        blk = host[ch, f * 1024:(f + 2) * 1024]
        smr = None
        verdicts = set()
        for s_, (a, b) in enumerate(zip(ux, uy)):
            if a == b:
                continue
            st = oracle_unit(blk, fx, s_)
            X = st["mdct"]
            top = float(np.max(np.abs(X)))
            if degenerate(po, blk[448 + 128 * s_:448 + 128 * s_ + 256] if short else blk, X):
                verdicts.add("degenerate")                 # the reference's FFT rounding noise decides this (sub-)block
                continue
            if a[1] != b[1] and sbr and not short:
                verdicts.add("REAL (allocation of a long block of an SBR file: not taken apart)")
            elif a[1] != b[1]:
                if smr is None:
                    view = A.engine.PcmView.frames(torch.as_tensor(np.ascontiguousarray(blk), device=enc.device).view(1, 1, 2048))
                    smr = enc.smr(view, enc.mdct(view, [tuple(fx)], short=short), short=short).cpu().numpy()[0]
                    if short:
                        p.nMDCTLines = p.nSamplesPerBlock = 128
                    try:
                        budget = po.bit_budget(p, bool(fx[0]), short, bool(fx[2]))
                    finally:
                        p.nMDCTLines = p.nSamplesPerBlock = 1024
                mine = smr[s_ * nb:(s_ + 1) * nb]
                again = po.bit_alloc(budget, max_mant, nb, bands.nLines, mine)
                worst = float(np.max(np.abs(st["smr"][:nb] - mine)))
                if again.tolist() == a[1] and worst < 1e-9:
                    verdicts.add("alloc-follows-smr")
                    continue
                # SPL() takes 1e-8 (+16 dB) for an intensity that is EXACTLY zero and the -30 dB floor for one of
                # 1e-40 (coder/psychoac.py:13-24): a line that is zero in exact arithmetic gets one or the other
                # depending on whether the FFT's rounding left 0.0 or 1e-21 -- 46 dB apart in that line's SMR.
                # The oracle's SMRs redone with its rounding-noise lines set to exact zeros must then be the product's
                sub = blk[448 + 128 * s_:448 + 128 * s_ + 256] if short else blk
                X0 = np.where(np.abs(X) < 1e-12 * top, 0.0, X)
                ov = po.scale_factor(top, p.nScaleBits)
                smr0 = po.calc_smrs(po.pcm16_to_fraction(sub), X0 * (1 << ov), ov, case["sr"], bands)
                if again.tolist() == a[1] and np.any(X0 != X) and float(np.max(np.abs(smr0[:nb] - mine))) < 1e-9:
                    verdicts.add("zero-line")
                else:
                    verdicts.add(f"REAL (allocation; SMRs {worst:.1e} dB apart)")
            elif a[0] != b[0] or a[2] != b[2]:
                verdicts.add("REAL (overall scale or scale factors)")
            else:
                ok = True
                for k in range(nb):
                    half = 1 << (a[1][k] - 1) if a[1][k] else 0
                    for j, (u, v) in enumerate(zip(a[3][k], b[3][k])):
                        if u != v and ((u & (half - 1)) or (v & (half - 1)) or abs(X[bands.lowerLine[k] + j]) > 1e-12 * top):
                            ok = False
                verdicts.add("zero-line" if ok else "REAL (mantissas)")
        for v in sorted(verdicts):
            classes.append(f"{v}: {where}")
    return classes


def first_difference(a, b):
    n = min(len(a), len(b))
    d = np.nonzero(np.frombuffer(a[:n], np.uint8) != np.frombuffer(b[:n], np.uint8))[0]
    return int(d[0]) if len(d) else n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=8.0)
    ap.add_argument("--workers", type=int, default=14)
    ap.add_argument("--seed0", type=int, default=1000)
    ap.add_argument("--seeds", default="", help="comma-separated seeds to replay instead of a timed run")
    ap.add_argument("--rates", default="", help="comma-separated sample rates instead of the 32 / 44.1 / 48 / 96 kHz mix")
    ap.add_argument("--channels", default="", help="comma-separated channel counts instead of the 1 / 2 / 3 mix")
    ap.add_argument("--coders", default="", help="comma-separated subset of scalar,scalar_bs,vq (default: all three)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "soak.txt"))
    a = ap.parse_args()
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    ctx = mp.get_context("spawn")
    pool = ctx.Pool(a.workers)
    import importlib
    A = importlib.import_module("audio_codec_amd")
    deadline = time.time() + 60.0 * a.minutes
    coders = [c for c in a.coders.split(",") if c] or None
    assert not coders or all(c in CODERS + EXTRA_CODERS for c in coders)
    rates = [int(r) for r in a.rates.split(",") if r] or None
    channels = [int(r) for r in a.channels.split(",") if r] or None
    if a.seeds:
        replay = [int(x) for x in a.seeds.split(",")]
        seeds = iter(())
        deadline = 0.0
        pending = [pool.apply_async(oracle_side, (draw_case(x, coders, rates, channels),)) for x in replay]
    else:
        seeds = iter(range(a.seed0, a.seed0 + 10 ** 6))
        pending = [pool.apply_async(oracle_side, (draw_case(next(seeds), coders, rates, channels),)) for _ in range(2 * a.workers)]
    tally, bad, raised, dec_raised, last_print, noise = {}, [], 0, 0, time.time(), {}
    log = open(a.out, "w")
    def say(s):
        print(s, flush=True)
        log.write(s + "\n")
        log.flush()
    say(f"parity soak: {a.minutes} min, {a.workers} oracle workers, seeds from {a.seed0}" + (f", coders {coders}" if coders else "") + (f", rates {rates}" if rates else "") + (f", channels {channels}" if channels else ""))
    while pending:
        res = pending.pop(0)
        case, want, dec_want, secs = res.get()
        if time.time() < deadline:
            pending.append(pool.apply_async(oracle_side, (draw_case(next(seeds), coders, rates, channels),)))
        pcm = programme(case["seed"], case["n_hops"], case["n_ch"], case["sr"])
        vq = case["coder"] == "vq"
        try:
            got = A.pacfile.encode_stream(pcm, case["sr"], case["kbps"], block_switching=case["coder"] != "scalar",
                                          use_vq=vq, use_sbr=(vq and case["kbps"] < 128) or case["coder"] == "scalar_sbr")
        except Exception as e:
            got = None
            err = repr(e)
        key = (case["coder"], case["sr"], case["kbps"])
        n_cf = case["n_hops"] * case["n_ch"]
        t = tally.setdefault(key, [0, 0, 0, 0])
        t[0] += 1
        t[1] += n_cf
        if want is None or got is None:
            raised += 1
            if (want is None) != (got is None):
                bad.append((case, f"encoder, one side raised: oracle {dec_want if want is None else 'ok'}, product {err if got is None else 'ok'}"))
                say(f"MISMATCH {case}: {bad[-1][1]}")
            continue
        if got != want and not vq:
            cls = classify_scalar_mismatch(A, case, pcm, got, want)
            kinds = sorted({c.split(":")[0] for c in cls})
            for k in kinds:
                noise[k] = noise.get(k, 0) + 1
            if not any(k.startswith("REAL") for k in kinds):
                t[3] += 1
                say(f"  noise-decided {case}: " + "; ".join(cls[:4]) + (" ..." if len(cls) > 4 else ""))
                if hashlib.sha256(np.ascontiguousarray(A.pacfile.decode_stream(want)).tobytes()).hexdigest() != dec_want:
                    bad.append((case, "the product decodes the oracle's stream differently"))
                    say(f"MISMATCH {case}: {bad[-1][1]}")
                continue
            t[2] += 1
            bad.append((case, f"bytes differ at {first_difference(got, want)} of {len(want)} (product {len(got)}): " + "; ".join(cls[:4])))
            say(f"MISMATCH {case}: {bad[-1][1]}")
            continue
        if got != want:
            cls = classify_vq_mismatch(A, case, pcm, got, want)
            kinds = sorted({c.split(":")[0] for c in cls})
            for k in kinds:
                noise[k] = noise.get(k, 0) + 1
            if not any(k.startswith("REAL") for k in kinds):
                t[3] += 1
                say(f"  noise-decided {case}: " + "; ".join(cls[:4]) + (" ..." if len(cls) > 4 else ""))
                continue
            t[2] += 1
            bad.append((case, f"bytes differ at {first_difference(got, want)} of {len(want)} (product {len(got)}): " + "; ".join(cls[:4])))
            say(f"MISMATCH {case}: {bad[-1][1]}")
            continue
        try:
            dec = A.pacfile.decode_stream(got)
            dec_got = hashlib.sha256(np.ascontiguousarray(dec).tobytes()).hexdigest()
        except Exception as e:
            dec_got = "raised " + type(e).__name__
        if dec_got.startswith("raised") and dec_want.startswith("raised"):
            dec_raised += 1                                               # both decoders refuse the stream
        elif dec_got != dec_want:
            cls = ["REAL: one decoder raised"] if "raised" in dec_got + dec_want else classify_decode_mismatch(case, want, dec)
            if cls[0].startswith("decode-tie"):
                t[3] += 1
                noise["decode-tie"] = noise.get("decode-tie", 0) + 1
                say(f"  noise-decided {case}: {cls[0]}")
            else:
                t[2] += 1
                bad.append((case, f"decoder: product {dec_got[:24]}, oracle {dec_want[:24]}: {cls[0]}"))
                say(f"MISMATCH {case}: {bad[-1][1]}")
        if time.time() - last_print > 45:
            last_print = time.time()
            say(f"  ... {sum(v[0] for v in tally.values())} streams, {sum(v[1] for v in tally.values())} channel-frames, {len(bad)} mismatches")
    pool.close()
    pool.join()
    say("coder      rate  kb/s  streams  channel-frames  mismatching streams  streams differing only where the reference's rounding noise decides")
    for key in sorted(tally):
        v = tally[key]
        say(f"{key[0]:<10} {key[1]:>5} {key[2]:>4}  {v[0]:>7}  {v[1]:>14}  {v[2]:>8}  {v[3]:>8}")
    say(f"noise-decided streams by class (a stream can be in both): {noise}")
    n_diff = sum(v[3] for v in tally.values())
    say(f"total: {sum(v[0] for v in tally.values())} streams, {sum(v[1] for v in tally.values())} channel-frames, "
        f"{raised} streams where both encoders raised, {dec_raised} where both decoders raised; "
        f"{n_diff + len(bad)} streams differ from the oracle in bytes or decoded PCM: {n_diff} classified (above), {len(bad)} NOT classified")
    for case, why in bad:
        say(f"  {case}: {why}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
