"""Machine-independent test programmes for the parity soak slice (test infrastructure).

tests/soak_parity.py draws its programmes with NumPy's float sin / standard_normal, whose last place differs
between CPU generations (SIMD dispatch): its cases reproduce from their seed on the same machine only.  The
programmes here are made with INTEGER arithmetic alone -- PCG64 integers (bit-exact on every platform), a committed
4096-entry int16 sine table (tests/golden/soak_sine.npy; its sha256 is checked), 32-bit phase accumulators,
piecewise-linear integer envelopes -- so a seed names the same int16 PCM everywhere, and the digests committed in
tests/golden/soak_slice.json are ASSERTED on the GPU box, not skipped.

Same ingredients as the soak: partials with slow envelopes, a sweep, four kinds of noise bed, level steps of up
to ~70 dB, clicks and bursts, gaps of digital silence, clipping.
"""
import hashlib
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SINE_PATH = os.path.join(HERE, "golden", "soak_sine.npy")
SINE_SHA256 = "70a520195643f9ec9aa892ec529c410634068e1ffd18fa327c73b91d04261bc7"      # replaced by make_slice()
_sine = None


def sine_table():
    global _sine
    if _sine is None:
        t = np.load(SINE_PATH)
        assert t.dtype == np.int16 and t.shape == (4096,)
        assert hashlib.sha256(t.tobytes()).hexdigest() == SINE_SHA256, "tests/golden/soak_sine.npy is not the committed table"
        _sine = t.astype(np.int64)
    return _sine


def _osc(phase0, inc):
    """table oscillator: 32-bit phase, 12-bit table index.  inc: int or int64 array of per-sample increments"""
    n = len(inc)
    ph = (np.uint64(phase0) + np.concatenate(([np.uint64(0)], np.cumsum(inc[:-1].astype(np.uint64))))) & np.uint64(0xFFFFFFFF)
    return sine_table()[(ph >> np.uint64(20)).astype(np.int64)]


def _envelope(rng, n, lo, hi, seg=512):
    """piecewise-linear gain in [lo, hi] (Q15), one point every `seg` samples"""
    k = n // seg + 2
    pts = rng.integers(lo, hi + 1, k).astype(np.int64)
    i = np.arange(n, dtype=np.int64)
    a, r = i // seg, i % seg
    return (pts[a] * (seg - r) + pts[a + 1] * r) // seg


def programme(seed, n_hops, n_ch, sr):
    """int16 [n_hops*1024, n_ch]; integer arithmetic only"""
    rng = np.random.default_rng(int(seed))
    n = n_hops * 1024
    out = np.zeros((n, n_ch), dtype=np.int64)
    common = rng.integers(-32767, 32768, n).astype(np.int64) if rng.integers(0, 2) else None
    for ch in range(n_ch):
        x = np.zeros(n, dtype=np.int64)
        for _ in range(int(rng.integers(0, 9))):                           # partials with slow envelopes
            f_mhz = int(rng.integers(30_000, 450 * sr))                      # 30 Hz .. 0.45 sr, in mHz
            inc = np.full(n, (f_mhz << 32) // (1000 * sr), dtype=np.int64)
            amp = int(rng.integers(160, 9830))                               # 0.005 .. 0.3 of full scale (Q15)
            env = _envelope(rng, n, 0, 32767, seg=int(rng.integers(256, 4097)))
            x += (((_osc(int(rng.integers(0, 1 << 32)), inc) * env) >> 15) * amp) >> 15
        if rng.integers(0, 5) < 2:                                           # a sweep
            f0 = int(rng.integers(50_000, 2_000_000))
            f1 = int(rng.integers(2_000_000, 400 * sr))
            i = np.arange(n, dtype=np.int64)
            f = f0 + ((f1 - f0) * i) // max(n - 1, 1)
            inc = (f << 32) // (1000 * sr)
            x += (_osc(int(rng.integers(0, 1 << 32)), inc) * int(rng.integers(650, 9830))) >> 15
        kind = int(rng.integers(0, 4))
        noise = rng.integers(-32767, 32768, n).astype(np.int64) if common is None or rng.integers(0, 2) else common
        if kind == 1:                                                        # 8-tap boxcar
            c = np.concatenate(([0], np.cumsum(noise)))
            noise = (c[8:] - c[:-8])
            noise = np.concatenate((np.zeros(n - len(noise), dtype=np.int64), noise)) >> 3
        elif kind == 2:                                                      # differentiated
            noise = np.diff(noise, prepend=0)
        elif kind == 3:                                                      # gated by a slow square wave
            period = int(rng.integers(sr // 8, sr * 2))
            noise = noise * (((np.arange(n, dtype=np.int64) + ch * 977) // max(period // 2, 1)) % 2)
        shift = int(rng.integers(2, 16))                                     # bed level: -12 dB .. -90 dB
        x += noise >> shift
        step = int(rng.integers(2, 12)) * 1024                               # level steps of up to ~70 dB
        drop = int(rng.integers(0, 12))
        seg = (np.arange(n, dtype=np.int64) // step) % 3
        x = np.where(seg == 0, x, np.where(seg == 1, x >> (drop // 2), x >> drop))
        for c in rng.integers(1100, max(n - 1100, 1101), int(rng.integers(0, 6))):   # clicks and bursts
            w = int(rng.integers(4, 200))
            sign = 1 if rng.integers(0, 2) else -1
            level = int(rng.integers(6553, 31130))
            if rng.integers(0, 2):
                x[c:c + w] += sign * ((rng.integers(-32767, 32768, len(x[c:c + w])) * level) >> 15)
            else:
                x[c:c + w] += sign * level
        if rng.integers(0, 10) < 3 and n > 4000:                             # digital silence
            g = int(rng.integers(0, n - 4000))
            x[g:g + int(rng.integers(1500, 4000))] = 0
        if rng.integers(0, 20) < 3:                                          # hot master: clipping
            x = (x * int(rng.integers(512, 1537))) >> 8
        out[:, ch] = x
    return np.clip(out, -32768, 32767).astype(np.int16)


CODERS = ["scalar", "scalar_bs", "vq"]


def draw(seed, coder):
    """the slice's draw of a case: short streams (the oracle has to finish 900 of them in well under a minute and a half)"""
    rng = np.random.default_rng((int(seed) << 2) | CODERS.index(coder))
    sr = [48000, 48000, 44100, 44100, 32000, 96000][int(rng.integers(0, 6))]
    n_ch = [1, 2, 2, 2, 3][int(rng.integers(0, 5))]
    if coder == "vq":
        kbps = [48, 64, 96, 128, 192, 256][int(rng.integers(0, 6))]
        n_hops = int(rng.integers(6, 13))
    else:
        kbps = [32, 64, 96, 128, 192, 320][int(rng.integers(0, 6))]
        n_hops = int(rng.integers(8, 21))
    return dict(seed=int(seed), coder=coder, sr=sr, n_ch=n_ch, kbps=kbps, n_hops=n_hops)


def digest(pcm):
    return hashlib.sha256(np.ascontiguousarray(pcm).tobytes()).hexdigest()


def make_slice(n_seeds=800, seed0=30_000):
    """(run once, in the build container) writes the sine table and tests/golden/soak_slice.json: every case with the
    sha256 of its PCM"""
    import json
    import re
    t = np.round(32767.0 * np.sin(2.0 * np.pi * np.arange(4096) / 4096.0)).astype(np.int16)
    np.save(SINE_PATH, t)
    sha = hashlib.sha256(t.tobytes()).hexdigest()
    src = open(__file__).read()
    src = re.sub(r'SINE_SHA256 = "[0-9a-f]{64}"', f'SINE_SHA256 = "{sha}"', src)
    open(__file__, "w").write(src)
    globals()["SINE_SHA256"] = sha
    cases = []
    for s in range(seed0, seed0 + n_seeds):
        for coder in CODERS:
            c = draw(s, coder)
            c["pcm_sha256"] = digest(programme(c["seed"], c["n_hops"], c["n_ch"], c["sr"]))
            cases.append(c)
    with open(os.path.join(HERE, "golden", "soak_slice.json"), "w") as f:
        json.dump({"made_by": "tests/soak_programmes.py make_slice()", "cases": cases}, f, indent=0)
    return cases


if __name__ == "__main__":
    cs = make_slice()
    print(len(cs), "cases,", sum(c["n_hops"] * c["n_ch"] for c in cs), "channel-frames")


def oracle_side(case):
    """worker process (NumPy only, never touches the GPU): the oracle's .pac bytes of the case and the sha256 of its
    decoder's PCM -- or what it raised, as the reference would"""
    import sys
    root = os.path.dirname(HERE)
    if root not in sys.path:
        sys.path.insert(0, root)
    from oracle import pac_oracle as po, pac_oracle_vq as pv
    pcm = programme(case["seed"], case["n_hops"], case["n_ch"], case["sr"])
    vq = case["coder"] == "vq"
    try:
        pac = pv.encode_stream_vq(pcm, case["sr"], case["kbps"]) if vq else \
            po.encode_stream(pcm, case["sr"], case["kbps"], case["coder"] != "scalar")
    except Exception as e:
        return None, "raised " + type(e).__name__
    try:
        dec = pv.decode_stream_vq(pac) if vq else po.decode_stream(pac)
    except Exception as e:
        return pac, "raised " + type(e).__name__
    return pac, digest(dec)
