"""CPU checks of audio-codec_amd/csrc/pacx_exact.h (the order-sensitive scalar
arithmetic shared by the HIP kernels) against the oracle: the header is built
for the host with g++ and driven through ctypes.  No GPU needed."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from oracle import pac_oracle as po

SRC = os.path.join(ROOT, "tests", "hostcheck", "hostcheck.cpp")
INC = os.path.join(ROOT, "audio-codec_amd", "csrc")


@pytest.fixture(scope="module")
def hc(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("hostcheck") / "libhostcheck.so")
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-fPIC", "-shared",
                           "-I", INC, SRC, "-o", out])
    lib = ctypes.CDLL(out)
    lib.hc_pcm16_to_f64.restype = ctypes.c_double
    lib.hc_pcm16_to_f64.argtypes = [ctypes.c_int]
    lib.hc_quant_mag.restype = ctypes.c_longlong
    lib.hc_quant_mag.argtypes = [ctypes.c_double, ctypes.c_int]
    lib.hc_scale_factor.argtypes = [ctypes.c_double, ctypes.c_int, ctypes.c_int]
    lib.hc_mantissa.argtypes = [ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.hc_np_sum.restype = ctypes.c_double
    lib.hc_np_sum.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.hc_bit_budget.restype = ctypes.c_double
    lib.hc_bit_budget.argtypes = [ctypes.c_double] + [ctypes.c_int] * 6
    lib.hc_bit_alloc.argtypes = [ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                 ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    for f in ("hc_spl_array", "hc_spl_scalar", "hc_bark", "hc_thresh_quiet", "hc_round_trip"):
        getattr(lib, f).restype = ctypes.c_double
        getattr(lib, f).argtypes = [ctypes.c_double]
    lib.hc_dequant_uniform.restype = ctypes.c_double
    lib.hc_dequant_uniform.argtypes = [ctypes.c_longlong, ctypes.c_int]
    for f in ("hc_dequantize", "hc_dequantize_fp"):
        getattr(lib, f).restype = ctypes.c_double
        getattr(lib, f).argtypes = [ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.hc_mantissa_fp.argtypes = [ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.hc_window_kind.argtypes = [ctypes.c_uint]
    lib.hc_quant_guard.argtypes = [ctypes.c_double, ctypes.c_int, ctypes.c_double]
    lib.hc_scale_guard.argtypes = [ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_double]
    return lib


def c_bit_alloc(hc, budget, max_mant, n_lines, smr):
    n_lines = np.ascontiguousarray(n_lines, dtype=np.int32)
    smr = np.ascontiguousarray(smr, dtype=np.float64)
    bits = np.zeros(len(n_lines), dtype=np.int32)
    cap = ctypes.c_int(0)
    passes = hc.hc_bit_alloc(budget, max_mant, len(n_lines), n_lines.ctypes.data,
                             smr.ctypes.data, bits.ctypes.data, ctypes.byref(cap))
    return bits, passes, cap.value


def test_pcm_all_codes(hc, tables):
    got = np.array([hc.hc_pcm16_to_f64(int(c)) for c in range(-32768, 32768)])
    want = tables["pcm_all_fraction"]
    assert np.array_equal(got, want)
    assert np.array_equal(np.signbit(got), np.signbit(want))


def test_scale_factor_and_mantissa(hc, tables):
    sweep = tables["sf_sweep_in"]
    for mb in (5, 0, 2, 7, 16):
        got = [hc.hc_scale_factor(abs(float(v)), 4, mb) for v in sweep]
        assert got == tables[f"sf_sweep_4_{mb}"].tolist()
    rng = np.random.default_rng(11)
    for _ in range(300):
        ba = int(rng.integers(2, 17))
        mag = 10.0 ** rng.uniform(-7, 0.02)
        x = rng.uniform(-1, 1, 64) * mag
        x[0] = 0.0
        x[1] = -0.0
        x[2] = -1e-300
        sc = po.scale_factor(np.max(np.abs(x)), 4, ba)
        assert sc == hc.hc_scale_factor(float(np.max(np.abs(x))), 4, ba)
        want = po.mantissa_vec(x, sc, 4, ba)
        got = [hc.hc_mantissa(float(v), sc, 4, ba) for v in x]
        assert got == want.tolist()


def test_dequantizers(hc, tables):
    """pacx_dequant_uniform / pacx_dequantize (decode path and the quantize.py mirrors) against the oracle's
    restatement of coder/quantize.py:82-95, 254-274 and the reference's own self-test outputs"""
    q_in = tables["quant_in"]
    for bits in (8, 12):
        codes = tables[f"quant_v{bits}"]
        got = [hc.hc_dequant_uniform(int(c), bits) for c in codes]
        assert got == tables[f"dequant_v{bits}"].tolist()
    got = [hc.hc_dequantize(int(m), int(s), 3, 5) for m, s in zip(tables["quant_mant_3_5"], tables["quant_scale_3_5"])]
    assert got == tables["quant_deq_3_5"].tolist()
    rng = np.random.default_rng(5)
    for nsb, nmb in ((4, 2), (4, 7), (4, 16), (3, 5)):
        r = (1 << nsb) - 1 + nmb
        codes = rng.integers(0, 1 << r, 500)
        assert [hc.hc_dequant_uniform(int(c), r) for c in codes] == po.dequantize_uniform_vec(codes, r).tolist()
        mant = rng.integers(0, 1 << nmb, 300)
        for scale in range(1 << nsb):
            want = po.dequantize_vec(scale, mant, nsb, nmb)
            assert [hc.hc_dequantize(int(m), scale, nsb, nmb) for m in mant] == want.tolist()
    assert hc.hc_dequant_uniform(1 << 7, 8) == 0.0 and np.signbit(hc.hc_dequant_uniform(1 << 7, 8)) == False  # -0 code -> +0.0


def test_np_sum_order(hc):
    rng = np.random.default_rng(5)
    for n in range(0, 27):
        for _ in range(40):
            a = np.ascontiguousarray(rng.standard_normal(n) * 10.0 ** rng.integers(-6, 6, n))
            assert hc.hc_np_sum(a.ctypes.data, n) == np.sum(a)


def test_window_kind(hc):
    for f in range(8):
        assert hc.hc_window_kind(f) == po.window_kind(f & 1, (f >> 1) & 1, (f >> 2) & 1)


def test_bit_budget(hc):
    for sr in (48000, 44100):
        for kbps in (128, 96):
            p = po.make_params(sr, 1, kbps)
            for flags in range(8):
                last, cur, nxt = flags & 1, (flags >> 1) & 1, (flags >> 2) & 1
                if cur:
                    p.nMDCTLines = 128
                want = po.bit_budget(p, last, cur, nxt)
                nb = (p.sfBandsShort if cur else p.sfBands).nBands
                got = hc.hc_bit_budget(p.targetBitsPerSample, p.nMDCTLines, cur,
                                       int(bool(last or nxt)), 4, 12, nb)
                p.nMDCTLines = 1024
                assert got == want


def test_bit_alloc_on_golden_smrs(hc, stages):
    for kind in ("long", "short"):
        for i in range(len(stages[f"{kind}_sr"])):
            sr = int(stages[f"{kind}_sr"][i])
            p = po.make_params(sr, 1, int(stages[f"{kind}_kbps"][i]))
            if kind == "short":
                p.nMDCTLines = 128
            flags = [bool(f) for f in stages[f"{kind}_flags"][i]]
            bands = p.sfBandsShort if kind == "short" else p.sfBands
            nb = bands.nBands
            budget = po.bit_budget(p, *flags)
            bits, _, cap = c_bit_alloc(hc, budget, 16, bands.nLines, stages[f"{kind}_smr"][i][:nb])
            assert bits.tolist() == stages[f"{kind}_ba"][i][:nb].tolist()
            assert cap == 0


def test_bit_alloc_random(hc):
    rng = np.random.default_rng(21)
    bands = [po.band_table(1024, 48000), po.band_table(1024, 44100), po.band_table(128, 48000)]
    n_cap = 0
    for t in range(3000):
        b = bands[t % 3]
        spread = rng.choice([3.0, 15.0, 40.0, 90.0])
        smr = rng.standard_normal(b.nBands) * spread + rng.uniform(-30, 30)
        if t % 7 == 0:
            smr = np.round(smr)               # ties in the rounding ladder
        if t % 11 == 0:
            smr[:] = smr[0]
        budget = float(rng.choice([2454.6666666666665, 2044.0, 1772.0, 157 * 2.9, 300.0, 12000.0]))
        want = po.bit_alloc(budget, 16, b.nBands, b.nLines, smr)
        got, passes, cap = c_bit_alloc(hc, budget, 16, b.nLines, smr)
        n_cap += cap
        assert got.tolist() == want.tolist(), (t, smr.tolist(), budget)
    # all-dropped corner: every band below two bits
    b = bands[0]
    smr = np.full(b.nBands, -200.0)
    assert c_bit_alloc(hc, 10.0, 16, b.nLines, smr)[0].tolist() == \
        po.bit_alloc(10.0, 16, b.nBands, b.nLines, smr).tolist()


def test_masker_round_trip(hc):
    """SPL(Intensity(x)) as the kernels evaluate it (one exp2 + log1p series)
    against the reference's pow/log10 evaluation, over the whole range a masker
    curve can take, the -30 dB floor included."""
    x = np.concatenate((np.linspace(-800, 120, 20001), np.linspace(-31.5, -29.5, 4001),
                        [-30.0, -30.57, -30.566, 96.0, 0.0]))
    want = po.spl_of(po.intensity_of(x.copy()))
    got = np.array([hc.hc_round_trip(float(v)) for v in x])
    assert np.max(np.abs(got - want)) < 5e-13
    assert np.all(got >= -30.0) and np.all(got[x < -30.6] == -30.0)


def test_lean_exp2(hc):
    """pacx_exp2_lean (the mask kernel's per-line 2^y) against np.exp2: relative 3e-13 over the
    range the round trip uses and well beyond"""
    hc.hc_exp2_lean.restype = ctypes.c_double
    hc.hc_exp2_lean.argtypes = [ctypes.c_double]
    rng = np.random.default_rng(9)
    y = np.concatenate((rng.uniform(-60, 300, 20000), np.linspace(-0.5, 0.5, 2001), [0.0, 1.0, -1.0, 0.5, -0.5, 1000.0]))
    got = np.array([hc.hc_exp2_lean(float(v)) for v in y])
    want = np.exp2(y)
    assert np.max(np.abs(got / want - 1.0)) < 3e-13
    assert hc.hc_exp2_lean(0.0) == 1.0 and hc.hc_exp2_lean(10.0) == 1024.0


def test_psycho_scalars(hc, tables):
    rng = np.random.default_rng(2)
    v = 10.0 ** rng.uniform(-20, 2, 500)
    want = po.spl_of(v.copy())
    got = np.array([hc.hc_spl_array(float(x)) for x in v])
    assert np.max(np.abs(got - want)) < 1e-12
    assert hc.hc_spl_array(0.0) == po.spl_of(np.array([0.0]))[0]
    assert hc.hc_spl_scalar(0.0) == -30 and hc.hc_spl_scalar(1e-300) == -30
    f = 48000 / 2048 * (np.arange(1024) + 0.5)
    got = np.array([hc.hc_bark(float(x)) for x in f])
    assert np.max(np.abs(got - tables["bark_1024_48000"])) < 1e-13
    got = np.array([hc.hc_thresh_quiet(float(x)) for x in f])
    assert np.max(np.abs(got - tables["thresh_1024_48000"]) / np.abs(tables["thresh_1024_48000"])) < 1e-12


def test_lean_log10_within_one_ulp(hc):
    """pacx_log10_pos (the mask kernel's per-line log10): against np.log10 evaluated in
    extended precision, over the range its argument takes (eps .. 4*8^2) and at the
    powers of ten; the reference's own np.log10 is the host libm's, itself 1-2 ulp."""
    hc.hc_log10_pos.restype = ctypes.c_double
    hc.hc_log10_pos.argtypes = [ctypes.c_double]
    rng = np.random.default_rng(11)
    x = np.concatenate((10.0 ** rng.uniform(-16, 3, 20000), 1.0 + rng.uniform(-0.3, 0.45, 5000),
                        10.0 ** np.arange(-15, 3), [2.220446049250313e-16, 1e-8 + 2.220446049250313e-16]))
    got = np.array([hc.hc_log10_pos(float(v)) for v in x])
    want = (np.log(x.astype(np.longdouble)) / np.log(np.longdouble(10))).astype(np.longdouble)
    ulp = np.spacing(np.abs(want.astype(np.float64))).astype(np.longdouble)
    # values next to log10(1) = 0 are measured against the spacing of the argument's distance from 1
    err = np.abs(got.astype(np.longdouble) - want) / np.maximum(ulp, np.longdouble(1e-300))
    assert float(err.max()) <= 1.0, float(err.max())
    assert hc.hc_log10_pos(1.0) == 0.0 and hc.hc_log10_pos(100.0) == 2.0


# ------------------------------------------------- gain-shape (PVQ) host tables
def test_vq_budget_rule(hc):
    """coder/codec.py:292-294 (VQ) and :446-453 (SBR long block: flags ignored)."""
    hc.hc_bit_budget_vq.restype = ctypes.c_double
    hc.hc_bit_budget_vq.argtypes = [ctypes.c_double] + [ctypes.c_int] * 7
    for sr in (48000, 44100):
        for kbps in (128, 96):
            tbps = kbps / (sr / 1000)
            nb_long = po.band_table(1024, sr).nBands
            nb_short = po.band_table(128, sr).nBands
            for flags in range(8):
                last, cur, nxt = flags & 1, (flags >> 1) & 1, (flags >> 2) & 1
                half_n = 128 if cur else 1024
                n_eff = int(1.45 * half_n) if cur else half_n
                if last or nxt:
                    n_eff = int(0.85 * n_eff)
                nb = nb_short if cur else nb_long
                want = tbps * n_eff
                want -= 4
                want -= 12 * nb
                got = hc.hc_bit_budget_vq(tbps, half_n, cur, int(bool(last or nxt)), 4, 12, nb, 0)
                assert got == want
                if not cur:
                    want = tbps * 1024
                    want -= 4
                    want -= 12 * nb
                    assert hc.hc_bit_budget_vq(tbps, 1024, 0, int(bool(last or nxt)), 4, 12, nb, 1) == want


def test_vq_tables_match_oracle(hc):
    from oracle import pac_oracle_vq as pv
    l_max = 363
    hc.hc_vq_build(l_max)
    hc.hc_vq_n.restype = ctypes.c_ulonglong
    hc.hc_vq_p.restype = ctypes.c_ulonglong
    for l in list(range(2, 40)) + [41, 64, 65, 91, 128, 149, 182, 304, 363]:
        for bits in (1, 2, 3, 7, 8, 9, 15, 16, 17, 24, 31, 32):
            k, w = pv.pulses_for_bits(l, bits) if l > 2 or bits <= 20 else (2 ** (bits - 2), bits)
            assert (hc.hc_vq_k(l, bits), hc.hc_vq_w(l, bits)) == (k, w), (l, bits)
    for bits in range(1, 33):
        assert hc.hc_vq_k(1, bits) == -1            # the reference never returns for L = 1
    for l in (3, 4, 5, 7, 13, 29, 100, 363):
        n = hc.hc_vq_row_len(l)
        assert pv.codebook_size(l, n - 1) > 2 ** 32 >= pv.codebook_size(l, n - 2)
        run = 0
        for k in list(range(min(n, 40))) + ([n - 1] if n > 40 else []):
            assert hc.hc_vq_n(l, k) == pv.codebook_size(l, k)
        for k in range(min(n, 300)):
            run += pv.codebook_size(l, k)
            assert hc.hc_vq_p(l, k) == run


def test_guard_band_of_the_quantiser(hc):
    """PACX_ST_GUARD helpers: a magnitude is flagged exactly when a perturbation of up to `err`
    can change its code (coder/quantize.py:73: floor(((2^R - 1) ax + 1) / 2)), and the
    ScaleFactor guard only at the boundaries where the leading-zero count changes."""
    rng = np.random.default_rng(5)
    err = 5e-13
    for r_bits in (15, 17, 20, 24, 31):
        s = float((1 << r_bits) - 1)
        ax = rng.uniform(0, 1, 4000)
        # some values parked right at boundaries: t = even integer +- tiny
        k = rng.integers(1, 1 << (r_bits - 1), 400).astype(np.float64)
        ax[:400] = (2 * k - 1) / s * (1 + rng.uniform(-1, 1, 400) * 1e-15)
        for a in ax:
            a = float(min(max(a, 0.0), 0.999999))
            lo, hi = max(a - err, 0.0), a + err
            changes = hc.hc_quant_mag(lo, r_bits) != hc.hc_quant_mag(hi, r_bits)
            flagged = hc.hc_quant_guard(a, r_bits, err)
            if changes:
                assert flagged, (a, r_bits)          # never misses a code that can move
            if not flagged:
                assert hc.hc_quant_mag(lo, r_bits) == hc.hc_quant_mag(a, r_bits) == hc.hc_quant_mag(hi, r_bits)
    assert hc.hc_quant_guard(1.0, 20, err) and hc.hc_quant_guard(1.0 - 1e-13, 20, err)
    assert not hc.hc_quant_guard(1.5, 20, err) and not hc.hc_quant_guard(0.3, 10, err)
    # ScaleFactor: 4 scale bits, 5 mantissa bits -> R = 20; boundaries where the code is a power of two
    for j in range(4, 18):                                       # below 2^4 the count is capped at 15
        edge = (2.0 * (1 << j) - 1) / float((1 << 20) - 1)       # t = 2 * 2^j exactly
        assert hc.hc_scale_factor(edge * (1 - 1e-12), 4, 5) != hc.hc_scale_factor(edge * (1 + 1e-12), 4, 5)
        assert hc.hc_scale_guard(edge, 4, 5, err)
        assert not hc.hc_scale_guard(edge * 1.37, 4, 5, err)
