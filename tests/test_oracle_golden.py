"""Pins oracle/pac_oracle.py (the CPU restatement) to vectors produced by the
reference itself (tests/golden/make_golden.py).  Everything here is bit-exact:
the oracle reaches NumPy through the same calls as the reference."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import EXCERPTS, GOLDEN, load_excerpt
from oracle import pac_oracle as po


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.array_equal(a, b), f"max |d| = {np.max(np.abs(a - b))}"


# ------------------------------------------------------------------ tables
@pytest.mark.parametrize("n_lines", [1024, 128, 512])
@pytest.mark.parametrize("sr", [48000, 44100])
def test_band_tables(tables, n_lines, sr):
    b = po.band_table(n_lines, sr)
    key = f"bands_{n_lines}_{sr}"
    same(b.nLines, tables[key + "_nLines"])
    same(b.lowerLine, tables[key + "_lower"])
    same(b.upperLine, tables[key + "_upper"])
    same(po.omitted_bands(b), tables[key + "_omitted"])
    assert int(np.sum(b.nLines)) == n_lines


def test_band_tables_known_counts():
    # SURVEY.md section 8a row A7 [measured]
    assert po.band_table(1024, 48000).nLines.tolist() == [
        13, 14, 19, 17, 22, 14, 16, 19, 24, 30, 38, 47, 56, 76, 107, 149, 363]
    assert po.band_table(1024, 44100).nLines.tolist() == [
        14, 15, 14, 16, 21, 13, 15, 17, 21, 26, 32, 42, 51, 61, 83, 116, 163, 304]
    assert po.band_table(128, 48000).nLines.tolist() == [14, 14, 13, 23, 19, 45]


def test_windows(tables):
    same(po.sine_window(2048), tables["win_sine_2048"])
    same(po.sine_window(256), tables["win_sine_256"])
    same(po.hann_window(2048), tables["win_hann_2048"])
    same(po.hann_window(256), tables["win_hann_256"])
    same(po.start_window(2048, 256), tables["win_start_2048"])
    same(po.stop_window(2048, 256), tables["win_stop_2048"])
    same(po.start_stop_window(2048, 256), tables["win_startstop_2048"])


def test_kbd_window_and_its_mdct():
    """window.KBDWindow (coder/window.py:45-57) and MDCT(KBDWindow(x), N//2, N//2), the
    expression at coder/bitalloc.py:161, against the reference run by make_golden.py --kbd."""
    g = np.load(os.path.join(GOLDEN, "kbd.npz"))
    for n in (2048, 256, 1024):
        same(po.kbd_window(n), g[f"kbd_{n}"])
        same(po.kbd_window(n) * g[f"x_{n}"], g[f"kbd_x_{n}"])
        same(po.mdct_forward(po.kbd_window(n) * g[f"x_{n}"], n // 2, n // 2), g[f"mdct_kbd_x_{n}"])
    same(po.kbd_window(2048, 2.5), g["kbd_2048_alpha2p5"])
    same(po.kbd_window(512), g["kbd_512_alpha4"])
    same(po.kbd_window(2048) * g["rand_x"], g["rand_kbd_x"])


def test_window_kind_priority():
    K = po.window_kind
    assert K(0, 0, 0) == po.WINDOW_SINE and K(1, 1, 1) == po.WINDOW_SINE
    assert K(1, 0, 1) == po.WINDOW_STARTSTOP
    assert K(1, 0, 0) == po.WINDOW_STOP and K(0, 0, 1) == po.WINDOW_START


@pytest.mark.parametrize("sr", [48000, 44100])
@pytest.mark.parametrize("n_lines", [1024, 128])
def test_bark_thresh(tables, sr, n_lines):
    f = sr / (2 * n_lines) * (np.arange(n_lines) + 0.5)
    same(po.bark_of(f.copy()), tables[f"bark_{n_lines}_{sr}"])
    same(po.thresh_quiet(f.copy()), tables[f"thresh_{n_lines}_{sr}"])


def test_pcm_contract(tables):
    got = po.pcm16_to_fraction(np.arange(-32768, 32768))
    same(got, tables["pcm_all_fraction"])
    assert np.array_equal(np.signbit(got), np.signbit(tables["pcm_all_fraction"]))
    assert got[0] == 0.0 and got[-1] == 2 * 32767 / 65535


# ------------------------------------------- the reference's own self tests
def test_mdct_self_test(tables):
    x = tables["mdct_ramp_in"]
    same(po.mdct_forward(x, 10, 10), tables["mdct_ramp_fast"])
    same(po.mdct_slow(x, 10, 10), tables["mdct_ramp_slow"])
    same(po.mdct_inverse(tables["mdct_ramp_fast"], 10, 10), tables["imdct_ramp_fast"])
    assert np.allclose(po.mdct_forward(x, 10, 10), po.mdct_slow(x, 10, 10))


def test_mdct_tdac_round_trip():
    # coder/mdct.py:86-98: overlap-add of IMDCT(MDCT)/2 rebuilds the signal
    x = np.array([0, 1, 2, 3, 4, 4, 4, 4, 3, 1, -1, -3], dtype=float)
    x = np.concatenate([np.zeros(4), x, np.zeros(4)])
    y = np.zeros_like(x)
    for i in range(0, len(x) - 4, 4):
        y[i:i + 8] += po.mdct_inverse(po.mdct_forward(x[i:i + 8], 4, 4), 4, 4) / 2
    assert np.allclose(x, y)


def test_quantizer_self_test(tables):
    q = tables["quant_in"]
    for bits in (8, 12):
        same([po.quantize_uniform(v, bits) for v in q], tables[f"quant_u{bits}"])
        same(po.quantize_uniform_vec(q, bits), tables[f"quant_v{bits}"])
        same(po.dequantize_uniform_vec(po.quantize_uniform_vec(q, bits), bits),
             tables[f"dequant_v{bits}"])
    same([po.scale_factor(v) for v in q], tables["quant_scale_3_5"])
    same([po.mantissa_vec(np.array([v]), po.scale_factor(v))[0] for v in q],
         tables["quant_mant_3_5"])
    same([po.dequantize_vec(po.scale_factor(v), po.mantissa_vec(
        np.array([v]), po.scale_factor(v)))[0] for v in q], tables["quant_deq_3_5"])


def test_scale_factor_sweep(tables):
    for mb in (5, 0, 2, 7, 16):
        same([po.scale_factor(v, 4, mb) for v in tables["sf_sweep_in"]],
             tables[f"sf_sweep_4_{mb}"])
    assert po.scale_factor(0.0, 4, 0) == 14      # silent band, ba=0
    assert po.scale_factor(0.0, 4) == 15


def test_bitpack_demo(tables):
    bw = po.BitWriter(2)
    for v, w in zip((3, 5, 11, 3, 1), (4, 3, 5, 3, 1)):
        bw.put(v, w)
    assert bw.bytes() == bytes(tables["bitpack_demo"]) == b":\xb7"
    br = po.BitReader(bw.bytes())
    assert [br.get(w) for w in (4, 3, 5, 3, 1)] == [3, 5, 11, 3, 1]


# ------------------------------------------------------------ per-stage pins
def _params_for(stages, kind, i):
    sr = int(stages[f"{kind}_sr"][i])
    p = po.make_params(sr, 1, int(stages[f"{kind}_kbps"][i]))
    if kind == "short":
        p.nMDCTLines = p.nSamplesPerBlock = 128
    return p


@pytest.mark.parametrize("kind", ["long", "short"])
def test_stage_vectors(stages, kind):
    n = len(stages[f"{kind}_sr"])
    n_lines = 1024 if kind == "long" else 128
    for i in range(n):
        tag = str(stages[f"{kind}_tag"][i])
        flags = [bool(f) for f in stages[f"{kind}_flags"][i]]
        p = _params_for(stages, kind, i)
        x = po.pcm16_to_fraction(stages[f"{kind}_x_i16"][i])
        same(x, stages[f"{kind}_x"][i])
        st = {}
        sf, ba, mant, ov = po.encode_channel(x.copy(), p, *flags, stages=st)
        same(st["windowed"], stages[f"{kind}_windowed"][i])
        same(st["mdct"], stages[f"{kind}_mdct"][i])
        assert ov == int(stages[f"{kind}_overall"][i]), tag
        nb = int(stages[f"{kind}_nbands"][i])
        same(st["smr"], stages[f"{kind}_smr"][i][:nb])
        same(ba, stages[f"{kind}_ba"][i][:nb])
        same(sf, stages[f"{kind}_sf"][i][:nb])
        nm = int(stages[f"{kind}_n_mant"][i])
        assert len(mant) == nm, tag
        same(mant, stages[f"{kind}_mant"][i][:nm])
        assert sf.dtype == np.int32 and mant.dtype == np.int32
        # side chain
        inten = po.sidechain_intensity(x)
        same(inten, stages[f"{kind}_inten"][i])
        pf, ps = po.find_peaks(inten, np.fft.rfftfreq(2 * n_lines, d=1 / p.sampleRate))
        npk = int(stages[f"{kind}_n_peaks"][i])
        assert len(pf) == npk, tag
        same(np.array(pf, dtype=float), stages[f"{kind}_pk_f"][i][:npk])
        same(np.array([float(s) for s in ps]), stages[f"{kind}_pk_spl"][i][:npk])
        same(po.masked_threshold(x, n_lines, p.sampleRate), stages[f"{kind}_thr"][i])


def test_known_answers_48k(stages):
    # SURVEY.md section 8c [measured]
    tags = [str(t) for t in stages["long_tag"]]
    i = tags.index("zeros")
    assert stages["long_overall"][i] == 15
    assert stages["long_ba"][i][:17].tolist() == [4, 4, 5, 5, 5, 5, 6, 6, 6, 6, 5, 5, 5, 4, 4, 0, 0]
    assert not stages["long_mant"][i].any()


# --------------------------------------------------------------- file level
@pytest.mark.parametrize("name", EXCERPTS)
def test_excerpt_pac_long_only(name):
    ex = load_excerpt(name)
    got = po.encode_stream(ex["pcm"], int(ex["sr"]), 128, block_switching=False)
    assert got == bytes(ex["pac_long"])


@pytest.mark.parametrize("name", ["castanet", "spmg"])
def test_excerpt_pac_block_switched(name):
    ex = load_excerpt(name)
    seen = []
    got = po.encode_stream(ex["pcm"], int(ex["sr"]), 128, block_switching=True,
                           collect=seen)
    assert got == bytes(ex["pac_bs"])
    flags = np.array([[int(bool(f)) for f in fl] for fl, _ in seen[:-1]])
    same(flags, ex["flags_bs"])


def test_excerpt_pac_96k():
    ex = load_excerpt("harpsichord")
    got = po.encode_stream(ex["pcm"], int(ex["sr"]), 96, block_switching=False)
    assert got == bytes(ex["pac_long96"])


@pytest.mark.parametrize("name", EXCERPTS)
def test_decode_matches_reference_decoder(name):
    """oracle.decode_stream vs the PCM the reference's own decoder produced
    from the golden excerpt .pac files (tests/golden/decoded_*.npz)."""
    ex = load_excerpt(name)
    d = np.load(os.path.join(GOLDEN, f"decoded_{name}.npz"))
    for tag in ("long", "bs"):
        got = po.decode_stream(bytes(ex[f"pac_{tag}"]))
        assert got.dtype == np.int16 and np.array_equal(got, d[f"pcm_{tag}"]), (name, tag)


def test_numpy_sum_order():
    """np.sum on a contiguous float64 vector = 8-accumulator pairwise scheme;
    the HIP bit allocator (csrc) re-implements exactly this order."""
    rng = np.random.default_rng(3)

    def pw(a):
        n = len(a)
        if n < 8:
            r = -0.0
            for v in a:
                r = r + v
            return r
        r = [a[j] for j in range(8)]
        i = 8
        while i < n - (n % 8):
            for j in range(8):
                r[j] = r[j] + a[i + j]
            i += 8
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
        while i < n:
            res = res + a[i]
            i += 1
        return res
    for n in range(1, 27):
        for _ in range(50):
            a = rng.standard_normal(n) * 10.0 ** rng.integers(-6, 6, n)
            assert pw([np.float64(v) for v in a]) == np.sum(a)


@pytest.mark.skipif(not os.path.exists("/root/reference/test_signals"),
                    reason="reference WAVs only exist in the build container")
@pytest.mark.skipif(not os.environ.get("PACX_FULLFILE"),
                    reason="minutes of CPU; set PACX_FULLFILE=1")
@pytest.mark.parametrize("name", EXCERPTS)
def test_full_file_hash(name):
    want = json.load(open(os.path.join(GOLDEN, "fullfile.json")))
    raw = open(f"/root/reference/test_signals/{name}.wav", "rb").read()
    sr, pcm, declared = po.wav_effective_stream(raw)
    for tag, bs in (("long", False), ("bs", True)):
        got = po.encode_stream(pcm, sr, 128, block_switching=bs, header_samples=declared)
        assert hashlib.sha256(got).hexdigest() == want[f"{name}:{tag}"]["sha256"]


# ------------------------------------------- scalar mantissas + SBR (useVQ off, useSBR on)
def _sbr_scalar_cases():
    import json
    return json.load(open(os.path.join(GOLDEN, "sbr_scalar.json")))


@pytest.mark.parametrize("case", _sbr_scalar_cases(), ids=lambda e: f"{e['excerpt']}_{e['kbps_per_channel']}")
def test_scalar_sbr_restatement_follows_the_reference(case):
    """tests/golden/sbr_scalar.json holds what the reference's own file loop does with useVQ off and
    useSBR on (make_golden.py --sbr-scalar): whole-file hashes where it finishes, its exception where
    an omitted band receives bits (coder/codec.py:541-546 -> coder/quantize.py:73-74)."""
    import hashlib
    ex = np.load(os.path.join(GOLDEN, f"excerpt_{case['excerpt']}.npz"))
    pcm, sr = ex["pcm"][:case["hops"] * 1024], int(ex["sr"])
    if case["outcome"] == "raised":
        assert case["error"] == "TypeError: " + po.REF_SCALAR_SBR_ERROR
        assert case["where"][-1].startswith("coder/quantize.py:74")
        with pytest.raises(TypeError, match="item assignment"):
            po.encode_stream(pcm, sr, case["kbps_per_channel"], case["block_switching"], use_sbr=True)
    else:
        pac = po.encode_stream(pcm, sr, case["kbps_per_channel"], case["block_switching"], use_sbr=True)
        assert len(pac) == case["bytes"]
        assert hashlib.sha256(pac).hexdigest() == case["sha256"]
        # the reference's decoder on its own file (no omitted band is coded: plain Decode)
        dec = po.decode_stream(pac)
        assert list(dec.shape) == case["decoded_shape"]
        assert hashlib.sha256(np.ascontiguousarray(dec).astype("<i2").tobytes()).hexdigest() == case["decoded_sha256"]
