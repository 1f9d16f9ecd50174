"""The gain-shape (PVQ) / SBR oracle against the reference's own outputs.

Pins (tests/golden/make_golden.py --vq, run in the build container):
  * excerpt_vq_<wav>.npz -- 24-hop excerpts through the reference's shipped
    configuration (useVQ, useSBR below 128 kb/s, block switching);
  * vqfile.json -- sha256 of whole-file encodes by the reference run here, next
    to the sha256 of the .pac files the reference itself committed under
    test_decoded_full/ (6 of 8 identical; quar48_1 differs in the four
    rounding-noise-decided short blocks also seen on the scalar path).
"""
import hashlib
import itertools
import json
import os

import numpy as np
import pytest

from conftest import EXCERPTS, GOLDEN
from oracle import pac_oracle_vq as pv


def _brute(l, k):
    return [v for v in itertools.product(range(-k, k + 1), repeat=l)
            if sum(abs(c) for c in v) == k]


@pytest.mark.parametrize("l,k", [(1, 1), (1, 4), (2, 3), (3, 2), (3, 5), (4, 3), (5, 2)])
def test_codebook_size_and_index_bijection(l, k):
    vecs = _brute(l, k)
    assert pv.codebook_size(l, k) == len(vecs)
    idx = sorted(pv.pvq_index(np.array(v), k) for v in vecs)
    assert idx == list(range(len(vecs)))


def test_pulses_for_bits_known_answers():
    # N(2,K) = 4K, N(3,K) = 4K^2 + 2
    assert pv.pulses_for_bits(2, 10) == (256, 10)
    assert pv.pulses_for_bits(3, 32)[0] == 32767
    # nothing fits: K = 0 still costs one bit (ceil(log2(1 + eps)) = 1)
    assert pv.pulses_for_bits(363, 9) == (0, 1)
    for l, bits in [(4, 30), (7, 32), (13, 17), (29, 32), (182, 20)]:
        k, w = pv.pulses_for_bits(l, bits)
        assert pv.codebook_size(l, k) <= 2 ** bits < pv.codebook_size(l, k + 1)
        assert w == (pv.codebook_size(l, k) - 1).bit_length() or pv.codebook_size(l, k) == 1


def test_search_keeps_pulse_count_and_signs():
    rng = np.random.default_rng(5)
    for l, k in [(4, 700), (13, 9), (29, 6), (91, 3)]:
        x = rng.standard_normal(l)
        x /= np.linalg.norm(x)
        y = pv.pvq_search(x, k)
        assert np.sum(np.abs(y)) == k
        assert np.all((y == 0) | (np.sign(y) == np.sign(x)))


def test_gain_shape_uses_the_whole_band_budget():
    rng = np.random.default_rng(6)
    for l, ba in [(13, 2), (13, 16), (47, 5), (149, 3), (363, 2), (363, 7)]:
        x = rng.standard_normal(l) * 0.01
        idx, bits = pv.quantize_gain_shape(x, ba * l)
        assert sum(bits) == ba * l
        assert all(0 <= i < (1 << w) if w else i == 0 for i, w in zip(idx, bits))
    assert pv.quantize_gain_shape(np.zeros(13), 26) == ([0], [0])


@pytest.mark.parametrize("name", EXCERPTS)
@pytest.mark.parametrize("kbps", [128, 96])
def test_excerpt_matches_reference(name, kbps):
    ex = np.load(os.path.join(GOLDEN, f"excerpt_{name}.npz"))
    gold = np.load(os.path.join(GOLDEN, f"excerpt_vq_{name}.npz"))
    hops = int(gold["hops"])
    carries = []
    out = pv.encode_stream_vq(ex["pcm"][:hops * 1024], int(ex["sr"]), kbps,
                              carries=carries)
    assert out == bytes(gold[f"pac_vq{kbps}"])
    # the SBR spill rule (coder/codec.py:522-524) compares a band's total bits
    # with its per-line allocation and so never fires on real material
    assert carries == []


def test_vqfile_records():
    rec = json.load(open(os.path.join(GOLDEN, "vqfile.json")))
    assert len(rec) == 8
    for key, r in rec.items():
        assert r["size"] == r["committed_size"]
        if key.startswith("quar48_1"):
            assert r["blocks_differing_from_committed"] == [978, 979, 1026, 1027]
        else:
            assert r["sha256"] == r["committed_sha256"]
            assert r["blocks_differing_from_committed"] == []


@pytest.fixture(scope="module")
def harpsichord_96_pac():
    full = np.load(os.path.join(GOLDEN, "full_harpsichord.npz"))
    return pv.encode_stream_vq(full["pcm"], int(full["sr"]), 96,
                               header_samples=int(full["declared"]))


def test_whole_file_matches_reference_committed_pac(harpsichord_96_pac):
    """harpsichord at 96 kb/s (SBR + PVQ): the oracle reproduces, byte for
    byte, the .pac file the reference's author committed."""
    rec = json.load(open(os.path.join(GOLDEN, "vqfile.json")))["harpsichord:96"]
    assert len(harpsichord_96_pac) == rec["committed_size"]
    assert hashlib.sha256(harpsichord_96_pac).hexdigest() == rec["committed_sha256"]


# ------------------------------------------------------------------ decode side
@pytest.mark.parametrize("l,k", [(2, 5), (3, 4), (4, 3), (5, 2)])
def test_pvq_decode_inverts_index(l, k):
    for v in _brute(l, k):
        assert tuple(pv.pvq_decode(pv.pvq_index(np.array(v), k), l, k)) == v


def test_scipy_restatements_are_bit_exact():
    """gaussian_filter1d(sigma=200) and interp1d(kind='slinear') as Decode_SBR
    calls them (coder/codec.py:147, 183) against SciPy itself."""
    from scipy import interpolate
    from scipy.ndimage import gaussian_filter1d
    rng = np.random.default_rng(0)
    x = np.zeros(1024)
    x[0], x[1] = 0.3, 0.01
    want = gaussian_filter1d(x, sigma=200)
    assert np.array_equal(pv.gaussian_smooth(x), want)
    assert np.array_equal(pv.gaussian_smooth_sparse(x), want)
    x = rng.standard_normal(608)
    assert np.array_equal(pv.gaussian_smooth(x), gaussian_filter1d(x, sigma=200))
    xs = (np.arange(255, 513) + .5) * 23.4375
    ys = rng.standard_normal(len(xs))
    xq = (np.arange(512, 1024) + .5) * 23.4375 / 2
    assert np.array_equal(pv.slinear(xs, ys, xq), interpolate.interp1d(xs, ys, kind='slinear')(xq))


@pytest.mark.parametrize("name", EXCERPTS)
@pytest.mark.parametrize("kbps", [128, 96])
def test_decode_matches_reference_decoder(name, kbps):
    gold = np.load(os.path.join(GOLDEN, f"excerpt_vq_{name}.npz"))
    want = np.load(os.path.join(GOLDEN, f"decoded_vq_{name}.npz"))[f"pcm_vq{kbps}"]
    got = pv.decode_stream_vq(bytes(gold[f"pac_vq{kbps}"]))
    assert np.array_equal(got, want)


def test_whole_file_decode_matches_committed_wav(harpsichord_96_pac):
    """encode -> decode of harpsichord at 96 kb/s (SBR + PVQ) equals the decoded
    WAV the reference's author committed (test_decoded_full/harpsichord_96.wav)."""
    rec = json.load(open(os.path.join(GOLDEN, "vqwav.json")))["harpsichord:96"]
    pcm = pv.decode_stream_vq(harpsichord_96_pac)[:rec["n_samples"]]
    assert hashlib.sha256(np.ascontiguousarray(pcm).astype("<i2").tobytes()).hexdigest() == rec["pcm_sha256"]
