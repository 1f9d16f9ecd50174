"""CPU checks of the parity soak's own helpers (tests/soak_parity.py): the soak runs by hand on the GPU box, so what
its verdicts rest on -- the case generator, the block parser, the 'degenerate' test -- is pinned here."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import soak_parity as S
from oracle import pac_oracle as po


def test_cases_are_reproducible_and_cover_the_coders():
    cases = [S.draw_case(s) for s in range(1000, 1300)]
    assert cases == [S.draw_case(s) for s in range(1000, 1300)]
    assert {c["coder"] for c in cases} == {"scalar", "scalar_bs", "vq"}
    assert {c["sr"] for c in cases} == {32000, 44100, 48000, 96000}
    assert {c["n_ch"] for c in cases} == {1, 2, 3}
    only = [S.draw_case(s, ["vq"], [88200, 192000], [4, 8]) for s in range(50)]
    assert {c["coder"] for c in only} == {"vq"} and {c["sr"] for c in only} == {88200, 192000}
    assert {c["n_ch"] for c in only} == {4, 8}
    assert {S.draw_case(s, ["scalar_sbr"])["coder"] for s in range(20)} == {"scalar_sbr"}
    a = S.programme(1234, 8, 2, 48000)
    assert a.dtype == np.int16 and a.shape == (8 * 1024, 2)
    assert np.array_equal(a, S.programme(1234, 8, 2, 48000))


def test_degenerate_blocks():
    rng = np.random.default_rng(3)
    n = 2048
    noise = np.rint(rng.standard_normal(n) * 300).astype(np.int16)
    assert not S.degenerate(po, noise)
    assert S.degenerate(po, np.zeros(n, np.int16))                                  # silence
    imp = np.zeros(n, np.int16)
    imp[[6, 1030]] = 1
    assert S.degenerate(po, imp)                                                    # impulses: a flat spectrum
    assert S.degenerate(po, np.full(n, 12345, np.int16))                            # constant (a clipped stretch)
    assert S.degenerate(po, np.tile(np.array([0, 2, 0, -2], np.int16), n // 4))     # fs/4 at 2 LSB: one bin
    assert not S.degenerate(po, (noise[:256]))                                      # a short sub-block of noise
    tone = np.rint(8000 * np.sin(2 * np.pi * 997 / 48000 * np.arange(n))).astype(np.int16)
    assert not S.degenerate(po, tone)                                               # a rounded sine has a noise floor


def test_block_parser_reads_what_the_oracle_wrote():
    pcm = S.programme(77, 6, 2, 48000)
    pac = po.encode_stream(pcm, 48000, 128, True)
    p = po.make_params(48000, 2, 128)
    hdr = len(po.pac_header(p, len(pcm)))
    blocks = S._blocks(pac, hdr)
    assert len(blocks) % 2 == 0 and 2 * 6 <= len(blocks) <= 2 * 8
    n_short = 0
    for b in blocks:
        fl, units = S._parse_scalar_block(po, p, b)
        assert len(units) == (8 if fl[1] else 1)
        n_short += fl[1]
        for ov, ba, sf, mant in units:
            bands = p.sfBandsShort if fl[1] else p.sfBands
            assert 0 <= ov < 16 and len(ba) == len(sf) == bands.nBands
            assert [len(m) for m in mant] == [int(bands.nLines[k]) if ba[k] else 0 for k in range(bands.nBands)]
    case, want, dec, _ = S.oracle_side(dict(seed=77, coder="scalar_bs", sr=48000, n_ch=2, kbps=128, n_hops=6))
    assert want == pac and len(dec) == 64                                           # the worker's side of a case
