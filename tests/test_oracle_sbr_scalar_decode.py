"""Decode_SBR's scalar branch (coder/codec.py:117-134, useVQ off) and the reader rule that feeds it
(coder/pacfile.py:203-205, 659-663): the oracle against what the REFERENCE made of the same inputs
(tests/golden/sbr_scalar_decode.npz, tests/golden/make_golden.py --sbr-scalar-decode)."""
import os

import numpy as np
import pytest

from oracle import pac_oracle as po

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "sbr_scalar_decode.npz"))
CASES = [str(c) for c in G["stream_cases"]]


@pytest.mark.parametrize("tag", CASES)
def test_streams_decode_to_the_reference_decoders_pcm(tag):
    pac = bytes(G[f"pac_{tag}"])
    p, _, _ = po.parse_header(pac)
    assert p.useSBR and len(p.omittedBands) > 0
    pcm = po.decode_stream(pac)
    assert pcm.shape == G[f"pcm_{tag}"].shape
    assert np.array_equal(pcm, G[f"pcm_{tag}"])


def test_streams_do_code_omitted_bands():
    """the material exercises the branch: every stream has long blocks with bits in an omitted band"""
    for tag in CASES:
        pac = bytes(G[f"pac_{tag}"])
        p, _, pos = po.parse_header(pac)
        hit = 0
        while pos < len(pac):
            n = int.from_bytes(pac[pos:pos + 4], "little")
            br = po.BitReader(pac[pos + 4:pos + 4 + n])
            pos += 4 + n
            fl = (br.get(1), br.get(1), br.get(1))
            if not fl[1]:
                _, alloc, _, _ = po.parse_block_body(br, p, False)
                hit += int(any(alloc[b] for b in p.omittedBands))
        assert hit >= 4, tag


@pytest.mark.parametrize("sr", [48000, 44100, 32000])
def test_decode_sbr_scalar_blocks(sr):
    p = po.make_params(sr, 1, 96)
    p.useSBR = True
    p.omittedBands = list(po.omitted_bands(p.sfBands))
    want = G[f"fn_{sr}_block"]
    for k in range(len(want)):
        fl = G[f"fn_{sr}_flags"][k]
        got = po.decode_block_sbr_scalar(p, G[f"fn_{sr}_sf"][k], G[f"fn_{sr}_ba"][k], G[f"fn_{sr}_mant"][k],
                                         int(G[f"fn_{sr}_overall"][k]), bool(fl[0]), False, bool(fl[1]))
        assert np.array_equal(got, want[k]), (sr, k, float(np.abs(got - want[k]).max()))
