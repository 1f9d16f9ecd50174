"""nMDCTLines = 512 ('and 512 cheaply'): the oracle against what the REFERENCE wrote and decoded with
512-line long blocks (tests/golden/lines512.npz, tests/golden/make_golden.py --lines512)."""
import os

import numpy as np
import pytest

from oracle import pac_oracle as po

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lines512.npz"))
CASES = [str(c) for c in G["cases"]]


@pytest.mark.parametrize("tag", CASES)
def test_oracle_writes_the_references_bytes_at_512_lines(tag):
    name, kbps, kind = tag.rsplit("_", 2)
    pac = po.encode_stream(G[f"pcm_{tag}"], int(G[f"sr_{tag}"]), int(kbps), kind == "bs", n_lines=512)
    assert pac == bytes(G[f"pac_{tag}"])


@pytest.mark.parametrize("tag", CASES)
def test_oracle_decodes_to_the_references_pcm_at_512_lines(tag):
    pcm = po.decode_stream(bytes(G[f"pac_{tag}"]))
    assert pcm.shape == G[f"dec_{tag}"].shape and np.array_equal(pcm, G[f"dec_{tag}"])
