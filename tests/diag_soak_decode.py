#!/usr/bin/env python3
"""Diagnose a decoder mismatch of tests/soak_parity.py (gain-shape streams): python3 tests/diag_soak_decode.py <seed> ...
decodes the oracle's stream with the product and with the oracle and reports where the PCM differs."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import audio_codec_amd as A
import soak_parity as S
from oracle import pac_oracle_vq as pv

for seed in map(int, sys.argv[1:]):
    c = S.draw_case(seed, ["vq"])
    pcm = S.programme(c["seed"], c["n_hops"], c["n_ch"], c["sr"])
    want = pv.encode_stream_vq(pcm, c["sr"], c["kbps"])
    got = A.pacfile.encode_stream(pcm, c["sr"], c["kbps"], block_switching=True, use_vq=True, use_sbr=c["kbps"] < 128)
    from oracle import pac_oracle as po
    floats, orig = [], po.fraction_to_pcm16
    def spy(x):
        floats.append(np.array(x, dtype=np.float64))
        return orig(x)
    po.fraction_to_pcm16 = spy
    try:
        d_o = pv.decode_stream_vq(want)
    finally:
        po.fraction_to_pcm16 = orig
    n_ch = c["n_ch"]
    fl = np.concatenate([np.stack(floats[i:i + n_ch], axis=1) for i in range(0, len(floats), n_ch)])
    d_p = A.pacfile.decode_stream(want)
    print(f"== {c}: streams equal {got == want}; decoded shapes {d_p.shape} {d_o.shape}")
    if d_p.shape != d_o.shape:
        continue
    diff = d_p.astype(np.int64) - d_o.astype(np.int64)
    idx = np.argwhere(diff != 0)
    print(f"  {len(idx)} samples differ, max |difference| {int(np.abs(diff).max())}")
    for (n, ch) in idx[:12]:
        t = 65535.0 * abs(fl[n, ch]) + 1.0                 # the quantiser takes floor(t / 2): coder/quantize.py:73
        print(f"    sample {n} (hop {n // 1024}, offset {n % 1024}) channel {ch}: product {int(d_p[n, ch])} oracle {int(d_o[n, ch])}; "
              f"the oracle's sample {fl[n, ch]!r}, (2^16 - 1)|x| + 1 = {t!r}: {abs(t - 2 * round(t / 2)):.3e} from an even integer")
    hops = sorted({int(n) // 1024 for n, _ in idx})
    print("  hops touched:", hops[:20])
