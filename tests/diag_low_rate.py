#!/usr/bin/env python3
"""What the product does at sample rates where the reference's band layout has empty bands (below 32 kHz the oracle,
like the reference, raises ValueError in CalcSMRs): python3 tests/diag_low_rate.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import audio_codec_amd as A
import soak_parity as S
from oracle import pac_oracle as po
for sr in (8000, 16000, 22050, 24000, 30000):
    pcm = S.programme(5, 6, 2, sr)
    for name, f in (("oracle", lambda: po.encode_stream(pcm, sr, 96, True)), ("product", lambda: A.pacfile.encode_stream(pcm, sr, 96, block_switching=True))):
        try:
            print(sr, name, "ok", len(f()))
        except Exception as e:
            print(sr, name, "raised", type(e).__name__, str(e)[:100])
