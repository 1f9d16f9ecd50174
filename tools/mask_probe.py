#!/usr/bin/env python3
"""Runs pacx_smr_batch (k_side_long + k_mask) on different inputs, for rocprofv3."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import audio_codec_amd as A
enc = A.engine.Encoder(48000, 128 / 48.0)
n = 4096
cases = {"synth": A.synth.stream(n, 2), "zeros": np.zeros((n * 1024, 2), np.int16),
         "tones_only": None}
x = A.synth.stream(n, 2).astype(np.float64)
rng = np.random.default_rng(0)
# tones without the noise floor: quantisation noise only
amps, freqs = A.synth.AMPS, A.synth.FREQS
t = np.arange(n * 1024)
tone = 0.5 * sum(a * np.cos(2 * np.pi * f * t / 48000) for a, f in zip(amps, freqs))
cases["tones_only"] = np.stack([np.rint(32767 * tone).astype(np.int16)] * 2, axis=1)
for name, pcm in cases.items():
    planar = torch.as_tensor(A.synth.planar_with_halo(pcm), device=enc.device)
    view = A.engine.PcmView.stream(planar)
    lines = enc.mdct(view)
    for _ in range(3):
        smr, npk = enc.smr(view, lines, want_peaks=True)
    torch.cuda.synchronize()
    print(name, "mean peaks", float(npk.float().mean()), flush=True)
