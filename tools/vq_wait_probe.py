#!/usr/bin/env python3
"""Where the waves of k_vq_frame wait: busy / waiting cycles per wave ROLE at the two barriers of a level
(library built with -DPACX_VQ_WAITDBG on k_vq.hip: `python audio-codec_amd/build.py --variant wait k_vq.hip
-DPACX_VQ_WAITDBG`, PACX_LIB=audio-codec_amd/variants/libpacx_wait.so).  Stamps sit next to barriers only."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import audio_codec_amd as A

kbps = int(sys.argv[1]) if len(sys.argv) > 1 else 128
enc = A.context.encoder(48000, kbps / 48.0, use_vq=True, use_sbr=kbps < 128)
pcm = A.synth.stream(4096, 2)
planar = torch.as_tensor(A.synth.planar_with_halo(pcm), device=enc.device)
view = A.engine.PcmView.stream(planar)
lib = A._lib.load()
lib.pacx_debug_read_vqw.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
out = (ctypes.c_longlong * 32)()
enc.encode_vq(view)
torch.cuda.synchronize()
lib.pacx_debug_read_vqw(out, 32, 1)
for _ in range(3):
    enc.encode_vq(view)
torch.cuda.synchronize()
lib.pacx_debug_read_vqw(out, 32, 0)
t = np.array(out[:], dtype=np.float64)
units = t[18]
print(f"k_vq_frame at {kbps} kb/s: {units:.0f} units, {t[17] / units:.2f} levels per unit; cycles per unit: whole "
      f"{t[20] / units:.0f}, phase A {t[19] / units:.0f}, level loop {t[16] / units:.0f}")
print("role   split passes busy / wait at barrier 1   |  scalar or leaves busy / wait at barrier 2   (cycles per unit)")
for r in range(4):
    b1, w1, b2, w2 = t[4 * r:4 * r + 4] / units
    print(f"  {r}    {b1:9.0f} / {w1:9.0f}                   | {b2:9.0f} / {w2:9.0f}")
if os.environ.get("PACX_VQ_FRAME") == "1":
    names = ["tables + lines into LDS + barrier", "band gains, x / gain + barrier", "header, roots, first classify + barriers",
             "fields ORed in", "gains quantised + barrier"]
    for k, n in enumerate(names):
        print(f"  thread 0: {n:45s} {t[21 + k] / units:9.0f}")
else:
    names = ["phase A (lines, gains, header, roots, first sort)", "leaf lists + barrier", "small leaves, one per lane (wave 0)",
             "larger leaves + barrier", "enumeration terms + barrier", "widths, positions", "fields, gains, hand-over"]
    for k, n in enumerate(names):
        print(f"  wave 0: {n:50s} {t[21 + k] / units:9.0f}")
