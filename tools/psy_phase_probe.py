#!/usr/bin/env python3
"""Phase breakdown of k_mask (long blocks) from in-kernel s_memtime stamps.  Needs the
library built by `python audio-codec_amd/build.py --phase-debug` (PACX_LIB=...libpacx_dbg.so)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import audio_codec_amd as A

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
enc = A.context.encoder(48000, 128 / 48.0)
pcm = A.synth.stream(n_frames, 2)
planar = torch.as_tensor(A.synth.planar_with_halo(pcm), device=enc.device)
view = A.engine.PcmView.stream(planar)
for _ in range(3):
    enc.encode_pack(view)
torch.cuda.synchronize()
lib = A._lib.load()
out = (ctypes.c_longlong * 256)()
lib.pacx_debug_read_psy.argtypes = [ctypes.c_void_p, ctypes.c_int]
print("rc", lib.pacx_debug_read_psy(out, 256))
a = np.array(out[:]).reshape(16, 16)[:4, :8]
names = ["init+peak load", "masker loop", "line round trips", "band maxima"]
tot = a[:, :4].sum()
print("k_mask<1024>, workgroup 5, ticks per wave:")
for k in range(4):
    print("  %-18s %s  share %.1f %%" % (names[k], a[:, k].tolist(), 100.0 * a[:, k].sum() / tot))
b = np.array(out[128:136], dtype=np.float64)
names = ["stage PCM", "Hann window", "2 x FFT-512", "split + intensities", "peak picking", "make_peak (log10, atan)",
         "pruning scans + output"]
print("k_side_long, all blocks, share of a block's time:")
for k in range(7):
    print("  %-26s %.1f %%" % (names[k], 100.0 * b[k] / b[:7].sum()))
st = np.array(out[160:164], dtype=np.float64)
n_units = 3 * 2 * n_frames          # three encode calls
print("k_mask<1024> screen, per channel-frame: %.1f masker batches, %.1f (batch, chunk) pairs past the batch screen, "
      "%.1f of them with survivors, %.1f survivor evaluations" % (st[3] / n_units, st[0] / n_units, st[2] / n_units, st[1] / n_units))
t = (ctypes.c_longlong * 16)()
lib.pacx_debug_read_tail.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.pacx_debug_read_tail(t, 16)
t = np.array(t[:3], dtype=np.float64)
print("k_tail_long, all blocks, share of a block's time:")
for name, v in zip(["BitAlloc (half wave)", "scale factors + mantissas", "payload"], t):
    print("  %-26s %.1f %%" % (name, 100.0 * v / t.sum()))
