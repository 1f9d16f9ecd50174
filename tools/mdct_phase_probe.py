#!/usr/bin/env python3
"""Phase breakdown of the pipelined MDCT kernel from in-kernel s_memtime stamps.
Needs a library built with -DPACX_MDCT_DEBUG (PACX_LIB=...libpacx_dbg.so) and
the default MDCT dispatch."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import audio_codec_amd as A
from audio_codec_amd.engine import _ptr

n_cf = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
enc = A.engine.Encoder(48000, 128 / 48.0)
base = A.synth.stream(4096, 2)
n_frames = n_cf // 2
pcm = np.tile(base, (-(-n_frames // 4096), 1))[:n_frames * 1024]
planar = torch.as_tensor(A.synth.planar_with_halo(pcm), device=enc.device)
view = A.engine.PcmView.stream(planar)
lines = torch.empty((view.n_cf, 1024), dtype=torch.float64, device=enc.device)
scale = torch.empty((view.n_cf,), dtype=torch.int32, device=enc.device)
for _ in range(3):
    enc._call("pacx_mdct_batch", ctypes.byref(view.c), None, 0, _ptr(lines), _ptr(scale), enc._stream())
torch.cuda.synchronize()
lib = A._lib.load()
out = (ctypes.c_longlong * 128)()
lib.pacx_debug_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
print("rc", lib.pacx_debug_read(out, 128))
a = np.array(out[:]).reshape(8, 16)[:, :8]
iters = view.n_cf / (256 * 8 * 2)
names = ["wait_dma", "fold", "init+dma_issue", "fft", "epilogue_A", "dft8B+epilogue_B", "-", "-"]
print("iterations per wave", iters)
for k in range(6):
    print("%-18s %s  mean %.0f" % (names[k], (a[:, k] / iters).astype(int).tolist(), a[:, k].mean() / iters))
print("total per iteration", a[:, :6].sum(axis=1).mean() / iters)
