#!/usr/bin/env python3
"""Stage split of k_vq_dec_frame from in-kernel s_memtime stamps (debug build, PACX_LIB=...libpacx_dbg.so)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import audio_codec_amd as A

kbps = int(sys.argv[1]) if len(sys.argv) > 1 else 128
enc = A.context.encoder(48000, kbps / 48.0, use_vq=True, use_sbr=kbps < 128)
pcm = A.synth.stream(4096, 2)
planar = torch.as_tensor(A.synth.planar_with_halo(pcm), device=enc.device)
out = enc.encode_vq(A.engine.PcmView.stream(planar))
for _ in range(3):
    enc.decode_vq(out["payload"], out["n_bytes"], 2)
torch.cuda.synchronize()
lib = A._lib.load()
buf = (ctypes.c_longlong * 8)()
lib.pacx_debug_read_vqd.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.pacx_debug_read_vqd(buf, 8)
t = np.array(buf[:5], dtype=np.float64)
names = ["header (thread 0)", "parse: one band per lane (wave 0)", "leaves: one per lane", "combine, level by level", "gains + lines"]
print("k_vq_dec_frame at %d kb/s, thread 0 of every workgroup, share of its time:" % kbps)
for n, v in zip(names, t):
    print("  %-36s %.1f %%" % (n, 100.0 * v / t.sum()))
