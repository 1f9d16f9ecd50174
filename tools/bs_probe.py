#!/usr/bin/env python3
"""Block-switched (BASELINE configs[2]-style) throughput probe: castanet excerpt tiled."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import audio_codec_amd as A
ex = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "excerpt_castanet.npz"))
tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 64
pcm = np.tile(ex["pcm"], (tiles, 1))
sr = int(ex["sr"])
enc = A.context.encoder(sr, 128 / (sr / 1000))
planar = A.pacfile.device_stream(enc, pcm)
n_hops = len(pcm) // 1024
view = A.engine.PcmView.stream(planar)
enc.reserve(view.n_cf)
out = enc.alloc_outputs(view.n_cf, with_payload=True)
def step():
    tr, fl = enc.transient_flags(planar, n_hops)
    enc.encode_pack(view, fl, out)
    return fl
for _ in range(3):
    fl = step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 10
for _ in range(n):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
short = int(((fl.cpu().numpy() >> 1) & 1).sum())
print(f"hops {n_hops} cf {view.n_cf} short-coded frames {short} ({100*short/(n_hops+2):.0f}%)  {dt*1e3:.3f} ms/step  {view.n_cf/dt/1e6:.2f} M cf/s")
