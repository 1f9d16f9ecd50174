#!/usr/bin/env python3
"""Do two independent encode chains on two streams fill each other's idle issue slots?  Two handles, two batches
(n frames each), K steps each: one after the other on one stream, then side by side on two streams."""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import audio_codec_amd as A
from audio_codec_amd.engine import _ptr

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = 40
encs = [A.engine.Encoder(48000, 128 / 48.0) for _ in range(2)]
dev = encs[0].device
items = []
for i, enc in enumerate(encs):
    pcm = A.synth.stream(n_frames, 2, seed=422 + i)
    planar = torch.as_tensor(A.synth.planar_with_halo(pcm), device=dev)
    view = A.engine.PcmView.stream(planar)
    enc.reserve(view.n_cf)
    out = enc.alloc_outputs(view.n_cf, with_payload=True)
    body = torch.empty(view.n_cf * 512, dtype=torch.uint8, device=dev)
    total = torch.zeros(1, dtype=torch.int64, device=dev)
    items.append((enc, planar, view, out, body, total))


def step(i):
    enc, planar, view, out, body, total = items[i]
    enc.encode_pack(view, None, out)
    enc._call("pacx_gather_body", ctypes.c_int64(view.n_cf), _ptr(out["payload"]), _ptr(out["n_bytes"]), _ptr(body),
              ctypes.c_int64(body.numel()), _ptr(total), enc._stream())


for _ in range(5):
    step(0); step(1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    step(0); step(1)
torch.cuda.synchronize()
serial = (time.perf_counter() - t0) / K
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
for _ in range(5):
    for i in range(2):
        with torch.cuda.stream(streams[i]):
            step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    for i in range(2):
        with torch.cuda.stream(streams[i]):
            step(i)
torch.cuda.synchronize()
both = (time.perf_counter() - t0) / K
cf = 2 * items[0][2].n_cf
print(f"{n_frames} frames per batch: two batches one after the other {serial * 1e3:.3f} ms = {cf / serial / 1e6:.2f} M cf/s; "
      f"side by side on two streams {both * 1e3:.3f} ms = {cf / both / 1e6:.2f} M cf/s")
