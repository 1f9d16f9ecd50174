"""Does running two half batches on two streams (two handles) beat one full batch?
Tells whether stage-level pipelining inside the encode entry point would pay."""
import os, sys, time, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import audio_codec_amd as A
from audio_codec_amd.engine import _ptr

def make(n_frames, seed):
    pcm = A.synth.stream(n_frames, 2, seed=seed)
    enc = A.engine.Encoder(48000, 128 / 48.0)
    planar = torch.as_tensor(A.synth.planar_with_halo(pcm), device=enc.device)
    view = A.engine.PcmView.stream(planar)
    enc.reserve(view.n_cf)
    out = enc.alloc_outputs(view.n_cf, with_payload=True)
    cap = view.n_cf * 512
    body = torch.empty(cap, dtype=torch.uint8, device=enc.device)
    total = torch.zeros(1, dtype=torch.int64, device=enc.device)
    def step():
        enc.encode_pack(view, None, out)
        enc._call("pacx_gather_body", ctypes.c_int64(view.n_cf), _ptr(out["payload"]), _ptr(out["n_bytes"]),
                  _ptr(body), ctypes.c_int64(cap), _ptr(total), enc._stream())
    return step, view.n_cf

def timeit(fn, k=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k

full, n = make(4096, 1)
print("one stream, 8192 cf/step: %.4f ms" % (timeit(full) * 1e3))
for parts in (2, 4):
    steps = [make(4096 // parts, 10 + i)[0] for i in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    def both():
        for st, s in zip(streams, steps):
            with torch.cuda.stream(st):
                s()
    print("%d streams x %d cf: %.4f ms per 8192 cf" % (parts, 8192 // parts, timeit(both) * 1e3))
