#!/usr/bin/env python3
"""Times pacx_mdct_batch alone over a range of batch sizes (HIP events)."""
import ctypes, json, sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import audio_codec_amd as A
from audio_codec_amd.engine import _ptr

enc = A.engine.Encoder(48000, 128 / 48.0)
dev = enc.device
sizes = [int(s) for s in sys.argv[1:]] or [1536, 3072, 6144, 8192, 16384, 65536, 262144]
base = A.synth.stream(4096, 2)
for n_frames2 in sizes:
    n_frames = n_frames2 // 2
    reps = -(-n_frames // 4096)
    pcm = np.tile(base, (reps, 1))[:n_frames * 1024]
    planar = torch.as_tensor(A.synth.planar_with_halo(pcm), device=dev)
    view = A.engine.PcmView.stream(planar)
    n_cf = view.n_cf
    lines = torch.empty((n_cf, 1024), dtype=torch.float64, device=dev)
    scale = torch.empty((n_cf,), dtype=torch.int32, device=dev)
    def once():
        enc._call("pacx_mdct_batch", ctypes.byref(view.c), None, 0, _ptr(lines), _ptr(scale), enc._stream())
    for _ in range(5):
        once()
    torch.cuda.synchronize()
    n = 30
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); once(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    ms = float(np.median(ts))
    print(json.dumps({"n_cf": n_cf, "median_us": ms * 1e3, "min_us": ts[0] * 1e3,
                      "GBps": n_cf * 10240 / (ms * 1e-3) / 1e9, "frac_8TBs": n_cf * 10240 / (ms * 1e-3) / 8e12}), flush=True)
    del lines, planar
