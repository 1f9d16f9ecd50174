"""PCIe-inclusive rate of the headline workload: host int16 PCM (pinned) -> HBM,
encode + pack + body, packed body -> host.  Not overlapped (one stream)."""
import os, sys, time, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import audio_codec_amd as A
from audio_codec_amd.engine import _ptr

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
pcm = A.synth.stream(n_frames, 2)
host = torch.from_numpy(A.synth.planar_with_halo(pcm)).pin_memory()
enc = A.engine.Encoder(48000, 128 / 48.0)
dev = torch.empty_like(host, device=enc.device)
view = A.engine.PcmView.stream(dev)
n_cf = view.n_cf
enc.reserve(n_cf)
out = enc.alloc_outputs(n_cf, with_payload=True)
cap = n_cf * 512
body = torch.empty(cap, dtype=torch.uint8, device=enc.device)
total = torch.zeros(1, dtype=torch.int64, device=enc.device)
host_body = torch.empty(cap, dtype=torch.uint8).pin_memory()

def step():
    dev.copy_(host, non_blocking=True)
    enc.encode_pack(view, None, out)
    enc._call("pacx_gather_body", ctypes.c_int64(n_cf), _ptr(out["payload"]), _ptr(out["n_bytes"]),
              _ptr(body), ctypes.c_int64(cap), _ptr(total), enc._stream())
    n = int(total.item())
    host_body[:n].copy_(body[:n], non_blocking=True)
    torch.cuda.synchronize()
    return n

for _ in range(3):
    n = step()
t0 = time.perf_counter()
K = 20
for _ in range(K):
    n = step()
dt = (time.perf_counter() - t0) / K
print(f"{n_cf} cf/step, H2D {host.numel() * 2 / 1e6:.1f} MB, D2H {n / 1e6:.1f} MB: {dt * 1e3:.3f} ms/step = "
      f"{n_cf / dt / 1e6:.2f} M cf/s PCIe-inclusive (serial copies, one stream)")
