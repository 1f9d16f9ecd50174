"""PCIe-inclusive rate of the headline workload: host int16 PCM (pinned) -> HBM,
encode + pack + body, packed body -> host: serial on one stream, then with the copies
overlapped (three streams, double buffers)."""
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

# ---- overlapped: three streams (H2D, kernels, D2H), two buffers of everything -----------------
# Step i+1's PCM crosses PCIe while step i is encoded and step i-1's body goes back; no host
# synchronisation inside the loop (the body is fetched as its fixed-capacity slot, the
# valid length with it).
s_in, s_k, s_out = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
devs = [torch.empty_like(host, device=enc.device) for _ in range(2)]
views = [A.engine.PcmView.stream(d) for d in devs]
outs = [enc.alloc_outputs(n_cf, with_payload=True) for _ in range(2)]
bodies = [torch.empty(cap, dtype=torch.uint8, device=enc.device) for _ in range(2)]
totals = [torch.zeros(1, dtype=torch.int64, device=enc.device) for _ in range(2)]
host_bodies = [torch.empty(cap, dtype=torch.uint8).pin_memory() for _ in range(2)]
host_totals = [torch.zeros(1, dtype=torch.int64).pin_memory() for _ in range(2)]
ev_in = [torch.cuda.Event() for _ in range(2)]
ev_k = [torch.cuda.Event() for _ in range(2)]
ev_out = [torch.cuda.Event() for _ in range(2)]
slot = n + (1 << 16)                      # what is fetched per step: the valid bytes + slack


def pipelined(steps):
    for i in range(steps):
        k = i & 1
        with torch.cuda.stream(s_in):
            s_in.wait_event(ev_k[k])                       # kernels of step i-2 are done with devs[k]
            devs[k].copy_(host, non_blocking=True)
            ev_in[k].record(s_in)
        with torch.cuda.stream(s_k):
            s_k.wait_event(ev_in[k])
            s_k.wait_event(ev_out[k])                      # body of step i-2 has left bodies[k]
            enc.encode_pack(views[k], None, outs[k])
            enc._call("pacx_gather_body", ctypes.c_int64(n_cf), _ptr(outs[k]["payload"]),
                      _ptr(outs[k]["n_bytes"]), _ptr(bodies[k]), ctypes.c_int64(cap), _ptr(totals[k]),
                      enc._stream())
            ev_k[k].record(s_k)
        with torch.cuda.stream(s_out):
            s_out.wait_event(ev_k[k])
            host_bodies[k][:slot].copy_(bodies[k][:slot], non_blocking=True)
            host_totals[k].copy_(totals[k], non_blocking=True)
            ev_out[k].record(s_out)
    torch.cuda.synchronize()


pipelined(4)
assert int(host_totals[0].item()) == n and int(host_totals[1].item()) == n
assert torch.equal(host_bodies[1][:n], host_body[:n])
t0 = time.perf_counter()
K = 40
pipelined(K)
dt = (time.perf_counter() - t0) / K
print(f"overlapped (3 streams, double buffers): {dt * 1e3:.3f} ms/step = {n_cf / dt / 1e6:.2f} M cf/s PCIe-inclusive")
