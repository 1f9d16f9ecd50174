"""Host memory to host memory: the encode path as a pipeline of fixed-size chunks.

The C ABI takes device pointers (include/pacx.h); a caller whose PCM sits in host memory -- the
reference's driver reads a WAV file hop by hop (coder/pacfile.py:716-757) -- pays two PCIe crossings
around every batch.  Serial on one stream that is 12.5 M channel-frames/s at 8192 channel-frames per
step; here the copies of chunk i+1 (PCM in) and of chunk i-1 (packed body out) overlap the kernels of
chunk i: three HIP streams, `depth` buffers of everything (pinned on the host side), stream-to-stream
events and NO host synchronisation inside the loop -- the host only waits when it takes a finished
body (round 2 measured this pipeline in tools/pcie_probe.py: 24.8 M cf/s at 131 072 cf per chunk, the
268 MB of PCM then cross at ~50 GB/s, which is the bound).

    hs = HostStreamEncoder(enc, n_channels=2, hops_per_chunk=65536)
    buf = hs.input(k)                # pinned int16 [nCh, hops_per_chunk * 1024]: fill it in place (zero copy) ...
    hs.submit(k, n_hops)             # ... H2D, encode + pack + body, D2H are queued; returns at once
    body = hs.result(k)              # uint8 view of the pinned body of that chunk (waits for its D2H only)

or, for a stream already in memory, `for body in hs.encode(pcm): ...` / `pacfile.encode_stream(...,
chunk_hops=N)`, whose bytes equal the one-batch path's (tests/test_gpu_round3.py).

Chunks are consecutive pieces of ONE stream: the one-hop halo (frame f spans hops f-1, f) and, with
block switching, the transient decisions of the two hops before a chunk are carried from chunk to chunk
on the device.  finish() writes what the reference's driver writes after the last hop: that hop a second
time and the zero block of Close (coder/pacfile.py:743-757, 612-625).
"""
import ctypes

import numpy as np
import torch

from . import _lib
from .engine import PcmView, _ptr

HOP = 1024


class HostStreamEncoder:
    def __init__(self, enc, n_channels, hops_per_chunk, depth=2, block_switching=False):
        self.enc, self.n_ch, self.F, self.depth = enc, int(n_channels), int(hops_per_chunk), int(depth)
        self.block_switching = bool(block_switching)
        dev = enc.device
        F, n_ch = self.F, self.n_ch
        self.n_cf = F * n_ch
        enc.reserve(self.n_cf)
        # body capacity: the bit budget bounds a channel-block (dist.slot_bytes has the argument)
        from .dist import slot_bytes
        self.cap = slot_bytes(self.n_cf, enc.target_bits_per_sample)
        self.s_in, self.s_k, self.s_out = (torch.cuda.Stream(device=dev) for _ in range(3))
        self.host_in = [torch.zeros((n_ch, F * HOP), dtype=torch.int16).pin_memory() for _ in range(depth)]
        self.dev_in = [torch.zeros((n_ch, (F + 1) * HOP), dtype=torch.int16, device=dev) for _ in range(depth)]
        self.outs = [self._alloc_out() for _ in range(depth)]
        self.bodies = [torch.empty(self.cap, dtype=torch.uint8, device=dev) for _ in range(depth)]
        self.totals = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(depth)]
        self.host_body = [torch.empty(self.cap, dtype=torch.uint8).pin_memory() for _ in range(depth)]
        self.host_total = [torch.zeros(1, dtype=torch.int64).pin_memory() for _ in range(depth)]
        self.ev_in = [torch.cuda.Event() for _ in range(depth)]
        self.ev_k = [torch.cuda.Event() for _ in range(depth)]
        self.ev_out = [torch.cuda.Event() for _ in range(depth)]
        self.halo = torch.zeros((n_ch, HOP), dtype=torch.int16, device=dev)        # the hop before the next chunk
        self.carry = torch.zeros(2, dtype=torch.uint8, device=dev)                 # transient decisions of the two hops before it
        self.tr = torch.zeros(F, dtype=torch.uint8, device=dev)
        self.fetch = [self.cap] * depth             # bytes fetched per chunk: the whole slot until a length is known
        self.pending = [False] * depth
        self.n_hops_of = [0] * depth
        self.next_slot = 0
        self.raises = torch.zeros(1, dtype=torch.int32, device=dev)              # PACX_ST_REF_RAISES seen in any chunk

    def _alloc_out(self):
        if self.enc.use_vq:
            e = self.enc
            return {"overall": e._empty((self.n_cf, _lib.SUB), torch.int32),
                    "bit_alloc": torch.zeros((self.n_cf, e.band_stride), dtype=torch.int32, device=e.device),
                    "status": e._empty((self.n_cf,), torch.int32),
                    "payload": e._empty((self.n_cf, e.payload_stride), torch.uint8),
                    "n_bytes": e._empty((self.n_cf,), torch.int32)}
        out = self.enc.alloc_outputs(self.n_cf, with_payload=True)
        out["mantissa"] = None                      # 4 KB per channel-frame nobody reads (short frames pack from the handle's own)
        return out

    # ------------------------------------------------------------------ the three queues
    def input(self, k):
        """pinned staging buffer of slot k: int16 [nCh, hops_per_chunk*1024] as a NumPy view, planar"""
        if self.pending[k]:
            raise RuntimeError(f"slot {k} is still in flight: take result({k}) first")
        return self.host_in[k].numpy()

    def submit(self, k, n_hops=None, _tail=False):
        """queue chunk k (its first n_hops hops): H2D on the copy-in stream, encode + pack + body on the kernel
        stream, D2H on the copy-out stream.  Returns without waiting for any of it."""
        enc, F = self.enc, self.F
        n_hops = F if n_hops is None else int(n_hops)
        if not 0 < n_hops <= F:
            raise ValueError("n_hops must be in 1..hops_per_chunk")
        if self.pending[k]:
            raise RuntimeError(f"slot {k} is still in flight")
        n_cf = n_hops * self.n_ch
        dev_in, out = self.dev_in[k], self.outs[k]
        with torch.cuda.stream(self.s_in):
            self.s_in.wait_event(self.ev_k[k])                       # the kernels that last read dev_in[k] are done
            dev_in[:, HOP:(n_hops + 1) * HOP].copy_(self.host_in[k][:, :n_hops * HOP], non_blocking=True)
            self.ev_in[k].record(self.s_in)
        with torch.cuda.stream(self.s_k):
            self.s_k.wait_event(self.ev_in[k])
            self.s_k.wait_event(self.ev_out[k])                      # the body that last sat in bodies[k] has left
            dev_in[:, :HOP].copy_(self.halo)                         # frame 0 of the chunk starts in the previous chunk
            self.halo.copy_(dev_in[:, n_hops * HOP:(n_hops + 1) * HOP])
            view = PcmView(dev_in, self.n_ch, n_hops, HOP, dev_in.shape[1], 1)
            flags = None
            if self.block_switching:
                tr = self.tr[:n_hops]
                if _tail:
                    tr.zero_()                                       # the pass after EOF and Close: no detection
                else:
                    hops = _lib.PacxPcm(dev_in.data_ptr() + 2 * HOP, _lib.PCM_I16, self.n_ch, n_hops, HOP,
                                        dev_in.shape[1], 1)
                    enc._call("pacx_transient_flags", ctypes.byref(hops), _ptr(tr), None, enc._stream())
                ext = torch.cat((self.carry, tr))                    # decisions of hops g-2, g-1, g, ...
                flags = ext[:n_hops] | (ext[1:n_hops + 1] << 1) | (ext[2:n_hops + 2] << 2)
                if _tail:
                    flags[-1] = 0                                    # Close writes (0, 0, 0)
                self.carry.copy_(ext[n_hops:n_hops + 2])
                flags = flags.contiguous()
            sub = {name: (t[:n_cf] if t is not None else None) for name, t in out.items()}
            if enc.use_vq:
                enc.encode_vq(view, flags, sub)
            else:
                enc.encode_pack(view, flags, sub)
                if enc.use_sbr:                                      # scalar mantissas in an SBR file: where the reference raises
                    self.raises |= (sub["status"] & _lib.ST_REF_RAISES).max()
            enc._call("pacx_gather_body", ctypes.c_int64(n_cf), _ptr(sub["payload"]), _ptr(sub["n_bytes"]),
                      _ptr(self.bodies[k]), ctypes.c_int64(self.cap), _ptr(self.totals[k]), enc._stream())
            self.ev_k[k].record(self.s_k)
        with torch.cuda.stream(self.s_out):
            self.s_out.wait_event(self.ev_k[k])
            n = min(self.fetch[k], self.cap)
            self.host_body[k][:n].copy_(self.bodies[k][:n], non_blocking=True)
            self.host_total[k].copy_(self.totals[k], non_blocking=True)
            self.ev_out[k].record(self.s_out)
        self.pending[k] = True
        self.n_hops_of[k] = n_hops

    def result(self, k):
        """the packed body of chunk k ('<L nBytes' + payload per channel-block, in stream order): a uint8 NumPy
        view of pinned memory, valid until slot k is submitted again.  Waits for that chunk's D2H copy only."""
        if not self.pending[k]:
            raise RuntimeError(f"nothing was submitted in slot {k}")
        self.ev_out[k].synchronize()
        n = int(self.host_total[k].item())
        if n > self.cap:
            raise RuntimeError(f"body of {n} bytes does not fit its {self.cap}-byte buffer")
        if n > self.fetch[k]:                      # the fetch was sized from an earlier chunk: get the rest
            with torch.cuda.stream(self.s_out):
                self.host_body[k][self.fetch[k]:n].copy_(self.bodies[k][self.fetch[k]:n], non_blocking=True)
            self.s_out.synchronize()
        # later chunks fetch what this one needed plus a margin instead of the whole slot
        per_hop = -(-n // max(self.n_hops_of[k], 1))
        self.fetch = [min(self.cap, per_hop * self.F + (1 << 16))] * self.depth
        self.pending[k] = False
        return self.host_body[k][:n].numpy()

    # ------------------------------------------------------------------ a whole stream
    def encode(self, pcm, finish=True):
        """pcm: int16 [n_hops*1024, nCh] in host memory (any NumPy array).  Yields the chunks' bodies in order;
        with finish, the last two blocks of the file (the last hop again, Close) close the stream."""
        pcm = np.asarray(pcm)
        n_hops = len(pcm) // HOP
        order = []
        for h0 in range(0, n_hops, self.F):
            k = self.next_slot
            self.next_slot = (k + 1) % self.depth
            if self.pending[k]:
                yield self.result(order.pop(0))
            n = min(self.F, n_hops - h0)
            self.input(k)[:, :n * HOP] = pcm[h0 * HOP:(h0 + n) * HOP].T
            self.submit(k, n)
            order.append(k)
        if finish and n_hops:
            k = self.next_slot
            self.next_slot = (k + 1) % self.depth
            if self.pending[k]:
                yield self.result(order.pop(0))
            tail = self.input(k)
            tail[:, :HOP] = pcm[(n_hops - 1) * HOP:n_hops * HOP].T
            tail[:, HOP:2 * HOP] = 0
            self.submit(k, 2, _tail=True)
            order.append(k)
        for k in order:
            yield self.result(k)

    def reference_raises(self):
        """True if a block of the stream so far is one the reference cannot write (PACX_ST_REF_RAISES: scalar
        mantissas in an SBR file, an omitted band got bits -- coder/quantize.py:73-74 raises TypeError there)"""
        return bool(int(self.raises.item()))

    def reset(self):
        """start a new stream (halo and transient carry back to the start of a file)"""
        torch.cuda.synchronize(self.enc.device)
        self.halo.zero_()
        self.carry.zero_()
        self.raises.zero_()
        self.pending = [False] * self.depth
