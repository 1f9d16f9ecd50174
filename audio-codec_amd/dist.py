"""Multi-GPU sharding of a stream: frames shard
embarrassingly -- rank r encodes a contiguous range of hops plus a one-hop halo
on its left -- and the only collective is the final gather of the packed
bitstream to rank 0 (RCCL over xGMI through torch.distributed 'nccl'; 'gloo' on
CPU for the tests).  No data-path collective before that."""
import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(n_hops, world_size, rank):
    """Contiguous hop range [lo, hi) of rank `rank`; the first ranks take the
    remainder.  Frames lo..hi-1 need hops lo-1..hi-1 (hop -1 = zeros)."""
    base, rem = divmod(n_hops, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_with_halo(pcm, world_size, rank, hop=1024):
    """pcm: int16 [n_hops*hop, nCh] (whole stream, host).  Returns this rank's
    planar [nCh, (hi-lo+1)*hop] slice with its left halo hop."""
    n_hops = len(pcm) // hop
    lo, hi = shard_bounds(n_hops, world_size, rank)
    out = np.zeros((pcm.shape[1], (hi - lo + 1) * hop), dtype=pcm.dtype)
    src_lo = max(lo - 1, 0) * hop
    seg = pcm[src_lo:hi * hop].T
    out[:, out.shape[1] - seg.shape[1]:] = seg
    return out


def shard_flags(flags, world_size, rank):
    """Block-switching flags of this rank's frames.  `flags` holds one
    (last, cur, next) byte per hop of the WHOLE stream: a frame's flags look two
    hops back (coder/pacfile.py:732-741), so they are computed once for the
    stream (pacx_transient_flags, per-hop independent) and sliced, not
    recomputed per shard."""
    lo, hi = shard_bounds(len(flags), world_size, rank)
    return flags[lo:hi]


def gather_bitstream(body, n_bytes_total, group=None, dst=0):
    """body: uint8 device (or CPU, for gloo) tensor holding this rank's packed
    '<L nBytes'+payload records in its first n_bytes_total bytes.  Rank `dst`
    gets the concatenation in rank order (a uint8 tensor), others get None.
    Two collectives: all_gather of the sizes (8 bytes/rank), then a gather of
    the payloads padded to the largest."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    size = torch.as_tensor([int(n_bytes_total)], dtype=torch.int64, device=body.device)
    sizes = [torch.zeros_like(size) for _ in range(world)]
    dist.all_gather(sizes, size, group=group)
    sizes = [int(s.item()) for s in sizes]
    cap = max(max(sizes), 1)
    send = body[:cap] if body.numel() >= cap else torch.cat(
        (body, torch.zeros(cap - body.numel(), dtype=body.dtype, device=body.device)))
    send = send.contiguous()
    if rank == dst:
        bufs = [torch.empty(cap, dtype=torch.uint8, device=body.device) for _ in range(world)]
        dist.gather(send, bufs, dst=dst, group=group)
        return torch.cat([b[:n] for b, n in zip(bufs, sizes)])
    dist.gather(send, None, dst=dst, group=group)
    return None


# ---- asynchronous, fixed-slot flavour (what bench.py uses for N > 1) -----------------
HEADER = 8          # bytes: little-endian int64 = valid bytes that follow


def slot_bytes(n_cf, target_bits_per_sample, hop=1024):
    """Upper bound on one rank's body ('<L nBytes' + payload per cf) so that the
    gather can use fixed-size slots with no size exchange: a channel-block never
    exceeds its bit budget (long: target*hop bits; short frames are budgeted from
    int(1.45*128) lines per sub-block, i.e. 1.45x) plus flags, per-band headers,
    rounding and the 4-byte length."""
    per_cf = int(np.ceil(target_bits_per_sample * hop * 1.45 / 8)) + 128
    return HEADER + int(n_cf) * per_cf


class BitstreamGather:
    """Fixed-slot gather of every rank's packed body to rank `dst`, enqueued
    asynchronously: no host synchronisation, no size exchange.  Each rank sends one
    slot = [int64 valid bytes][body ... padding]; rank dst receives world slots and
    can slice them later with unpack().  Two send buffers alternate so that the
    gather of step i overlaps the encode of step i+1 (the caller must not reuse a
    buffer before wait(k))."""

    def __init__(self, slot, device, group=None, dst=0, depth=2):
        self.group, self.dst, self.slot = group, dst, int(slot)
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.send = [torch.zeros(self.slot, dtype=torch.uint8, device=device) for _ in range(depth)]
        self.recv = [[torch.empty(self.slot, dtype=torch.uint8, device=device) for _ in range(self.world)]
                     if self.rank == dst else None for _ in range(depth)]
        self.work = [None] * depth
        self.sent = [None] * depth          # valid-byte counts of the bodies handed to launch()

    def capacity(self):
        """bytes a body may hold"""
        return self.slot - HEADER

    def body(self, k):
        """Where the encoder writes its records of step k (after the header)."""
        return self.send[k][HEADER:]

    def launch(self, k, total):
        """total: int64 device tensor [1] = valid bytes in body(k) (device-side, no .item())."""
        if not total.is_cuda and int(total.item()) > self.capacity():
            # a host-side count can be checked before anything is sent
            raise RuntimeError(f"rank {self.rank}: body of {int(total.item())} bytes overflows its "
                               f"{self.capacity()}-byte gather slot")
        self.send[k][:HEADER].copy_(total.view(torch.uint8))
        self.sent[k] = total.clone()        # device-side copy: read by check() outside the timed loop
        self.work[k] = dist.gather(self.send[k], self.recv[k], dst=self.dst, group=self.group, async_op=True)

    def wait(self, k):
        if self.work[k] is not None:
            self.work[k].wait()
            self.work[k] = None

    def check(self, k):
        """On the SENDING rank, after wait(k): the body this rank handed to launch(k) must have
        fitted its slot (pacx_gather_body skips records that do not fit but still counts them,
        so the count tells).  Synchronises with the device; call it outside timed regions."""
        if self.sent[k] is not None:
            n = int(self.sent[k].item())
            if n > self.capacity():
                raise RuntimeError(f"rank {self.rank}: body of {n} bytes overflowed its "
                                   f"{self.capacity()}-byte gather slot")

    def unpack(self, k):
        """Rank dst, after wait(k): the concatenated bodies in rank order (uint8 tensor)."""
        parts = []
        for buf in self.recv[k]:
            n = int(buf[:HEADER].view(torch.int64).item())
            if n > self.slot - HEADER:
                raise RuntimeError("a rank's body overflowed its gather slot")
            parts.append(buf[HEADER:HEADER + n])
        return torch.cat(parts)
