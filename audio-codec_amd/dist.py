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
