"""N > 1 path on CPU: two gloo ranks shard a stream by hop ranges with a one-hop
halo and gather their packed bitstreams to rank 0 (audio-codec_amd/dist.py).
The encode itself needs a GPU, so each rank's 'bitstream' here is produced by
the oracle; what is under test is the sharding arithmetic and the gather."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_hops, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    import audio_codec_amd as A
    from oracle import pac_oracle as po
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pcm = A.synth.stream(n_hops, 2)
    lo, hi = A.dist.shard_bounds(n_hops, world, rank)
    shard = A.dist.shard_with_halo(pcm, world, rank)             # [2, (hi-lo+1)*1024]
    assert shard.shape == (2, (hi - lo + 1) * 1024)
    p = po.make_params(48000, 2, 128)
    recs = []
    for f in range(hi - lo):
        for ch in range(2):
            x = po.pcm16_to_fraction(shard[ch, f * 1024:f * 1024 + 2048])
            n_bytes, payload = po.pack_channel_block(p, (0, 0, 0), [po.encode_channel(x, p)])
            recs.append(int(n_bytes).to_bytes(4, "little") + payload)
    mine = b"".join(recs)
    body = torch.frombuffer(bytearray(mine + b"\0" * 64), dtype=torch.uint8)
    got = A.dist.gather_bitstream(body, len(mine))
    if rank == 0:
        open(os.path.join(out_dir, "gathered.bin"), "wb").write(got.numpy().tobytes())
    else:
        assert got is None
    # the asynchronous fixed-slot flavour bench.py uses: two steps in flight
    slot = A.dist.slot_bytes((hi - lo) * 2, 128 / 48.0)
    assert len(mine) + A.dist.HEADER <= slot
    g = A.dist.BitstreamGather(slot, torch.device("cpu"))
    for k in range(2):
        g.body(k)[:len(mine)] = torch.frombuffer(bytearray(mine), dtype=torch.uint8)
        g.launch(k, torch.tensor([len(mine)], dtype=torch.int64))
    for k in range(2):
        g.wait(k)
        if rank == 0:
            open(os.path.join(out_dir, f"gathered_async{k}.bin"), "wb").write(g.unpack(k).numpy().tobytes())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds_cover():
    import audio_codec_amd as A
    for n in (1, 7, 8, 9, 4096, 4097):
        for w in (1, 2, 3, 8):
            cuts = [A.dist.shard_bounds(n, w, r) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in cuts) - min(h - l for l, h in cuts) <= 1


@pytest.mark.timeout(600)
def test_two_rank_gather_equals_single_stream(tmp_path):
    from oracle import pac_oracle as po
    import audio_codec_amd as A
    n_hops, world = 5, 2
    mp.spawn(_worker, args=(world, _free_port(), n_hops, str(tmp_path)), nprocs=world, join=True)
    got = open(tmp_path / "gathered.bin", "rb").read()
    # single-rank stream: the same frames in order
    pcm = A.synth.stream(n_hops, 2)
    halo = np.concatenate((np.zeros((1024, 2), np.int16), pcm))
    p = po.make_params(48000, 2, 128)
    want = b""
    for f in range(n_hops):
        for ch in range(2):
            x = po.pcm16_to_fraction(halo[f * 1024:f * 1024 + 2048, ch])
            n_bytes, payload = po.pack_channel_block(p, (0, 0, 0), [po.encode_channel(x, p)])
            want += int(n_bytes).to_bytes(4, "little") + payload
    assert got == want
    for k in range(2):
        assert open(tmp_path / f"gathered_async{k}.bin", "rb").read() == want
