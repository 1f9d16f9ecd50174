"""Whole-batch vs two-shard encode (what multi-GPU sharding relies on): reports
any channel-frame whose outputs differ."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import audio_codec_amd as A

n_frames = 4096
pcm = A.synth.stream(n_frames, 2)
enc = A.engine.Encoder(48000, 128 / 48.0)
planar = torch.as_tensor(A.synth.planar_with_halo(pcm), device=enc.device)
half = n_frames // 2
keys = ("overall", "scale_factor", "bit_alloc", "mantissa", "status")
for rep in range(int(os.environ.get('REPS', '4'))):
    whole = enc.encode(A.engine.PcmView.stream(planar))
    if len(sys.argv) > 1:          # temporaries: the PCM tensors die as soon as encode() returns
        lo = enc.encode(A.engine.PcmView.stream(planar[:, :(half + 1) * 1024].contiguous()))
        hi = enc.encode(A.engine.PcmView.stream(planar[:, half * 1024:].contiguous()))
    else:
        tl = planar[:, :(half + 1) * 1024].contiguous()
        th = planar[:, half * 1024:].contiguous()
        lo = enc.encode(A.engine.PcmView.stream(tl))
        hi = enc.encode(A.engine.PcmView.stream(th))
    torch.cuda.synchronize()
    for k in keys:
        both = torch.cat((lo[k], hi[k]))
        if not torch.equal(both, whole[k]):
            d = (both != whole[k]).reshape(both.shape[0], -1).any(dim=1).nonzero().flatten().cpu().numpy()
            print(rep, k, len(d), "cf differ:", d[:10], "whole", whole[k][d[0]].flatten()[:8].tolist(),
                  "shard", both[d[0]].flatten()[:8].tolist())
print("done")
