#!/usr/bin/env python3
"""Diagnose whole-file .pac differences between the GPU path and the oracle: walks both
files block by block and reports which channel-blocks differ and how."""
import sys, os, hashlib, json
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import audio_codec_amd as A
from oracle import pac_oracle as po
name = sys.argv[1] if len(sys.argv) > 1 else "harpsichord"
bs = len(sys.argv) > 2 and sys.argv[2] == "bs"
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
d = np.load(os.path.join(G, f"full_{name}.npz"))
pcm, sr, decl = d["pcm"], int(d["sr"]), int(d["declared"])
max_hops = int(sys.argv[3]) if len(sys.argv) > 3 else None
if max_hops:
    pcm = pcm[:max_hops * 1024]
got = A.pacfile.encode_stream(pcm, sr, 128, block_switching=bs, header_samples=decl)
want = po.encode_stream(pcm, sr, 128, block_switching=bs, header_samples=decl)
print("sizes", len(got), len(want), "equal", got == want)
p = po.make_params(sr, 2, 128)
hdr = len(po.pac_header(p, decl))
def blocks(b):
    out = []; pos = hdr
    while pos < len(b):
        n = int.from_bytes(b[pos:pos + 4], "little"); out.append(b[pos + 4:pos + 4 + n]); pos += 4 + n
    return out
bg, bw = blocks(got), blocks(want)
print("blocks", len(bg), len(bw))
nb = p.sfBands.nBands; nl = p.sfBands.nLines
ndiff = 0
for i, (x, y) in enumerate(zip(bg, bw)):
    if x == y:
        continue
    ndiff += 1
    if ndiff > 12:
        continue
    def parse(b):
        br = po.BitReader(b); fl = [br.get(1) for _ in range(3)]
        if fl[1]:
            subs = []
            nbs = p.sfBandsShort.nBands; nls = p.sfBandsShort.nLines
            for sbk in range(8):
                ov = br.get(4); ba = []; sf = []; mant = []
                for k in range(nbs):
                    a = br.get(12); a = a + 1 if a else 0; ba.append(a); sf.append(br.get(4))
                    mant.append([br.get(a) for _ in range(nls[k])] if a else [])
                subs.append((ov, ba, sf, mant))
            return fl, subs
        ov = br.get(4); ba = []; sf = []; mant = []
        for k in range(nb):
            a = br.get(12); a = a + 1 if a else 0; ba.append(a); sf.append(br.get(4))
            mant.append([br.get(a) for _ in range(nl[k])] if a else [])
        return fl, (ov, ba, sf, mant)
    fx, px = parse(x); fy, py = parse(y)
    msg = f"block {i} (hop {i//2} ch {i%2}) flags {fx} vs {fy} len {len(x)} vs {len(y)}"
    if isinstance(px, list):
        for sbk, (a_, b_) in enumerate(zip(px, py)):
            if a_ == b_: continue
            what = []
            if a_[0] != b_[0]: what.append(f"ov {a_[0]}/{b_[0]}")
            if a_[1] != b_[1]: what.append(f"ba {a_[1]}/{b_[1]}")
            if a_[2] != b_[2]: what.append(f"sf {a_[2]}/{b_[2]}")
            nd = sum(1 for k in range(len(a_[3])) for u, v in zip(a_[3][k], b_[3][k]) if u != v) if a_[1] == b_[1] else -1
            what.append(f"{nd} mant")
            if nd > 0 and sbk == 1:
                for k in range(len(a_[3])):
                    dd = [(jj, u, v) for jj, (u, v) in enumerate(zip(a_[3][k], b_[3][k])) if u != v]
                    if dd:
                        what.append(f"band {k} ba {a_[1][k]} sf {a_[2][k]} first diffs {dd[:6]}")
            msg += f" | sub {sbk}: " + ", ".join(what)
    elif px and py:
        if px[0] != py[0]: msg += f" overall {px[0]} vs {py[0]}"
        if px[1] != py[1]: msg += f" ba differs {px[1]} vs {py[1]}"
        elif px[2] != py[2]: msg += " sf differs"
        else:
            nd = 0; signonly = True
            for k in range(nb):
                for a, b_ in zip(px[3][k], py[3][k]):
                    if a != b_:
                        nd += 1
                        half = 1 << (px[1][k] - 1)
                        if (a & (half - 1)) != (b_ & (half - 1)): signonly = False
            msg += f" {nd} mantissas differ, sign-of-zero only: {signonly}"
    print(msg)
print("differing blocks", ndiff, "of", len(bw))
