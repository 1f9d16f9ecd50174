#!/usr/bin/env python3
"""Diagnose a scalar-coder mismatch of tests/soak_parity.py: python3 tests/diag_soak_case.py <seed> [<seed> ...]
walks the product's and the oracle's .pac stream block by block and reports which channel-blocks differ and how
(test infrastructure: imports the oracle)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import audio_codec_amd as A
import soak_parity as S
from oracle import pac_oracle as po


def blocks(b, hdr):
    out, pos = [], hdr
    while pos < len(b):
        n = int.from_bytes(b[pos:pos + 4], "little")
        out.append(b[pos + 4:pos + 4 + n])
        pos += 4 + n
    return out


def parse(p, b):
    br = po.BitReader(b)
    fl = [br.get(1) for _ in range(3)]
    def unit(bands):
        ov = br.get(4)
        ba, sf, mant = [], [], []
        for k in range(bands.nBands):
            a = br.get(12)
            a = a + 1 if a else 0
            ba.append(a)
            sf.append(br.get(4))
            mant.append([br.get(a) for _ in range(bands.nLines[k])] if a else [])
        return ov, ba, sf, mant
    if fl[1]:
        return fl, [unit(p.sfBandsShort) for _ in range(8)]
    return fl, [unit(p.sfBands)]


for seed in map(int, [a for a in sys.argv[1:] if not a.startswith('--')]):
    c = S.draw_case(seed)
    pcm = S.programme(c["seed"], c["n_hops"], c["n_ch"], c["sr"])
    if c["coder"] == "vq":
        from oracle import pac_oracle_vq as pv
        got = A.pacfile.encode_stream(pcm, c["sr"], c["kbps"], block_switching=True, use_vq=True, use_sbr=c["kbps"] < 128)
        want = pv.encode_stream_vq(pcm, c["sr"], c["kbps"])
        p = po.make_params(c["sr"], c["n_ch"], c["kbps"])
        hdr = len(po.pac_header(p, len(pcm)))
        bg, bw = blocks(got, hdr), blocks(want, hdr)
        print(f"== {c}: sizes {len(got)} {len(want)}, blocks {len(bg)} {len(bw)}")
        def head(b):
            br = po.BitReader(b)
            fl = [br.get(1) for _ in range(3)]
            units = []
            # only the first unit's header is at a fixed place (the shape bits follow it)
            bands = p.sfBandsShort if fl[1] else p.sfBands
            ov = br.get(4)
            ba = []
            for k in range(bands.nBands):
                a = br.get(12)
                ba.append(a + 1 if a else 0)
            return fl, ov, ba
        import torch
        cp_bits = c["kbps"] / (c["sr"] / 1000)
        enc = A.context.encoder(c["sr"], cp_bits, use_vq=True, use_sbr=c["kbps"] < 128) if "--smr" in sys.argv else None
        if enc is not None:
            planar = A.pacfile.device_stream(enc, pcm)
            flags = enc.transient_flags(planar, len(pcm) // 1024, 1024)[1]
            status = enc.encode_vq(A.engine.PcmView.stream(planar, 1024), flags)["status"].cpu().numpy()
            n_ch = c["n_ch"]
            kept = [f for f in range(len(status) // n_ch) if not any(int(status[f * n_ch + k]) & 2 for k in range(n_ch))]
            host = planar.cpu().numpy()
            print(f"  frames {len(status) // n_ch}, kept {len(kept)}")
        for i, (x, y) in enumerate(zip(bg, bw)):
            if x != y:
                hx, hy = head(x), head(y)
                print(f"  block {i} (block-hop {i // c['n_ch']} ch {i % c['n_ch']}) len {len(x)} vs {len(y)}: product {hx}")
                print(f"  {'':>60} oracle  {hy}")
                if enc is not None and hx[0] == hy[0] and hx[2] != hy[2]:
                    f, ch = kept[i // n_ch], i % n_ch
                    fx = hx[0]
                    short = bool(fx[1])
                    blk = host[ch, f * 1024:(f + 2) * 1024]
                    view = A.engine.PcmView.frames(torch.as_tensor(np.ascontiguousarray(blk), device=enc.device).view(1, 1, 2048))
                    smr = enc.smr(view, enc.mdct(view, [tuple(fx)], short=short), short=short).cpu().numpy()[0]
                    bands = p.sfBandsShort if short else p.sfBands
                    nb = bands.nBands
                    n_eff = int(1.45 * 128) if short else 1024
                    if fx[0] or fx[2]:
                        n_eff = int(0.85 * n_eff)
                    budget = cp_bits * n_eff - 4 - 12 * nb
                    sub = blk[448:448 + 256] if short else blk
                    data = po.pcm16_to_fraction(sub)
                    half_n = 128 if short else 1024
                    lines = po.mdct_forward(po.apply_window(data, bool(fx[0]), short, bool(fx[2])), half_n, half_n)[:half_n]
                    ov = po.scale_factor(np.max(np.abs(lines)), 4)
                    lines *= (1 << ov)
                    want_smr = po.calc_smrs(data, lines, ov, c["sr"], bands)
                    print(f"    frame {f}: product SMR {np.round(smr[:nb], 9).tolist()}")
                    print(f"    {'':>9} oracle  SMR {np.round(want_smr[:nb], 9).tolist()}  max difference {np.max(np.abs(want_smr[:nb] - smr[:nb])):.2e} dB")
                    print(f"    budget {budget}: BitAlloc of the product's SMRs {po.bit_alloc(budget, 16, nb, bands.nLines, smr[:nb]).tolist()}, of the oracle's {po.bit_alloc(budget, 16, nb, bands.nLines, want_smr[:nb]).tolist()}")
                    print(f"    samples min/max {int(sub.min())} {int(sub.max())} nonzero {int(np.count_nonzero(sub))}")
        continue
    bs = c["coder"] == "scalar_bs"
    got = A.pacfile.encode_stream(pcm, c["sr"], c["kbps"], block_switching=bs)
    want = po.encode_stream(pcm, c["sr"], c["kbps"], bs)
    p = po.make_params(c["sr"], c["n_ch"], c["kbps"])
    hdr = len(po.pac_header(p, len(pcm)))
    bg, bw = blocks(got, hdr), blocks(want, hdr)
    print(f"== {c}: sizes {len(got)} {len(want)} equal {got == want}, blocks {len(bg)} {len(bw)}, header equal {got[:hdr] == want[:hdr]}")
    shown = 0
    for i, (x, y) in enumerate(zip(bg, bw)):
        if x == y:
            continue
        shown += 1
        if shown > 6:
            continue
        fx, ux = parse(p, x)
        fy, uy = parse(p, y)
        msg = f"  block {i} (hop {i // c['n_ch']} ch {i % c['n_ch']}) flags {fx} vs {fy} len {len(x)} vs {len(y)}"
        if fx == fy:
            for s, (a, b) in enumerate(zip(ux, uy)):
                if a == b:
                    continue
                what = [f"unit {s}:"]
                if a[0] != b[0]:
                    what.append(f"overall {a[0]}/{b[0]}")
                if a[1] != b[1]:
                    what.append(f"ba {a[1]} / {b[1]} (sums {sum(a[1])} {sum(b[1])})")
                elif a[2] != b[2]:
                    what.append(f"sf {a[2]} / {b[2]}")
                else:
                    for k in range(len(a[3])):
                        dd = [(j, u, v) for j, (u, v) in enumerate(zip(a[3][k], b[3][k])) if u != v]
                        if dd:
                            what.append(f"band {k} ba {a[1][k]} sf {a[2][k]} mantissa (line, got, want) {dd[:6]}")
                msg += " | " + " ".join(what)
        print(msg)
        if fx == fy and any(a[1] != b[1] for a, b in zip(ux, uy)) and "--smr" in sys.argv:
            import torch
            enc = A.context.encoder(c["sr"], c["kbps"] / (c["sr"] / 1000))
            host = A.pacfile.device_stream(enc, pcm).cpu().numpy()
            f, ch = i // c["n_ch"], i % c["n_ch"]          # valid while no hop was dropped before this block
            blk = host[ch, f * 1024:(f + 2) * 1024]
            view = A.engine.PcmView.frames(torch.as_tensor(np.ascontiguousarray(blk), device=enc.device).view(1, 1, 2048))
            short = bool(fx[1])
            lines = enc.mdct(view, [tuple(fx)], short=short)
            smr, thr, npk = enc.smr(view, lines, short=short, want_threshold=True, want_peaks=True)
            smr, npk = smr.cpu().numpy()[0], npk.cpu().numpy()[0]
            nb = (p.sfBandsShort if short else p.sfBands).nBands
            for s, (a, b) in enumerate(zip(ux, uy)):
                if a[1] == b[1]:
                    continue
                sub = blk[448 + 128 * s:448 + 128 * s + 256] if short else blk
                st = {}
                if short:
                    p.nMDCTLines = p.nSamplesPerBlock = 128
                po.encode_channel(po.pcm16_to_fraction(sub), p, bool(fx[0]), short, bool(fx[2]), stages=st)
                p.nMDCTLines = p.nSamplesPerBlock = 1024
                print(f"    unit {s}: peaks (product) {npk[s] if short else npk}; stages of the oracle: {sorted(st)}")
                print("      product SMR", np.round(smr[s * nb:(s + 1) * nb], 6).tolist())
                print("      oracle  SMR", np.round(st['smr'][:nb], 6).tolist())
                print("      samples min/max", int(sub.min()), int(sub.max()), "nonzero", int(np.count_nonzero(sub)), "first 24", sub[:24].tolist())
    print(f"  differing blocks {sum(1 for x, y in zip(bg, bw) if x != y)} of {len(bw)}")
