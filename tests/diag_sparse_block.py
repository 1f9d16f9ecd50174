import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import audio_codec_amd as A
import soak_parity as S
from oracle import pac_oracle as po
rng = np.random.default_rng(5)
enc = A.context.encoder(48000, 2.0)
p = po.make_params(48000, 2, 96)
blocks = []
for arg in sys.argv[1:]:                                    # seed:channel of a soak case: its last hop, written twice
    seed, ch = (int(v) for v in arg.split(":"))
    c = S.draw_case(seed)
    pcm = S.programme(c["seed"], c["n_hops"], c["n_ch"], c["sr"])
    a = pcm[-1024:, ch]
    print("case", c, "non-zero at", np.nonzero(a)[0].tolist(), "values", a[np.nonzero(a)[0]].tolist())
    blocks.append(np.concatenate((a, a)))
for trial in range(2):
    a = np.zeros(1024, np.int16)
    pos = rng.choice(1024, 10, replace=False); a[pos] = rng.choice([-1, 1], 10)
    blocks.append(np.concatenate((a, a)))
for trial, blk in enumerate(blocks):
    view = A.engine.PcmView.frames(torch.as_tensor(blk, device=enc.device).view(1, 1, 2048))
    lines = enc.mdct(view)
    smr, thr, npk = enc.smr(view, lines, want_threshold=True, want_peaks=True)
    smr, thr, lines = smr.cpu().numpy()[0], thr.cpu().numpy()[0], lines.cpu().numpy()[0]
    f = po.pcm16_to_fraction(blk)
    othr = po.masked_threshold(f, 1024, 48000)
    st = {}
    po.encode_channel(f, p, stages=st)
    spl = po.spl_of(4 * lines ** 2)
    d = thr - othr
    k = int(np.argmax(np.abs(d)))
    print(trial, "peaks", int(npk[0]), "max |thr - oracle thr|", float(np.abs(d).max()), "at line", k, "product thr", thr[k], "oracle", othr[k],
          "lines equal", float(np.abs(lines - st["mdct"]).max()), "SMR diff", float(np.abs(smr[:17] - st["smr"]).max()))
    for b in range(p.sfBands.nBands):
        if abs(smr[b] - st["smr"][b]) > 1e-6:
            lo, hi = p.sfBands.lowerLine[b], p.sfBands.upperLine[b] + 1
            raw = 96 + 10 * np.log10(np.maximum(4 * lines[lo:hi] ** 2, 1e-300))
            need = smr[b] + thr[lo:hi]                 # the line SPL that would explain the product's band value
            j = int(np.argmin(np.abs(need - raw)))
            print(f"   band {b} lines {lo}..{hi - 1}: product {smr[b]:.4f} oracle {st['smr'][b]:.4f}; unfloored SPL range {raw.min():.2f}..{raw.max():.2f}; "
                  f"closest explanation: line {lo + j} raw SPL {raw[j]:.4f} needs {need[j]:.4f}, thr there {thr[lo + j]:.4f}, |line| {abs(lines[lo + j]):.3e}; thr range {thr[lo:hi].min():.3f}..{thr[lo:hi].max():.3f}")
    ipk = po.sidechain_intensity(f)
    fr = np.fft.rfftfreq(2048, d=1 / 48000)
    pf, ps = po.find_peaks(ipk, fr)
    bad = np.nonzero(np.abs(d) > 1e-9)[0]
    print("   lines where the thresholds differ:", bad[:12].tolist(), "product", np.round(thr[bad[:6]], 4).tolist(), "oracle", np.round(othr[bad[:6]], 4).tolist())
    print("   oracle peaks", len(pf), "SPL range", (float(np.min(ps)), float(np.max(ps))) if len(ps) else None, "product SMR", np.round(smr[:17], 3).tolist())
