#!/usr/bin/env python3
"""Phase split of k_vq from in-kernel s_memtime stamps (debug build, PACX_LIB=...libpacx_dbg.so)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import audio_codec_amd as A

kbps = int(sys.argv[1]) if len(sys.argv) > 1 else 128
enc = A.context.encoder(48000, kbps / 48.0, use_vq=True, use_sbr=kbps < 128)
pcm = A.synth.stream(4096, 2)
planar = torch.as_tensor(A.synth.planar_with_halo(pcm), device=enc.device)
view = A.engine.PcmView.stream(planar)
for _ in range(2):
    enc.encode_vq(view)
torch.cuda.synchronize()
lib = A._lib.load()
out = (ctypes.c_longlong * 32)()
lib.pacx_debug_read_vq.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.pacx_debug_read_vq(out, 32)
frame_mode = os.environ.get("PACX_VQ_FRAME", "1") != "0"
if not frame_mode:          # the stamps of k_vq (both kernels share the slots: k_vq_frame's are read below)
    t = np.array(out[:6], dtype=np.float64)
    names = ["phase A: band gains", "wait at barrier 1", "header (wave 0) + barrier 2", "phase B: bands (busy)",
             "wait for the slowest wave", "hand-over of the string"]
    print("k_vq at %d kb/s, all waves, share of a wave's time:" % kbps)
    for n, v in zip(names, t):
        print("  %-30s %.1f %%" % (n, 100.0 * v / t.sum()))
    inner = np.array(out[8:15], dtype=np.float64)
    inames = ["split arithmetic (fold, norms, angle, bit split)", "sibling bottom splits four leaves at once (or the attempt)",
              "two sibling leaves side by side", "single leaf", "climb to the next pending half", "band set-up (x / gain)",
              "gain quantisation"]
    print("inside the band walk (share of phase B's stamped time):")
    for n, v in zip(inames, inner):
        print("  %-58s %.1f %%" % (n, 100.0 * v / max(inner.sum(), 1.0)))
if frame_mode:
    fn = ["phase A gains + header + roots", "level: classify (wave 0) + barrier", "level: packed passes + barrier",
          "level: scalar arithmetic + children + barrier", "widths / positions (barrier per level)", "fields + gains",
          "hand-over"]
    tf = np.array(out[:7], dtype=np.float64)
    print("k_vq_frame at %d kb/s, thread 0 of every workgroup, share of its time:" % kbps)
    for n, v in zip(fn, tf):
        print("  %-48s %.1f %%" % (n, 100.0 * v / tf.sum()))
    cn = ["leaves <= 16 (4 per pass)", "leaves <= 32 (2 per pass)", "single leaf > 32", "splits, half <= 8", "splits, half <= 16",
          "splits, half <= 32", "splits, half <= 64", "split, half > 64"]
    n_units = 2 * 4096 * 3          # channel-frames x launches of this probe
    print("passes by class, per (sub-)block: count, ticks per pass, ticks in all")
    for c, n in enumerate(cn):
        cnt, tk = out[16 + c], out[8 + c]
        if cnt:
            print("  %-28s %6.2f  %8.0f  %9.0f" % (n, cnt / n_units, tk / cnt, tk / n_units))
    if out[25]:
        print("  %-28s %6.2f  %8.0f  %9.0f" % ("scalar stage (64 splits)", out[25] / n_units, out[24] / out[25], out[24] / n_units))
    sub = ["reads + atan", "angle bits, code, dequantised angle", "log2-tan lookup + bit split", "scan + node allocation", "(children written: rest of the stage)"]
    tot = float(out[24])
    for k, n in enumerate(sub[:4]):
        print("      %-42s %5.1f %% of the scalar stage" % (n, 100.0 * out[26 + k] / tot))
