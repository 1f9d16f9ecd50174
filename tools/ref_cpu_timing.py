#!/usr/bin/env python3
"""BUILD CONTAINER ONLY (needs /root/reference): times the reference itself, imported with
the shims of tests/golden/make_golden.py, on BASELINE configs[0] -- test_signals/harpsichord.wav
(44.1 kHz stereo), nMDCTLines 1024, scalar mantissas, no block switching, 128 kb/s/ch, through
its own PCMFile -> PACFile objects (coder/pacfile.py:674-757 driver loop) -- as ONE process and
as EIGHT processes the way its own driver parallelises (`Pool(8)` over independent encodes,
coder/pacfile.py:780-781).  Prints one JSON line; the figures go to BASELINE.md section 3.1.

    cd /tmp && MPLBACKEND=Agg python -B /root/repo/tools/ref_cpu_timing.py [n_hops]
"""
import json
import os
import sys
import tempfile
import time
from multiprocessing import Pool

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests", "golden"))


def one(args):
    n_hops, tag = args
    import make_golden as G          # imports the reference (shims applied there)
    sr, pcm = G.read_wav(os.path.join(G.REF, "test_signals", "harpsichord.wav"))
    d = tempfile.mkdtemp(prefix="reftime_")
    wav = os.path.join(d, "in.wav")
    open(wav, "wb").write(G.wav_bytes(sr, pcm[:n_hops * 1024]))
    t0 = time.perf_counter()
    data, flags = G.ref_encode_file(wav, 128, False, os.path.join(d, f"out_{tag}.pac"))
    dt = time.perf_counter() - t0
    return len(flags) * pcm.shape[1], dt          # channel-blocks written (hops + the flush block), seconds


if __name__ == "__main__":
    n_hops = int(sys.argv[1]) if len(sys.argv) > 1 else 160
    cf1, dt1 = one((n_hops, "single"))
    t0 = time.perf_counter()
    with Pool(8) as pool:
        res = pool.map(one, [(n_hops, f"p{i}") for i in range(8)])
    wall = time.perf_counter() - t0
    print(json.dumps({
        "what": "reference coder/pacfile.py driver loop, harpsichord.wav first %d hops, scalar mantissas, "
                "long blocks, 128 kb/s/ch" % n_hops,
        "host_cores": os.cpu_count(),
        "one_process_cf_per_s": cf1 / dt1, "one_process_s": dt1, "cf": cf1,
        "pool8_cf_per_s": sum(r[0] for r in res) / wall, "pool8_wall_s": wall,
        "pool8_per_process_cf_per_s": [r[0] / r[1] for r in res]}))
