"""A slice of the parity soak under the driver's eyes.

tests/soak_parity.py is run by hand and reproduces from its seeds on one machine only; this test replays a COMMITTED
list of 2400 cases (800 seeds x the three stream coders: scalar, scalar + block switching, gain-shape + block switching
(+ SBR below 128 kb/s)) whose int16 programmes are made with integer arithmetic alone (tests/soak_programmes.py) -- the
sha256 of every programme is committed in tests/golden/soak_slice.json and ASSERTED here.  For every case the product's
.pac bytes AND its decoder's PCM are compared with the oracle's (worker processes on the host cores).  A stream whose
bytes differ must fall, block by block, into the classes where the reference's own FFT rounding noise decides
(tests/soak_parity.py: guard flag up on every differing channel-frame / the product's allocation is the oracle's
BitAlloc of the product's SMRs and those agree to 1e-9 dB / the oracle's SMRs redone with its exactly-zero lines are the
product's / a degenerate block); anything else fails the test, and so do more than 1.5 % of the streams in classes.
The class counts are printed (pytest -s shows them; they are also written to gpurun_out/soak_slice.txt)."""
import json
import multiprocessing as mp
import os
import time

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu


def _workers():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(2, min(16, n - 1))


@pytest.mark.timeout(600)
def test_soak_slice_against_the_oracle():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import audio_codec_amd as A
    import soak_parity as S
    import soak_programmes as P
    cases = json.load(open(os.path.join(GOLDEN, "soak_slice.json")))["cases"]
    assert len(cases) >= 2400 and len({c["seed"] for c in cases}) >= 800
    t0 = time.time()
    pool = mp.get_context("spawn").Pool(_workers())
    pending = [pool.apply_async(P.oracle_side, (c,)) for c in cases]
    noise, real, both_enc_raise, both_dec_raise, byte_diff, n_cf = {}, [], 0, 0, 0, 0
    for case, res in zip(cases, pending):
        pcm = P.programme(case["seed"], case["n_hops"], case["n_ch"], case["sr"])
        assert P.digest(pcm) == case["pcm_sha256"], f"programme of {case} is not the committed one"
        n_cf += case["n_hops"] * case["n_ch"]
        vq = case["coder"] == "vq"
        try:
            got, err = A.pacfile.encode_stream(pcm, case["sr"], case["kbps"], block_switching=case["coder"] != "scalar",
                                               use_vq=vq, use_sbr=vq and case["kbps"] < 128), None
        except Exception as e:                                   # noqa: BLE001
            got, err = None, repr(e)
        want, dec_want = res.get(timeout=500)
        if want is None or got is None:
            if (want is None) != (got is None):
                real.append((case, f"one encoder raised: oracle {dec_want if want is None else 'ok'}, product {err or 'ok'}"))
            else:
                both_enc_raise += 1
            continue
        if got != want:
            byte_diff += 1
            cls = (S.classify_vq_mismatch if vq else S.classify_scalar_mismatch)(A, case, pcm, got, want)
            kinds = sorted({c.split(":")[0].split(" (")[0] for c in cls}) or ["REAL: no differing block found"]
            for k in kinds:
                noise[k] = noise.get(k, 0) + 1
            if any(k.startswith("REAL") for k in kinds):
                real.append((case, "; ".join(cls[:4])))
            elif not vq and P.digest(A.pacfile.decode_stream(want)) != dec_want:
                real.append((case, "the product decodes the oracle's stream differently"))
            continue
        try:
            dec = A.pacfile.decode_stream(got)
            dec_got = P.digest(dec)
        except Exception as e:                                   # noqa: BLE001
            dec, dec_got = None, "raised " + type(e).__name__
        if dec_got.startswith("raised") and dec_want.startswith("raised"):
            both_dec_raise += 1                                  # both decoders refuse the stream (SBR cut in the lower half)
        elif dec_got != dec_want:
            cls = ["REAL: one decoder raised"] if "raised" in dec_got + dec_want else S.classify_decode_mismatch(case, want, dec)
            if cls[0].startswith("decode-tie"):
                noise["decode-tie"] = noise.get("decode-tie", 0) + 1
            else:
                real.append((case, cls[0]))
    pool.close()
    pool.join()
    classified = byte_diff + noise.get("decode-tie", 0)
    lines = [f"soak slice: {len(cases)} streams, {n_cf} channel-frames, {time.time() - t0:.0f} s, {_workers()} oracle workers",
             f"{byte_diff} byte-different streams, " + ("all classified" if not real else f"{len(real)} NOT classified") +
             f": {dict(sorted(noise.items()))}",
             f"{both_enc_raise} streams where both encoders raised, {both_dec_raise} where both decoders raised"]
    lines += [f"  NOT CLASSIFIED {c}: {why}" for c, why in real]
    print("\n".join(lines))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "soak_slice.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    assert not real, real[:5]
    assert classified <= 0.015 * len(cases), f"{classified} of {len(cases)} streams in the noise-decided classes"
