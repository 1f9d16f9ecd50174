"""Round-3 GPU tests: what round 2's review asked to see under the driver's eyes.

* every PACX_* path switch (alternate kernels kept alive behind environment switches) gives the bytes of
  the default path -- the hand-run record profiles/r02_env_switch_parity.txt as a test, including the
  combinations ADVICE r2 listed as unrecorded (PACX_SPLIT_SHORT=0 with PACX_VQ_FUSE_ALLOC=0 and with
  PACX_FUSE_TAIL=0, PACX_VQ_FUSE_ALLOC=0 in the split branch);
* pacx_reserve: a hipMalloc that fails mid-sequence releases what the call had allocated, leaves the
  handle usable and the device memory where it was.
"""
import ctypes
import itertools
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def A():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import audio_codec_amd as a
    return a


def _mixed_stream():
    """a block-switched programme: the castanet excerpt (real attacks) followed by synthetic bursts"""
    ex = np.load(os.path.join(GOLDEN, "excerpt_castanet.npz"))
    pcm = ex["pcm"][:24 * 1024]
    rng = np.random.default_rng(5)
    t = np.arange(16 * 1024)
    tone = 0.3 * np.sin(2 * np.pi * 880 * t / 44100)
    burst = np.zeros(len(t))
    for k in (3000, 7000, 7100, 12000):
        burst[k:k + 200] = rng.standard_normal(200)
    syn = np.stack([tone + burst, 0.5 * tone - burst], axis=1)
    syn = np.clip(np.round(syn * 16000), -32767, 32767).astype(np.int16)
    return np.concatenate((pcm, syn)), int(ex["sr"])


def _set(monkeypatch, env):
    for k in ("PACX_SPLIT_SHORT", "PACX_FUSE_TAIL", "PACX_VQ_FUSE_ALLOC", "PACX_VQ_FRAME", "PACX_VQ_BFS",
              "PACX_VQ_DEC_FRAME"):
        if k in env and env[k] is not None:
            monkeypatch.setenv(k, env[k])
        else:
            monkeypatch.delenv(k, raising=False)


def test_scalar_coder_path_switches(A, monkeypatch):
    """scalar coder, block-switched and all-long batches: PACX_FUSE_TAIL x PACX_SPLIT_SHORT"""
    pcm, sr = _mixed_stream()
    for bs in (True, False):
        _set(monkeypatch, {})
        want = A.pacfile.encode_stream(pcm, sr, 128, block_switching=bs)
        n = 0
        for fuse, split in itertools.product((None, "0", "1"), (None, "0")):
            _set(monkeypatch, {"PACX_FUSE_TAIL": fuse, "PACX_SPLIT_SHORT": split})
            assert A.pacfile.encode_stream(pcm, sr, 128, block_switching=bs) == want, (bs, fuse, split)
            n += 1
        assert n == 6
        _set(monkeypatch, {})
        monkeypatch.setenv("PACX_BS_TWO_STREAMS", "0")   # round 2's four-stream schedule of block-switched batches
        assert A.pacfile.encode_stream(pcm, sr, 128, block_switching=bs) == want, (bs, "four streams")
        monkeypatch.delenv("PACX_BS_TWO_STREAMS")
        for fuse, one in itertools.product((None, "1"), ("1", "0")):   # all-long batches: the whole step on the caller's stream
            _set(monkeypatch, {"PACX_FUSE_TAIL": fuse})                  # (a handle's default) / the side chain forked to a second one
            monkeypatch.setenv("PACX_ONE_STREAM", one)
            assert A.pacfile.encode_stream(pcm, sr, 128, block_switching=bs) == want, (bs, fuse, "one stream", one)
            monkeypatch.delenv("PACX_ONE_STREAM")
    _set(monkeypatch, {})


@pytest.mark.parametrize("kbps", [96, 128])
def test_gain_shape_coder_path_switches(A, monkeypatch, kbps):
    """gain-shape coder (+ SBR at 96 kb/s), block-switched and all-long: PACX_SPLIT_SHORT x
    PACX_VQ_FUSE_ALLOC x (PACX_VQ_FRAME: unset = k_vq_frame, 2 = k_vq_frame2, 0 = k_vq with PACX_VQ_BFS) -- and
    PACX_VQ_ONE_LAUNCH=1 (one coder launch behind both chains); then the two decoders on the default stream"""
    pcm, sr = _mixed_stream()
    for bs in (True, False):
        _set(monkeypatch, {})
        want = A.pacfile.encode_stream(pcm, sr, kbps, block_switching=bs, use_vq=True, use_sbr=kbps < 128)
        for split, alloc, (frame, bfs) in itertools.product((None, "0"), (None, "0", "1"),
                                                            ((None, None), ("2", None), ("0", "0"), ("0", "1"))):
            _set(monkeypatch, {"PACX_SPLIT_SHORT": split, "PACX_VQ_FUSE_ALLOC": alloc, "PACX_VQ_FRAME": frame,
                               "PACX_VQ_BFS": bfs})
            got = A.pacfile.encode_stream(pcm, sr, kbps, block_switching=bs, use_vq=True, use_sbr=kbps < 128)
            assert got == want, (bs, split, alloc, frame, bfs)
        _set(monkeypatch, {})
        monkeypatch.setenv("PACX_VQ_ONE_LAUNCH", "1")
        assert A.pacfile.encode_stream(pcm, sr, kbps, block_switching=bs, use_vq=True, use_sbr=kbps < 128) == want
        monkeypatch.delenv("PACX_VQ_ONE_LAUNCH")
        ref = A.pacfile.decode_stream(want)
        _set(monkeypatch, {"PACX_VQ_DEC_FRAME": "0"})
        assert np.array_equal(A.pacfile.decode_stream(want), ref)
    _set(monkeypatch, {})


def test_reserve_failure_releases_everything(A):
    """pacx_reserve with a workspace the card cannot hold: the first buffers (lines, 8 KB per channel-frame)
    fit, a later one does not -- the call must fail with PACX_E_HIP, free what it had allocated, and the
    handle must go on working (round 2: the earlier buffers stayed allocated and ws_cf kept its old value)."""
    import torch
    enc = A.engine.Encoder(48000, 128 / 48.0)
    pcm = A.synth.stream(8, 2)
    planar = torch.as_tensor(A.synth.planar_with_halo(pcm), device=enc.device)
    view = A.engine.PcmView.stream(planar)
    before = enc.encode_pack(view)
    want = (before["payload"].cpu().numpy().copy(), before["n_bytes"].cpu().numpy().copy())
    torch.cuda.synchronize()
    free0, total = torch.cuda.mem_get_info()
    # lines alone: n * 8 KB must fit, lines + SMRs + maskers (n * ~20.4 KB) must not
    n = int(free0 * 0.7) // 8192
    assert n * 8192 < free0 < n * 20000
    rc = enc.lib.pacx_reserve(enc.h, ctypes.c_int64(n))
    assert rc != 0
    msg = enc.lib.pacx_last_error(enc.h).decode()
    assert "hipMalloc" in msg and "released" in msg, msg
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free1 >= free0 - (64 << 20), f"{(free0 - free1) >> 20} MiB still held after the failed reserve"
    # absurd counts are refused before anything is allocated
    assert enc.lib.pacx_reserve(enc.h, ctypes.c_int64(1 << 40)) != 0
    after = enc.encode_pack(view)                  # reserves again, small
    nb = after["n_bytes"].cpu().numpy()
    assert np.array_equal(nb, want[1])
    got = after["payload"].cpu().numpy()
    assert all(np.array_equal(got[i, :nb[i]], want[0][i, :nb[i]]) for i in range(len(nb)))


# ------------------------------------------------------------ function-level mirrors (VERDICT r2 missing #3)
def test_mdct_module_self_test(A, tables):
    """The reference's own mdct.py self-test (coder/mdct.py:86-107) run against the mirror: TDAC round trip
    with a = b = 4, MDCTslow == MDCT and their inverses on the 20-sample ramp block -- and the outputs the
    reference itself produced for that block (tests/golden/tables.npz, made by running it)."""
    M = A.mdct
    x = np.array([0, 1, 2, 3, 4, 4, 4, 4, 3, 1, -1, -3], dtype=np.float64)
    frame_size, noverlap = 8, 4
    x = np.concatenate([np.zeros(noverlap), x, np.zeros(noverlap)])
    x_hat = np.zeros_like(x)
    for i in range(0, len(x) - noverlap, noverlap):
        data = x[i:i + frame_size]
        mdct = M.MDCTslow(data, 4, 4, isInverse=False)
        x_hat[i:i + frame_size] += M.MDCTslow(mdct, 4, 4, isInverse=True) / 2
    assert np.allclose(x, x_hat)
    a = b = len(x) // 2
    mdct_a, mdct_b = M.MDCTslow(x, a, b), M.MDCT(x, a, b)
    assert np.allclose(mdct_a, mdct_b)
    assert np.allclose(M.MDCTslow(mdct_a, a, b, isInverse=True), M.MDCT(mdct_b, a, b, isInverse=True))
    # against the reference's own numbers
    ramp = tables["mdct_ramp_in"]
    for got, want in ((M.MDCT(ramp, 10, 10), tables["mdct_ramp_fast"]), (M.MDCTslow(ramp, 10, 10), tables["mdct_ramp_slow"]),
                      (M.IMDCT(tables["mdct_ramp_fast"], 10, 10), tables["imdct_ramp_fast"])):
        assert got.shape == want.shape and np.max(np.abs(got - want)) <= 1e-13 * max(1.0, np.max(np.abs(want)))
    # unequal halves (the reference's formula has n0 = (b + 1) / 2)
    from oracle import pac_oracle as po
    rng = np.random.default_rng(2)
    y = rng.standard_normal(24)
    assert np.max(np.abs(M.MDCT(y, 16, 8) - po.mdct_slow(y, 16, 8))) < 1e-13


@pytest.mark.parametrize("half", [1024, 128])
def test_imdct_at_the_codec_sizes(A, half):
    """mdct.IMDCT / MDCT(isInverse=True) at a = b = 1024 and 128 (k_imdct_long / k_imdct_short of the decode
    path, unwindowed) against the oracle's restatement of coder/mdct.py:56-62, the timing loop's round trip
    (coder/mdct.py:109-122) and TDAC with the sine window."""
    from oracle import pac_oracle as po
    rng = np.random.default_rng(half)
    for _ in range(4):
        lines = rng.standard_normal(half) * 10.0 ** rng.uniform(-4, 0)
        got = A.mdct.IMDCT(lines, half, half)
        want = po.mdct_inverse(lines, half, half)
        assert got.shape == (2 * half,)
        assert np.max(np.abs(got - want)) <= 2e-12 * np.max(np.abs(want))
    # windowed TDAC: three overlapping blocks, the middle hop comes back
    x = rng.standard_normal(4 * half)
    w = A.window.SineWindow(np.ones(2 * half))
    rec = np.zeros(4 * half)
    for i in range(3):
        blk = x[i * half:(i + 2) * half]
        rec[i * half:(i + 2) * half] += w * A.mdct.IMDCT(A.mdct.MDCT(w * blk, half, half), half, half)
    assert np.max(np.abs(rec[half:3 * half] - x[half:3 * half])) < 1e-11


def test_quantize_module_self_test(A, tables):
    """coder/quantize.py:283-319 against the mirror, plus the values the reference itself printed for it"""
    Q = A.quantize
    inputs = tables["quant_in"]
    for bits in (8, 12):
        scal = [Q.DequantizeUniform(Q.QuantizeUniform(float(v), bits), bits) for v in inputs]
        vec = Q.vDequantizeUniform(Q.vQuantizeUniform(inputs, bits), bits)
        assert np.array_equal(vec, np.array(scal))
        assert np.array_equal(vec, tables[f"dequant_v{bits}"])
        assert [Q.QuantizeUniform(float(v), bits) for v in inputs] == tables[f"quant_u{bits}"].tolist()
    deq = []
    for v in inputs:
        v = float(v)
        scale = Q.ScaleFactor(v)
        fp = Q.DequantizeFP(scale, Q.MantissaFP(v, scale))
        m = Q.Mantissa(v, scale)
        bfp = Q.Dequantize(scale, m)
        assert abs(fp - v) < 0.05 and abs(bfp - v) < 0.05
        mv = Q.vMantissa(np.array([v]), scale)
        assert int(mv[0]) == m
        assert abs(Q.vDequantize(scale, mv)[0] - bfp) < 1e-5            # the reference's own assertion
        assert Q.vDequantize(scale, mv)[0] == bfp
        deq.append(bfp)
    assert np.array_equal(np.array(deq), tables["quant_deq_3_5"])
    assert Q.DequantizeUniform(5, 0) == 0 and Q.QuantizeUniform(0.3, 0) == 0


def test_dequantizers_against_the_oracle(A):
    """vDequantize / vDequantizeUniform / MantissaFP / DequantizeFP over random codes at the codec's widths
    (nScaleBits 4, 2..16 mantissa bits) and the homework's (3, 5): bit-equal to the oracle's restatement"""
    from oracle import pac_oracle as po
    rng = np.random.default_rng(9)
    for nsb, nmb in [(4, b) for b in (2, 3, 5, 8, 12, 16)] + [(3, 5), (2, 4)]:
        r = (1 << nsb) - 1 + nmb
        codes = rng.integers(0, 1 << r, 4000)
        assert np.array_equal(A.quantize.vDequantizeUniform(codes, r), po.dequantize_uniform_vec(codes, r))
        mant = rng.integers(0, 1 << nmb, 4000)
        for scale in range(1 << nsb):
            assert np.array_equal(A.quantize.vDequantize(scale, mant, nsb, nmb), po.dequantize_vec(scale, mant, nsb, nmb)), (nsb, nmb, scale)


def test_fp_pair_matches_a_plain_python_restatement(A):
    """MantissaFP / DequantizeFP (coder/quantize.py:130-175) against the formula written out with Python ints"""
    def mant_fp(x, scale, nsb, nmb):
        r = 2 ** nsb - 1 + nmb
        code = 2 ** (r - 1) - 1 if abs(x) >= 1 else int(((2 ** r - 1) * abs(x) + 1) // 2)
        s = (1 << (nmb - 1)) if x < 0 else 0
        if scale == 2 ** nsb - 1:
            return s + (code & (2 ** (nmb - 1) - 1))
        return s + ((code >> (r - scale - nmb - 1)) & (2 ** (nmb - 1) - 1))

    def deq_fp(scale, m, nsb, nmb):
        r = 2 ** nsb - 1 + nmb
        a = (1 << (r - 1)) if m & (1 << (nmb - 1)) else 0
        code = m & (2 ** (nmb - 1) - 1)
        a += code << max(r - scale - nmb - 1, 0)
        if scale != 2 ** nsb - 1:
            a += 1 << (r - scale - 2)
        if r - scale - nmb - 2 > 0:
            a += 1 << (r - scale - nmb - 2)
        sign = -1 if a & (1 << (r - 1)) else 1
        return sign * 2 * (a & (2 ** (r - 1) - 1)) / (2 ** r - 1)

    rng = np.random.default_rng(4)
    for nsb, nmb in ((3, 5), (4, 8), (2, 3)):
        for _ in range(300):
            x = float(rng.uniform(-1.1, 1.1) * 10.0 ** rng.uniform(-4, 0))
            scale = A.quantize.ScaleFactor(x, nsb, nmb)
            m = A.quantize.MantissaFP(x, scale, nsb, nmb)
            assert m == mant_fp(x, scale, nsb, nmb)
            assert A.quantize.DequantizeFP(scale, m, nsb, nmb) == deq_fp(scale, m, nsb, nmb)


def test_bitalloc_sbr_mirror(A, tables):
    """BitAlloc_SBR (coder/bitalloc.py:123-145): omitted bands count one line -- written into the caller's
    array, as the reference does -- then BitAlloc; against the oracle's allocation of the same problem"""
    from oracle import pac_oracle as po
    rng = np.random.default_rng(3)
    n_lines0 = tables["bands_1024_48000_nLines"].astype(int)
    omitted = [int(b) for b in tables["bands_1024_48000_omitted"]]
    for _ in range(40):
        smr = rng.uniform(-20, 30, len(n_lines0))
        budget = float(rng.uniform(400, 3000))
        n_lines = n_lines0.copy()
        got = A.bitalloc.BitAlloc_SBR(budget, 16, len(n_lines), n_lines, smr, omitted)
        assert all(n_lines[b] == 1 for b in omitted)
        want_lines = n_lines0.copy()
        want_lines[omitted] = 1
        assert got.tolist() == po.bit_alloc(budget, 16, len(want_lines), want_lines, smr).tolist()


def test_encode_single_channel_sbr_mirror(A):
    """EncodeSingleChannel_SBR (coder/codec.py:426-555) = one channel of Encode_SBR, both coders"""
    from oracle import pac_oracle_vq as pv
    pcm = A.synth.stream(3, 2)
    p = pv.make_params_vq(48000, 2, 96)
    from oracle import pac_oracle as po
    x = [po.pcm16_to_fraction(pcm[1024:3072, ch]) for ch in range(2)]
    both = A.codec.Encode_SBR(x, p)
    for ch in range(2):
        one = A.codec.EncodeSingleChannel_SBR(x[ch], p)
        assert one[0].tolist() == both[0][ch].tolist() and one[1] == both[1][ch] and one[2] == both[2][ch]
        assert one[3] == both[3][ch]
        # the oracle, like the reference (BitAlloc_SBR, coder/bitalloc.py:141-143), leaves p.sfBands.nLines at 1
        # for the omitted bands from here on; the mirror must go on by the line ranges, as the reference does
        ba, idx, bits, ov = pv.encode_channel_sbr_vq(x[ch], p)
        assert one[0].tolist() == list(ba) and one[3] == ov and one[1] == [list(map(int, i)) for i in idx]
    assert [int(p.sfBands.nLines[b]) for b in p.omittedBands] == [1] * len(p.omittedBands)
    again = A.codec.Encode_SBR(x, p)
    assert [a.tolist() for a in again[0]] == [b.tolist() for b in both[0]] and again[1] == both[1]
    # the scalar-mantissa branch of the same function (useVQ False)
    ps = pv.make_params_vq(48000, 2, 96)
    ps.useVQ = False
    try:
        s2 = A.codec.Encode_SBR(x, ps)
        for ch in range(2):
            s1 = A.codec.EncodeSingleChannel_SBR(x[ch], ps)
            assert s1[0].tolist() == s2[0][ch].tolist() and s1[1].tolist() == s2[1][ch].tolist()
            assert s1[2].tolist() == s2[2][ch].tolist() and s1[3] == s2[3][ch]
    except TypeError as e:                     # an omitted band got bits: the reference raises there, and so do we
        assert str(e) == A._lib.REF_SCALAR_SBR_ERROR


@pytest.mark.parametrize("name", ["castanet", "harpsichord", "quar48_1", "spmg"])
def test_stream_flags_match_reference(A, name):
    """(moved from the CPU suite: the detector of the mirror runs on the GPU now) the block-switching flags of
    the excerpts as the reference's own driver loop produced them"""
    from conftest import load_excerpt
    ex = load_excerpt(name)
    pcm = ex["pcm"]
    pcm = np.concatenate((pcm, np.zeros((-len(pcm) % 1024, 2), pcm.dtype)))
    got = A.pacfile.stream_flags(pcm, True)
    assert got[:-1].tolist() == ex["flags_bs"].tolist()
    assert got[-1].tolist() == [0, 0, 0]
    assert not A.pacfile.stream_flags(pcm, False).any()


def test_transient_detector_mirror_matches_oracle(A):
    """detect_transients.parTransientDetect on the GPU (k_transient_f64: any float block, any threshold,
    both axes; mean in NumPy's pairwise order) against the oracle's restatement of coder/detect_transients.py"""
    from oracle import pac_oracle as po
    rng = np.random.default_rng(0)
    for t in range(200):
        blk = np.zeros((2, 2048))
        blk[:, :1024] = rng.standard_normal((2, 1024)) * 10.0 ** rng.uniform(-3, 0)
        if t % 3 == 0:
            blk[rng.integers(2), rng.integers(1024)] = rng.uniform(0.2, 1.0)
        if t % 17 == 0:
            blk[:] = 0
        got = A.detect_transients.parTransientDetect(blk)
        want = po.transient_detect(blk)
        assert got == want and type(got) is type(want), t
    # other shapes, thresholds, axis=0 (the reference's own __main__ calls it that way) and exact ties
    for t in range(100):
        n_ch, n = int(rng.integers(1, 5)), int(rng.integers(3, 3000))
        blk = rng.standard_normal((n_ch, n)) * (rng.random((n_ch, n)) < rng.uniform(0.01, 1.0))
        thr = float(rng.uniform(1.0, 8.0))
        assert A.detect_transients.parTransientDetect(blk, thr) == po.transient_detect(blk, thr), (t, n_ch, n)
        assert A.detect_transients.parTransientDetect(blk.T, thr, axis=0) == po.transient_detect(blk, thr)
    tie = np.zeros((1, 600))
    tie[0, 0] = 4.5
    tie[0, 1:500] = 1.0                      # peak / mean of the first 500 = 4.5 / ((4.5 + 499) / 500)
    assert A.detect_transients.parTransientDetect(tie) == po.transient_detect(tie)


# ------------------------------------------------------------ host memory to host memory (VERDICT r2 missing #5)
@pytest.mark.parametrize("coder", ["scalar", "scalar_bs", "shipped128", "shipped96"])
def test_host_stream_chunks_equal_one_batch(A, coder):
    """streaming.HostStreamEncoder: a stream cut into chunks of 7 hops (the one-hop halo and the transient decisions
    carried from chunk to chunk on the device, the copies of neighbouring chunks overlapping the kernels) gives the
    bytes of the one-batch path -- and with that the reference's (the excerpt is one of its own test files)"""
    pcm, sr = _mixed_stream()
    kw = {"scalar": dict(block_switching=False), "scalar_bs": dict(block_switching=True),
          "shipped128": dict(block_switching=True, use_vq=True),
          "shipped96": dict(block_switching=True, use_vq=True, use_sbr=True)}[coder]
    kbps = 96 if coder == "shipped96" else 128
    want = A.pacfile.encode_stream(pcm, sr, kbps, **kw)
    for chunk in (7, 16, 64):
        assert A.pacfile.encode_stream(pcm, sr, kbps, chunk_hops=chunk, **kw) == want, (coder, chunk)


def test_host_stream_zero_copy_api(A):
    """the caller fills the pinned staging buffers itself; slots in flight are refused until their result is taken"""
    import torch
    enc = A.engine.Encoder(48000, 128 / 48.0)
    pcm = A.synth.stream(24, 2)
    hs = A.streaming.HostStreamEncoder(enc, 2, 8, depth=2)
    want = A.pacfile.encode_stream(pcm, 48000, 128)
    head_len = len(want) - sum(len(b) for b in A.streaming.HostStreamEncoder(enc, 2, 8).encode(pcm))
    got = []
    for i in range(3):
        k = i % 2
        if i >= 2:
            got.append(bytes(hs.result(k)))
        hs.input(k)[:] = pcm[i * 8 * 1024:(i + 1) * 8 * 1024].T
        hs.submit(k)
        with pytest.raises(RuntimeError):
            hs.submit(k)                                  # still in flight
    got.append(bytes(hs.result(1)))
    got.append(bytes(hs.result(0)))
    body = b"".join(got)
    assert want[head_len:head_len + len(body)] == body     # the three chunks; the file's last two blocks follow


def test_bench_line_fields(A):
    """one small default-form bench run: the line carries roofline, verified_cf, the two steps in flight of the default
    timed path with the one-step-in-flight figure beside it, and the host-to-host figure (never `value`)"""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--frames", "256", "--steps", "2", "--repeats", "2", "--min-seconds", "0.2",
                        "--warmup", "1", "--host-stream-frames", "1024", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-500:], r.stderr[-1500:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["verified_cf"] == 32 and d["dtype"] == "f64"
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1
    assert d["config"]["steps_in_flight"] == 2 and "2 steps in flight" in d["config"]["launch"]
    assert 0 < d["config"]["value_one_step_in_flight"]
    assert d["config"]["host_to_host_cf_per_s"] > 0 and d["config"]["host_to_host"]["chunk_cf"] == 2048
    assert d["vs_baseline"] is None
    assert d["config"]["decode_cf_per_s"] > 0 and d["config"]["decode"]["pcm_samples"] == (256 + 1) * 1024
    q = d["config"]["ms_per_step_quantiles"]           # how the timed regions fell (two timing modes with two steps in flight)
    assert 0 < q["p10"] <= q["p25"] <= q["p50"] <= q["p75"] <= q["p90"]


# ------------------------------------------------------------ PACX_ST_GUARD of the gain-shape coder (VERDICT r2 missing #4)
@pytest.mark.parametrize("kbps", [96, 128])
def test_gain_shape_guard_flag(A, kbps):
    """a handle with pacx_config.guard and use_vq: same bytes as without, PACX_ST_GUARD on a small share of the
    channel-frames (split angles, band gains, pulse-search floors and ties within rounding distance of a boundary,
    lines at rounding-noise level), and certainly on a frame built to have exactly-zero lines in coded bands"""
    import torch
    pcm, sr = _mixed_stream()
    pcm = np.concatenate((pcm, A.synth.stream(40, 2, sample_rate=sr)))
    outs = {}
    for guard in (False, True):
        enc = A.engine.Encoder(sr, kbps / (sr / 1000), use_vq=True, use_sbr=kbps < 128, guard=guard)
        planar = A.pacfile.device_stream(enc, pcm)
        flags = enc.transient_flags(planar, len(pcm) // 1024, 1024)[1]
        o = enc.encode_vq(A.engine.PcmView.stream(planar, 1024), flags)
        outs[guard] = {k: o[k].cpu().numpy() for k in ("payload", "n_bytes", "status")}
    nb = outs[True]["n_bytes"]
    assert np.array_equal(nb, outs[False]["n_bytes"])
    assert all(np.array_equal(outs[True]["payload"][i, :nb[i]], outs[False]["payload"][i, :nb[i]]) for i in range(len(nb)))
    assert not (outs[False]["status"] & A._lib.ST_GUARD).any()
    flagged = int(np.count_nonzero(outs[True]["status"] & A._lib.ST_GUARD))
    print(f"gain-shape PACX_ST_GUARD at {kbps} kb/s: {flagged} of {len(nb)} channel-frames flagged")
    assert flagged <= 0.10 * len(nb)
    # a period-4 square wave: three quarters of its MDCT lines are zero in exact arithmetic
    sq = np.tile(np.array([9000, 9000, -9000, -9000], np.int16), 8 * 256)[:, None].repeat(2, axis=1)
    sq[:, 1] //= 2
    enc = A.engine.Encoder(sr, kbps / (sr / 1000), use_vq=True, use_sbr=kbps < 128, guard=True)
    planar = A.pacfile.device_stream(enc, sq)
    st = enc.encode_vq(A.engine.PcmView.stream(planar, 1024), None)["status"].cpu().numpy()
    assert (st[2:-4] & A._lib.ST_GUARD).all(), st


def test_encoder_pool_two_batches_in_flight(A):
    """engine.EncoderPool: independent batches alternating between two handles on two streams give the bytes of one
    handle doing them one after the other"""
    import torch
    pool = A.engine.EncoderPool(2, 48000, 128 / 48.0)
    one = A.engine.Encoder(48000, 128 / 48.0)
    views, outs, want = [], [], []
    for i in range(4):
        pcm = A.synth.stream(32, 2, seed=900 + i)
        planar = torch.as_tensor(A.synth.planar_with_halo(pcm), device=one.device)
        v = A.engine.PcmView.stream(planar)
        views.append(v)
        o = one.encode_pack(v)
        want.append((o["n_bytes"].cpu().numpy().copy(), o["payload"].cpu().numpy().copy()))
    torch.cuda.synchronize()
    outs = [pool.encs[k].alloc_outputs(views[0].n_cf, with_payload=True) for k in range(2)]
    got = []
    pool.align(50)                                     # both slots' first calls start together (a gate event): same bytes
    for i, v in enumerate(views):
        k = pool.next()
        if i >= 2:                                     # slot k's buffers are about to be reused: take its result first
            pool.wait(k)
            torch.cuda.current_stream().synchronize()
            got.append((outs[k]["n_bytes"].cpu().numpy().copy(), outs[k]["payload"].cpu().numpy().copy()))
        with pool.slot(k) as enc:
            enc.encode_pack(v, None, outs[k])
    pool.synchronize()
    for k in (0, 1):
        got.append((outs[k]["n_bytes"].cpu().numpy().copy(), outs[k]["payload"].cpu().numpy().copy()))
    assert len(pool) == 2 and len(got) == 4
    for (nb, pay), (nb_w, pay_w) in zip(got, want):
        assert np.array_equal(nb, nb_w)
        assert all(np.array_equal(pay[i, :nb[i]], pay_w[i, :nb[i]]) for i in range(len(nb)))


# ------------------------------------------------------------------ Decode_SBR, scalar branch (VERDICT r2 missing #7)
SBRD = np.load(os.path.join(GOLDEN, "sbr_scalar_decode.npz"))


@pytest.mark.parametrize("tag", [str(c) for c in SBRD["stream_cases"]])
def test_scalar_sbr_streams_with_coded_omitted_bands_decode_to_the_references_pcm(A, tag):
    """streams whose long blocks code their omitted bands (one mantissa each, coder/pacfile.py:203-205) through
    pacfile.decode_stream: the PCM the REFERENCE's own reader + Decode_SBR (coder/codec.py:117-134, useVQ off) made
    of the same bytes (tests/golden/make_golden.py --sbr-scalar-decode)"""
    pac = bytes(SBRD[f"pac_{tag}"])
    pcm = A.pacfile.decode_stream(pac)
    want = SBRD[f"pcm_{tag}"]
    assert pcm.shape == want.shape
    assert np.array_equal(pcm, want), int(np.abs(pcm.astype(int) - want.astype(int)).max())


@pytest.mark.parametrize("sr", [48000, 44100, 32000])
def test_decode_sbr_mirror_scalar_branch(A, sr):
    """codec.Decode_SBR with useVQ off on random code sets, against the reference's own outputs; allocation row 0
    has no coded omitted band (PACFile.Decode would send it to codec.Decode; the function takes it all the same)"""
    cp = A.audiofile.CodingParams()
    cp.sampleRate, cp.nChannels, cp.nMDCTLines, cp.nSamplesPerBlock = sr, 1, 1024, 1024
    cp.nScaleBits, cp.nMantSizeBits, cp.targetBitsPerSample = 4, 12, 96 / (sr / 1000)
    cp.useSBR, cp.useVQ = True, False
    cp.sfBands = A.psychoac.ScaleFactorBands(A.psychoac.AssignMDCTLinesFromFreqLimits(1024, sr))
    cp.sfBandsShort = A.psychoac.ScaleFactorBands(A.psychoac.AssignMDCTLinesFromFreqLimits(128, sr))
    cp.omittedBands = A.pacfile.omitted_bands(cp.sfBands)
    want = SBRD[f"fn_{sr}_block"]
    worst = 0.0
    for k in range(len(want)):
        fl = SBRD[f"fn_{sr}_flags"][k]
        got = A.codec.Decode_SBR(SBRD[f"fn_{sr}_sf"][k], SBRD[f"fn_{sr}_ba"][k], SBRD[f"fn_{sr}_mant"][k],
                                 int(SBRD[f"fn_{sr}_overall"][k]), None, cp, bool(fl[0]), False, bool(fl[1]))
        scale = max(float(np.abs(want[k]).max()), 1e-300)
        worst = max(worst, float(np.abs(got - want[k]).max()) / scale)
    assert worst <= 1e-12, worst                      # the IMDCT's tolerance, relative to the block's peak


def test_scalar_sbr_decode_routing_and_refusal(A):
    """pacx_decode_batch stays codec.Decode on an SBR handle; PACFile.Decode's routing picks Decode_SBR only for
    long blocks with a coded omitted band (lines compared with the oracle's); at 96 kHz Decode_SBR raises
    IndexError (the cut lies in the lower half) and so do the mirrors"""
    import torch
    from oracle import pac_oracle as po, pac_oracle_vq as pv
    tag = "harpsichord_96_long"
    pac = bytes(SBRD[f"pac_{tag}"])
    cp, pos = A.pacfile.parse_header(pac)
    enc = A.context.encoder_for_params(cp)
    offs, sizes = A.pacfile.record_chain(pac, pos, enc.payload_stride)
    body = torch.frombuffer(bytearray(pac) + bytearray(8), dtype=torch.uint8).to(enc.device)
    codes = enc.unpack(body, torch.tensor(sizes, dtype=torch.int32, device=enc.device),
                       torch.tensor(offs, dtype=torch.int64, device=enc.device))
    plain = enc.decode(codes, cp.nChannels, want_blocks=True, want_pcm=False).cpu().numpy()
    extra = {}
    routed = enc.decode(codes, cp.nChannels, want_blocks=True, want_pcm=False, extra=extra).cpu().numpy()
    p, _, _ = po.parse_header(pac)
    ba = codes["bit_alloc"].cpu().numpy()
    lines = extra["lines"].cpu().numpy()
    n_sbr = 0
    for i in range(len(sizes)):
        br = po.BitReader(pac[offs[i]:offs[i] + sizes[i]])
        fl = (br.get(1), br.get(1), br.get(1))
        sf, alloc, mant, ov = po.parse_block_body(br, p, False)
        assert list(ba[i, :len(alloc)]) == list(alloc)
        coded = any(alloc[b] for b in p.omittedBands)
        want_plain = po.decode_block(p, sf, alloc, mant, ov, *fl)
        assert np.abs(plain[i] - want_plain).max() <= 1e-12 * max(np.abs(want_plain).max(), 1e-300)
        want = po.decode_any_block(p, sf, alloc, mant, ov, *fl)
        assert np.abs(routed[i] - want).max() <= 1e-12 * max(np.abs(want).max(), 1e-300), i
        if coded:
            n_sbr += 1
            ln = np.zeros(1024)
            at = 0
            for b in range(p.sfBands.nBands):
                n = 1 if b in p.omittedBands else p.sfBands.nLines[b]
                if alloc[b]:
                    ln[at:at + n] = po.dequantize_vec(sf[b], mant[at:at + n], p.nScaleBits, alloc[b])
                at += n
            ln = pv.sbr_reconstruct(ln, p)
            assert np.abs(lines[i] - ln).max() <= 1e-12 * max(np.abs(ln).max(), 1e-300), i
        else:
            assert np.array_equal(routed[i], plain[i])
    assert n_sbr >= 8
    assert int(extra["status"].max().item()) == 0
    # 96 kHz: the band table stops at 24 kHz, the cut lies in the lower half
    cp96 = A.audiofile.CodingParams()
    cp96.sampleRate, cp96.nChannels, cp96.nMDCTLines, cp96.nSamplesPerBlock = 96000, 1, 1024, 1024
    cp96.nScaleBits, cp96.nMantSizeBits, cp96.targetBitsPerSample = 4, 12, 1.0
    cp96.useSBR, cp96.useVQ = True, False
    cp96.sfBands = A.psychoac.ScaleFactorBands(A.psychoac.AssignMDCTLinesFromFreqLimits(1024, 96000))
    cp96.sfBandsShort = A.psychoac.ScaleFactorBands(A.psychoac.AssignMDCTLinesFromFreqLimits(128, 96000))
    cp96.omittedBands = A.pacfile.omitted_bands(cp96.sfBands)
    nb = cp96.sfBands.nBands
    with pytest.raises(IndexError):
        A.codec.Decode_SBR(np.zeros(nb, np.int32), np.full(nb, 4, np.int32), np.ones(1024, np.int32), 0, None, cp96)


# ------------------------------------------------------------------ nMDCTLines other than 1024 / 128 (VERDICT r2 missing #6)
def _params(A, sr, half, kbps=128):
    cp = A.audiofile.CodingParams()
    cp.sampleRate, cp.nChannels, cp.nMDCTLines, cp.nSamplesPerBlock = sr, 1, half, half
    cp.nScaleBits, cp.nMantSizeBits, cp.targetBitsPerSample = 4, 12, kbps / (sr / 1000)
    cp.useSBR, cp.useVQ = False, False
    cp.sfBands = A.psychoac.ScaleFactorBands(A.psychoac.AssignMDCTLinesFromFreqLimits(half, sr))
    cp.sfBandsShort = A.psychoac.ScaleFactorBands(A.psychoac.AssignMDCTLinesFromFreqLimits(128, sr))
    cp.omittedBands = []
    return cp


def _blocks(n, count, seed):
    """blocks of n samples: stretches of the harpsichord and castanet excerpts (as fractions) and synthetic ones"""
    from oracle import pac_oracle as po
    rng = np.random.default_rng(seed)
    out = []
    for name in ("harpsichord", "castanet"):
        pcm = np.load(os.path.join(GOLDEN, f"excerpt_{name}.npz"))["pcm"]
        for k in range(count // 3):
            at = int(rng.integers(2000, len(pcm) - n - 1))
            out.append(po.pcm16_to_fraction(pcm[at:at + n, k & 1]))
    t = np.arange(n)
    while len(out) < count:
        x = sum(rng.uniform(0.01, 0.3) * np.cos(2 * np.pi * rng.uniform(50, 15000) * t / 48000 + rng.uniform(0, 6)) for _ in range(5))
        out.append(x + 10.0 ** rng.uniform(-4, -1.5) * rng.standard_normal(n))
    return out


@pytest.mark.parametrize("sr,half", [(48000, 512), (44100, 512), (48000, 256), (32000, 2048)])
def test_calc_smrs_any_block_length(A, sr, half):
    """psychoac.CalcSMRs / getMaskedThreshold through k_smr_generic (pacx_smr_generic_batch) against the oracle's
    calc_smrs: the tuned kernels' tolerance, 1e-9 dB ('and 512 cheaply')"""
    from oracle import pac_oracle as po
    cp = _params(A, sr, half)
    bands = po.band_table(half, sr)
    worst = worst_t = 0.0
    for x in _blocks(2 * half, 9, 50 + half):
        lines = po.mdct_forward(po.sine_window(2 * half) * x, half, half)[:half]
        ov = po.scale_factor(np.max(np.abs(lines)), 4)
        want = po.calc_smrs(x, lines * (1 << ov), ov, sr, bands)
        got = A.psychoac.CalcSMRs(x, lines * (1 << ov), ov, sr, cp.sfBands)
        worst = max(worst, float(np.abs(got - want).max()))
        thr = A.psychoac.getMaskedThreshold(x, lines * (1 << ov), ov, sr, cp.sfBands)
        worst_t = max(worst_t, float(np.abs(thr - po.masked_threshold(x, half, sr)).max()))
    assert worst <= 1e-9 and worst_t <= 1e-9, (worst, worst_t)


def test_generic_smr_kernel_agrees_with_the_tuned_ones(A):
    """the same blocks through k_smr_generic and through k_side_long + k_mask<1024> (and the 256-sample pair):
    SMRs within 1e-9 dB of one another, peak counts equal"""
    import torch
    enc = A.context.encoder(48000, 128 / 48.0)
    for n, short in ((2048, False), (256, True)):
        half = n // 2
        bands = enc.sfBandsShort if short else enc.sfBands
        t = A.psychoac._generic_tables(n, 48000, bands, enc.device)
        xs = np.stack(_blocks(n, 6, 70 + n))
        from oracle import pac_oracle as po
        lines = np.stack([po.mdct_forward(po.sine_window(n) * x, half, half)[:half] for x in xs])
        g_smr, g_npk = enc.smr_generic(torch.as_tensor(xs, device=enc.device), torch.as_tensor(lines, device=enc.device), t,
                                       want_peaks=True)
        for i, x in enumerate(xs):
            want = A.psychoac.CalcSMRs(x, lines[i], 0, 48000, bands)
            assert np.abs(g_smr[i].cpu().numpy() - want).max() <= 1e-9, (n, i)
        assert int(g_npk.min().item()) > 0


@pytest.mark.parametrize("sr,half,kbps", [(48000, 512, 128), (44100, 512, 96), (48000, 256, 192)])
def test_encode_single_channel_with_512_lines(A, sr, half, kbps):
    """codec.EncodeSingleChannel with nMDCTLines = 512 (and 256): composed from the GPU-backed module mirrors, the codes of the
    oracle's encode_channel -- overall scale, allocation, scale factors, mantissas -- exactly; all four window kinds"""
    from oracle import pac_oracle as po
    cp = _params(A, sr, half, kbps)
    p = po.make_params(sr, 1, kbps, half)
    assert list(p.sfBands.nLines) == list(cp.sfBands.nLines)
    n_mant = 0
    for i, x in enumerate(_blocks(2 * half, 12, 90 + half + kbps)):
        fl = ((False, False, False), (True, False, False), (False, False, True), (True, False, True))[i % 4]
        sf, ba, mant, ov = A.codec.EncodeSingleChannel(x, cp, *fl)
        w_sf, w_ba, w_mant, w_ov = po.encode_channel(x.copy(), p, *fl)
        assert ov == w_ov and list(ba) == list(w_ba) and list(sf) == list(w_sf), (i, fl)
        assert np.array_equal(mant, w_mant), (i, fl)
        n_mant += len(mant)
        # and back: codec.Decode wants the mantissas line-indexed, as PACFile.getDecodedBlock builds them
        by_line = np.zeros(half, np.int32)
        at = 0
        for b in range(cp.sfBands.nBands):
            if ba[b]:
                lo = int(cp.sfBands.lowerLine[b])
                by_line[lo:lo + cp.sfBands.nLines[b]] = mant[at:at + cp.sfBands.nLines[b]]
                at += cp.sfBands.nLines[b]
        got = A.codec.Decode(sf, ba, by_line, ov, None, cp, *fl)
        want = po.decode_block(p, w_sf, w_ba, by_line, w_ov, *fl)
        assert np.abs(got - want).max() <= 1e-12 * max(np.abs(want).max(), 1e-300), (i, fl)
    assert n_mant > 1000


L512 = np.load(os.path.join(GOLDEN, "lines512.npz"))


@pytest.mark.parametrize("tag", [str(c) for c in L512["cases"]])
def test_streams_with_512_lines_against_the_reference(A, tag):
    """whole streams with nMDCTLines = 512 through the PACFile mirror (function-level path: blocks composed from the
    GPU-backed module mirrors, a short-coded hop is four 128-line sub-blocks on the tuned kernels): the bytes the
    REFERENCE wrote with cp.nMDCTLines = 512, and its decoder's PCM"""
    name, kbps, kind = tag.rsplit("_", 2)
    pac = A.pacfile.encode_stream(L512[f"pcm_{tag}"], int(L512[f"sr_{tag}"]), int(kbps), block_switching=(kind == "bs"),
                                  n_lines=512)
    want = bytes(L512[f"pac_{tag}"])
    assert len(pac) == len(want) and pac == want
    pcm = A.pacfile.decode_stream(want)
    assert pcm.shape == L512[f"dec_{tag}"].shape and np.array_equal(pcm, L512[f"dec_{tag}"])
