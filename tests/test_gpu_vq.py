"""GPU parity of the gain-shape (PVQ) / SBR encode path -- BASELINE config 4,
the reference's shipped configuration (useVQ, useSBR below 128 kb/s, block
switching) -- against the reference's own outputs (tests/golden, made by
make_golden.py --vq) and the oracle.  Everything goes through the C ABI
(pacx_encode_vq_batch)."""
import hashlib
import json
import os
import struct
import zlib

import numpy as np
import pytest

from conftest import EXCERPTS, GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def A():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import audio_codec_amd as a
    return a


def split_blocks(pac):
    """-> (header bytes, [channel-block payload bytes])"""
    pos = 4 + struct.calcsize("<LHLLHHHH")
    nb = struct.unpack("<L", pac[pos:pos + 4])[0]
    pos += 4 + 2 * nb
    head, blocks = pac[:pos], []
    while pos < len(pac):
        n = struct.unpack("<L", pac[pos:pos + 4])[0]
        blocks.append(pac[pos + 4:pos + 4 + n])
        pos += 4 + n
    return head, blocks


def describe_diff(got, want, n_bands_long=17):
    """First differing channel-block, for a readable failure."""
    hg, bg = split_blocks(got)
    hw, bw = split_blocks(want)
    if hg != hw:
        return "headers differ"
    if len(bg) != len(bw):
        return f"{len(bg)} blocks, expected {len(bw)}"
    bad = [i for i in range(len(bg)) if bg[i] != bw[i]]
    i = bad[0]
    g, w = bg[i], bw[i]
    if len(g) != len(w):
        return f"{len(bad)} blocks differ; block {i}: {len(g)} bytes, expected {len(w)}"
    bit = next(8 * k + j for k in range(len(g)) for j in range(8)
               if ((g[k] ^ w[k]) >> (7 - j)) & 1)
    return (f"{len(bad)} of {len(bg)} blocks differ (first {bad[:8]}); block {i} flags {w[0] >> 5}: "
            f"first differing bit {bit} of {8 * len(w)}")


@pytest.mark.parametrize("name", EXCERPTS)
@pytest.mark.parametrize("kbps", [128, 96])
def test_excerpt_pac_bytes(A, name, kbps):
    ex = np.load(os.path.join(GOLDEN, f"excerpt_{name}.npz"))
    gold = np.load(os.path.join(GOLDEN, f"excerpt_vq_{name}.npz"))
    hops = int(gold["hops"])
    want = bytes(gold[f"pac_vq{kbps}"])
    got = A.pacfile.encode_stream(ex["pcm"][:hops * 1024], int(ex["sr"]), kbps, block_switching=True,
                                  use_vq=True, use_sbr=kbps < 128)
    assert got == want, describe_diff(got, want)


def test_synthetic_stream_vs_oracle(A):
    from oracle import pac_oracle_vq as pv
    pcm = A.synth.stream(12, 2, 48000, seed=7)
    for kbps in (128, 96):
        want = pv.encode_stream_vq(pcm, 48000, kbps)
        got = A.pacfile.encode_stream(pcm, 48000, kbps, block_switching=True, use_vq=True,
                                      use_sbr=kbps < 128)
        assert got == want, describe_diff(got, want)


def test_signal_zoo_vs_oracle(A):
    """The corner-case signals of test_gpu_parity._signal_zoo (silence, DC, clipping square
    wave, impulses, a few-LSB tone, noise, an attack, Nyquist; dithered where the exact
    values would be zeros) through the gain-shape + SBR coder, byte for byte."""
    import importlib.util
    from oracle import pac_oracle_vq as pv
    spec = importlib.util.spec_from_file_location(
        "tp", os.path.join(os.path.dirname(os.path.abspath(__file__)), "test_gpu_parity.py"))
    tp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tp)
    for name, pcm in tp._signal_zoo(6).items():
        want = pv.encode_stream_vq(pcm, 48000, 96)
        got = A.pacfile.encode_stream(pcm, 48000, 96, block_switching=True, use_vq=True, use_sbr=True)
        assert got == want, name + ": " + describe_diff(got, want)


def test_status_and_final_alloc(A):
    """No band of real material reaches a case the reference cannot code, every
    coded band fills its slot exactly, and silent bands end with allocation 0."""
    import torch
    ex = np.load(os.path.join(GOLDEN, "excerpt_castanet.npz"))
    pcm = np.ascontiguousarray(ex["pcm"][:16 * 1024])
    pcm[4 * 1024:6 * 1024] = 0                       # two silent hops
    enc = A.engine.Encoder(int(ex["sr"]), 128 / (int(ex["sr"]) / 1000), use_vq=True)
    planar = A.pacfile.device_stream(enc, pcm)
    _, flags = enc.transient_flags(planar, 16)
    out = enc.encode_vq(A.engine.PcmView.stream(planar), flags)
    st = out["status"].cpu().numpy()
    assert not np.any(st & A._lib.ST_VQ_UNDEFINED)
    ba = out["bit_alloc"].cpu().numpy()
    fl = flags.cpu().numpy()
    silent = [f for f in range(len(fl)) if not (fl[f] & 2) and f in (5,)]   # frame 5 = hops 4,5 -> all zero
    for f in silent:
        assert not ba[2 * f].any() and not ba[2 * f + 1].any()


@pytest.mark.parametrize("key", ["castanet:128", "castanet:96", "harpsichord:128", "harpsichord:96",
                                 "quar48_1:128", "quar48_1:96", "spmg:128", "spmg:96"])
def test_whole_file(A, key):
    """Whole test WAVs in the shipped configuration: sha256 of the reference's
    output (for 6 of 8 that is also the .pac file the reference committed);
    quar48_1 holds four rounding-noise-decided short blocks (DC sub-blocks,
    DESIGN.md) which are compared by size only."""
    name, kbps = key.split(":")
    rec = json.load(open(os.path.join(GOLDEN, "vqfile.json")))[key]
    crcs = np.load(os.path.join(GOLDEN, "vqfile.npz"))[f"{name}_{kbps}"]
    full = np.load(os.path.join(GOLDEN, f"full_{name}.npz"))
    got = A.pacfile.encode_stream(full["pcm"], int(full["sr"]), int(kbps), block_switching=True,
                                  header_samples=int(full["declared"]), use_vq=True,
                                  use_sbr=int(kbps) < 128)
    assert len(got) == rec["size"]
    if hashlib.sha256(got).hexdigest() == rec["sha256"]:
        return
    _, blocks = split_blocks(got)
    assert len(blocks) == len(crcs)
    bad = [i for i, b in enumerate(blocks)
           if zlib.crc32(struct.pack("<L", len(b)) + b) != int(crcs[i])]
    allowed = set(rec["blocks_differing_from_committed"])
    assert set(bad) <= allowed, f"{len(bad)} blocks differ: {bad[:12]}"


# ------------------------------------------------------------------ decode side
@pytest.mark.parametrize("name", EXCERPTS)
@pytest.mark.parametrize("kbps", [128, 96])
def test_decode_excerpt_vs_reference_decoder(A, name, kbps):
    """pacx_decode_vq_batch on the reference's .pac bytes against the PCM the
    reference's own decoder produced from them (bit-exact int16)."""
    gold = np.load(os.path.join(GOLDEN, f"excerpt_vq_{name}.npz"))
    want = np.load(os.path.join(GOLDEN, f"decoded_vq_{name}.npz"))[f"pcm_vq{kbps}"]
    got = A.pacfile.decode_stream(bytes(gold[f"pac_vq{kbps}"]))
    assert got.shape == want.shape
    bad = np.nonzero(np.any(got != want, axis=1))[0]
    assert len(bad) == 0, f"{len(bad)} samples differ, first at {bad[:5]}, max |d| " \
                          f"{np.max(np.abs(got.astype(int) - want.astype(int)))}"


def test_decode_lines_vs_oracle(A):
    """MDCT lines out of k_vq_dec / k_sbr_recon against the oracle's, block by block."""
    import struct as st
    import torch
    from oracle import pac_oracle as po, pac_oracle_vq as pv
    gold = np.load(os.path.join(GOLDEN, "excerpt_vq_harpsichord.npz"))
    for kbps in (128, 96):
        data = bytes(gold[f"pac_vq{kbps}"])
        cp, pos = A.pacfile.parse_header(data)
        enc = A.context.encoder_for_params(cp)
        offs, sizes = [], []
        while pos < len(data):
            n = int.from_bytes(data[pos:pos + 4], "little")
            offs.append(pos + 4)
            sizes.append(n)
            pos += 4 + n
        body = torch.frombuffer(bytearray(data) + bytearray(8), dtype=torch.uint8).to(enc.device)
        out = enc.decode_vq(body, torch.tensor(sizes, dtype=torch.int32, device=enc.device), cp.nChannels,
                            offsets=torch.tensor(offs, dtype=torch.int64, device=enc.device),
                            want_lines=True, want_pcm=False)
        assert not np.any(out["status"].cpu().numpy() & A._lib.ST_VQ_UNDEFINED)
        lines = out["lines"].cpu().numpy()
        fl = out["flags"].cpu().numpy()
        p = po.make_params(cp.sampleRate, cp.nChannels, 128)
        p.useVQ, p.useSBR = True, bool(cp.useSBR)
        p.omittedBands = list(po.omitted_bands(p.sfBands)) if p.useSBR else []
        checked = 0
        for i in range(0, len(offs), 3):
            if fl[i] & 2:
                continue
            br = po.BitReader(data[offs[i]:offs[i] + sizes[i]] + b"\0" * 8)
            br.get(3)
            overall = br.get(4)
            alloc = [a + 1 if a else 0 for a in (br.get(12) for _ in range(p.sfBands.nBands))]
            sbr = bool(p.useSBR and np.any(np.array(alloc)[np.array(p.omittedBands, dtype=int)] != 0))
            want = pv.decode_lines_vq(br, p, alloc, False, sbr)
            if sbr:
                want = pv.sbr_reconstruct(want, p)
            scale = max(np.max(np.abs(want)), 1e-300)
            assert np.max(np.abs(lines[i] - want)) <= 1e-12 * scale, (kbps, i)
            checked += 1
        assert checked >= 10


@pytest.mark.parametrize("key", ["castanet:128", "castanet:96", "harpsichord:128", "harpsichord:96",
                                 "spmg:128", "spmg:96"])
def test_whole_file_round_trip_matches_committed_wav(A, key):
    """GPU encode -> GPU decode of a whole test WAV in the shipped configuration
    gives the PCM of the decoded WAV the reference's author committed
    (test_decoded_full/<wav>_<rate>.wav)."""
    name, kbps = key.split(":")
    rec = json.load(open(os.path.join(GOLDEN, "vqwav.json")))[key]
    full = np.load(os.path.join(GOLDEN, f"full_{name}.npz"))
    pac = A.pacfile.encode_stream(full["pcm"], int(full["sr"]), int(kbps), block_switching=True,
                                  header_samples=int(full["declared"]), use_vq=True, use_sbr=int(kbps) < 128)
    pcm = A.pacfile.decode_stream(pac)[:rec["n_samples"]]
    assert hashlib.sha256(np.ascontiguousarray(pcm).astype("<i2").tobytes()).hexdigest() == rec["pcm_sha256"]


# ----------------------------------------------------------- reference-style API
def test_codec_encode_vq_mirror_vs_oracle(A):
    """codec.Encode / Encode_SBR with useVQ: (bitAlloc, indices, idx_bits,
    overallScale) lists as the reference returns them."""
    from oracle import pac_oracle as po, pac_oracle_vq as pv
    ex = np.load(os.path.join(GOLDEN, "excerpt_harpsichord.npz"))
    sr = int(ex["sr"])
    pcm = ex["pcm"][:3 * 1024]
    data = [po.pcm16_to_fraction(pcm[1024:3 * 1024, ch]) for ch in range(2)]
    for kbps, fn_name in ((128, "Encode"), (96, "Encode_SBR")):
        p = pv.make_params_vq(sr, 2, kbps)
        cp = A.audiofile.CodingParams()
        cp.sampleRate, cp.nChannels, cp.nMDCTLines = sr, 2, 1024
        cp.nScaleBits, cp.nMantSizeBits = 4, 12
        cp.targetBitsPerSample = kbps / (sr / 1000)
        cp.useVQ, cp.useSBR = True, kbps < 128
        ba, idx, bits, ov = getattr(A.codec, fn_name)(data, cp)
        for ch in range(2):
            want = (pv.encode_channel_sbr_vq if p.useSBR else pv.encode_channel_vq)(data[ch].copy(), p)
            assert np.array_equal(ba[ch], want[0])
            assert idx[ch] == [[int(v) for v in band] for band in want[1]]
            assert bits[ch] == [[int(v) for v in band] for band in want[2]]
            assert ov[ch] == want[3]


def test_pacfile_block_api_vq_round_trip(A, tmp_path):
    """PACFile.WriteDataBlock / ReadDataBlock with useVQ + useSBR, block by block,
    against the batched stream encode and the reference decoder's PCM."""
    ex = np.load(os.path.join(GOLDEN, "excerpt_spmg.npz"))
    gold = np.load(os.path.join(GOLDEN, "excerpt_vq_spmg.npz"))
    sr, hops = int(ex["sr"]), int(gold["hops"])
    pcm = ex["pcm"][:hops * 1024]
    flags = gold["flags_vq96"]
    path = str(tmp_path / "out.pac")
    cp = A.audiofile.CodingParams()
    cp.sampleRate, cp.nChannels, cp.numSamples = sr, 2, len(pcm)
    cp.nMDCTLines = cp.nSamplesPerBlock = 1024
    cp.nScaleBits, cp.nMantSizeBits = 4, 12
    cp.targetBitsPerSample = 96 / (sr / 1000)
    cp.useVQ, cp.useSBR = True, True
    f = A.pacfile.PACFile(path)
    f.OpenForWriting(cp)
    for h in range(hops + 1):                     # the driver writes the last hop twice
        hh = min(h, hops - 1)
        data = [A.pcmfile.codes_to_fraction(pcm[hh * 1024:(hh + 1) * 1024, ch]) for ch in range(2)]
        f.WriteDataBlock(data, cp, *[bool(v) for v in flags[h]])
    f.Close(cp)
    got = open(path, "rb").read()
    assert got == bytes(gold["pac_vq96"]), describe_diff(got, bytes(gold["pac_vq96"]))
    want_pcm = np.load(os.path.join(GOLDEN, "decoded_vq_spmg.npz"))["pcm_vq96"]
    g = A.pacfile.PACFile(path)
    cp2 = g.OpenForReading()
    out = []
    while True:
        d = g.ReadDataBlock(cp2)
        if not d:
            break
        out.append(np.stack([A.pcmfile.fraction_to_codes(c) for c in d], axis=1))
    g.Close(cp2)
    assert np.array_equal(np.concatenate(out), want_pcm)


@pytest.mark.parametrize("kbps", [48, 64, 192, 256, 384])
def test_other_bit_rates_vs_oracle(A, kbps):
    """Rates far from the shipped 96/128 kb/s reach the rare branches of the
    gain-shape coder: allocations of up to 16 bits per line (split angles beyond
    the 12-bit table, leaves of 3-4 components with hundreds of pulses, deep
    trees), or very few bits (K = 0 leaves, one-bit indices).  Encode bytes and
    decoded PCM against the oracle on a short stream with a transient."""
    from oracle import pac_oracle_vq as pv
    pcm = A.synth.stream(6, 2, 48000, seed=11)
    pcm[3 * 1024 + 200:3 * 1024 + 260] = 30000          # a click: block switching kicks in
    want = pv.encode_stream_vq(pcm, 48000, kbps)
    got = A.pacfile.encode_stream(pcm, 48000, kbps, block_switching=True, use_vq=True, use_sbr=kbps < 128)
    assert got == want, describe_diff(got, want)
    assert np.array_equal(A.pacfile.decode_stream(got), pv.decode_stream_vq(want))


@pytest.mark.parametrize("sr,kbps", [(44100, 96), (32000, 64), (44100, 128)])
def test_other_sample_rates_vs_oracle(A, sr, kbps):
    """The band layout (and with it the SBR cut, the PVQ dimensions and the table sizes)
    follows the sample rate: encode bytes and decoded PCM against the oracle at rates other
    than 48 kHz, on a short stream with a click."""
    from oracle import pac_oracle_vq as pv
    pcm = A.synth.stream(6, 2, sr, seed=5)
    pcm[2 * 1024 + 100:2 * 1024 + 150] = -28000
    want = pv.encode_stream_vq(pcm, sr, kbps)
    got = A.pacfile.encode_stream(pcm, sr, kbps, block_switching=True, use_vq=True, use_sbr=kbps < 128)
    assert got == want, describe_diff(got, want)
    assert np.array_equal(A.pacfile.decode_stream(got), pv.decode_stream_vq(want))


@pytest.mark.parametrize("n_ch", [1, 3])
def test_other_channel_counts_vs_oracle(A, n_ch):
    """Mono and three-channel streams (the block-switching flags are per hop, over all
    channels; the channel-frames of a hop are coded independently)."""
    from oracle import pac_oracle_vq as pv
    pcm = A.synth.stream(5, n_ch, 48000, seed=3)
    pcm[1024 + 300:1024 + 340, 0] = 25000
    want = pv.encode_stream_vq(pcm, 48000, 96)
    got = A.pacfile.encode_stream(pcm, 48000, 96, block_switching=True, use_vq=True, use_sbr=True)
    assert got == want, describe_diff(got, want)
    assert np.array_equal(A.pacfile.decode_stream(got), pv.decode_stream_vq(want))


@pytest.mark.parametrize("kbps", [96, 128])
def test_rich_synthetic_stream_vs_oracle(A, kbps):
    """48 hops of test_gpu_parity.rich_stream (modulated noise, chords, 60 dB level steps,
    clicks, a silent gap) through the gain-shape coder with block switching, bytes and
    decoded PCM against the oracle."""
    import importlib.util
    from oracle import pac_oracle_vq as pv
    spec = importlib.util.spec_from_file_location(
        "tp", os.path.join(os.path.dirname(os.path.abspath(__file__)), "test_gpu_parity.py"))
    tp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tp)
    pcm = tp.rich_stream(48)
    want = pv.encode_stream_vq(pcm, 48000, kbps)
    got = A.pacfile.encode_stream(pcm, 48000, kbps, block_switching=True, use_vq=True, use_sbr=kbps < 128)
    assert got == want, describe_diff(got, want)
    assert np.array_equal(A.pacfile.decode_stream(got), pv.decode_stream_vq(want))


# ------------------------------------------- codec.Decode / Decode_SBR on gain-shape blocks
@pytest.mark.parametrize("kbps", [128, 96])
def test_codec_decode_on_gain_shape_blocks(A, kbps):
    """codec.Decode with useVQ (coder/codec.py:47-92) and codec.Decode_SBR (:95-222) called the
    way PACFile.getDecodedBlock calls them (coder/pacfile.py:177-229): flags, overall scale and
    allocations already read, `pb` a bit cursor standing at the first coded band.  Blocks of the
    reference's own castanet stream (long, short and -- at 96 kb/s -- SBR blocks) against the
    oracle's decode_block_vq; the cursor must end where the reference's would."""
    from oracle import pac_oracle as po
    from oracle import pac_oracle_vq as pv
    seen = {"long": 0, "short": 0, "sbr": 0}
    for name in ("castanet", "harpsichord"):
        _decode_blocks_of(A, name, kbps, seen)
    assert seen["long"] + seen["sbr"] > 0 and seen["short"] > 0
    if kbps < 128:
        assert seen["sbr"] > 0, seen


def _decode_blocks_of(A, name, kbps, seen):
    from oracle import pac_oracle as po
    from oracle import pac_oracle_vq as pv
    ex = np.load(os.path.join(GOLDEN, f"excerpt_vq_{name}.npz"))
    pac = bytes(ex[f"pac_vq{kbps}"])
    head, blocks = split_blocks(pac)
    cp, _ = A.pacfile.parse_header(head)
    cp.omittedBands = A.pacfile.omitted_bands(cp.sfBands) if cp.useSBR else []
    p = pv.make_params_vq(cp.sampleRate, cp.nChannels, kbps)

    class Cursor:                      # the reference's PackedBits, as far as Decode uses it
        def __init__(self, data):
            self.br = po.BitReader(data)

        def ReadBits(self, n):
            return self.br.get(n)

    for blk in blocks[:36]:
        cur, ref = Cursor(blk), po.BitReader(blk)
        fl = [cur.ReadBits(1) for _ in range(3)]
        assert fl == [ref.get(1) for _ in range(3)]
        last_t, cur_t, next_t = fl
        n_sub = 8 if cur_t else 1
        bands = cp.sfBandsShort if cur_t else cp.sfBands
        cp.nMDCTLines = 128 if cur_t else 1024
        p.nMDCTLines = p.nSamplesPerBlock = cp.nMDCTLines
        try:
            for _ in range(n_sub):
                want = pv.decode_block_vq(ref, p, last_t, cur_t, next_t)
                overall = cur.ReadBits(cp.nScaleBits)
                alloc = []
                for b in range(bands.nBands):
                    a = cur.ReadBits(cp.nMantSizeBits)
                    alloc.append(a + 1 if a else 0)
                pf = A.pacfile.PACFile.__new__(A.pacfile.PACFile)
                got = pf.Decode(None, alloc, None, overall, cur, cp, last_t, cur_t, next_t)
                sbr = bool(cp.useSBR and not cur_t and np.any(np.array(alloc)[np.array(cp.omittedBands, dtype=int)] != 0))
                seen["sbr" if sbr else ("short" if cur_t else "long")] += 1
                assert got.shape == want.shape
                assert np.max(np.abs(got - want)) <= 1e-12 * max(np.max(np.abs(want)), 1e-300)
                assert cur.br.pos == ref.pos           # the cursor ends where the reference's would
        finally:
            cp.nMDCTLines = 1024
            p.nMDCTLines = p.nSamplesPerBlock = 1024


def test_level_walk_equals_depth_first_walk(A, monkeypatch):
    """Four coders for the split trees, same bytes: k_vq_frame (all bands of a block level by level, the
    default; trees beyond its node store fall back to k_vq), k_vq_frame2 (PACX_VQ_FRAME=2: in place, one leaf per lane), k_vq walking a band level by level
    (vq_shape_bfs, PACX_VQ_FRAME=0 PACX_VQ_BFS=1) and k_vq depth first (PACX_VQ_BFS=0) -- at bit rates that
    give one-level trees, deep trees and trees beyond either node store."""
    rng = np.random.default_rng(11)
    t = np.arange(20 * 1024)
    tone = 0.4 * np.sin(2 * np.pi * 440 * t / 48000) + 0.2 * np.sin(2 * np.pi * 5200 * t / 48000)
    pcm = np.stack([tone + 0.05 * rng.standard_normal(len(t)), 0.3 * rng.standard_normal(len(t))], axis=1)
    pcm = np.clip(np.round(pcm * 20000), -32767, 32767).astype(np.int16)
    for kbps in (64, 128, 256, 448):
        outs = []
        for frame, bfs in ((None, None), ("2", None), ("0", "0"), ("0", "1"), ("0", None)):
            if frame is None:
                monkeypatch.delenv("PACX_VQ_FRAME", raising=False)
            else:
                monkeypatch.setenv("PACX_VQ_FRAME", frame)
            if bfs is None:
                monkeypatch.delenv("PACX_VQ_BFS", raising=False)
            else:
                monkeypatch.setenv("PACX_VQ_BFS", bfs)
            outs.append(A.pacfile.encode_stream(pcm, 48000, kbps, block_switching=True, use_vq=True,
                                                use_sbr=kbps < 128))
        monkeypatch.delenv("PACX_VQ_FRAME", raising=False)
        assert outs[0] == outs[1] == outs[2] == outs[3] == outs[4], kbps


def test_frame_decoder_equals_band_decoder(A, monkeypatch):
    """k_vq_dec_frame (bands parsed one per lane, one leaf per lane, in-place combine; the default) and k_vq_dec
    (a wave per band, depth first; PACX_VQ_DEC_FRAME=0; also what takes the blocks whose trees do not fit the
    frame decoder's node store) give the same lines and the same PCM, at bit rates on both sides of that limit."""
    rng = np.random.default_rng(12)
    t = np.arange(16 * 1024)
    tone = 0.4 * np.sin(2 * np.pi * 523 * t / 48000) + 0.2 * np.sin(2 * np.pi * 6100 * t / 48000)
    pcm = np.stack([tone + 0.05 * rng.standard_normal(len(t)), 0.3 * rng.standard_normal(len(t))], axis=1)
    pcm = np.clip(np.round(pcm * 20000), -32767, 32767).astype(np.int16)
    for kbps in (64, 128, 320):
        pac = A.pacfile.encode_stream(pcm, 48000, kbps, block_switching=True, use_vq=True, use_sbr=kbps < 128)
        outs = []
        for mode in ("0", None):
            if mode is None:
                monkeypatch.delenv("PACX_VQ_DEC_FRAME", raising=False)
            else:
                monkeypatch.setenv("PACX_VQ_DEC_FRAME", mode)
            outs.append(A.pacfile.decode_stream(pac))
        assert np.array_equal(outs[0], outs[1]), kbps
