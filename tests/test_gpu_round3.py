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


@pytest.mark.parametrize("kbps", [96, 128])
def test_gain_shape_coder_path_switches(A, monkeypatch, kbps):
    """gain-shape coder (+ SBR at 96 kb/s), block-switched and all-long: PACX_SPLIT_SHORT x
    PACX_VQ_FUSE_ALLOC x (PACX_VQ_FRAME, PACX_VQ_BFS); then the two decoders on the default stream"""
    pcm, sr = _mixed_stream()
    for bs in (True, False):
        _set(monkeypatch, {})
        want = A.pacfile.encode_stream(pcm, sr, kbps, block_switching=bs, use_vq=True, use_sbr=kbps < 128)
        for split, alloc, (frame, bfs) in itertools.product((None, "0"), (None, "0", "1"),
                                                            ((None, None), ("0", "0"), ("0", "1"))):
            _set(monkeypatch, {"PACX_SPLIT_SHORT": split, "PACX_VQ_FUSE_ALLOC": alloc, "PACX_VQ_FRAME": frame,
                               "PACX_VQ_BFS": bfs})
            got = A.pacfile.encode_stream(pcm, sr, kbps, block_switching=bs, use_vq=True, use_sbr=kbps < 128)
            assert got == want, (bs, split, alloc, frame, bfs)
        _set(monkeypatch, {})
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
