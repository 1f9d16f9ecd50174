"""The C ABI from plain C (examples/cabi_demo.c): compiled with gcc against
include/pacx.h, linked to libpacx.so and the system HIP runtime, run as its own
process -- no Python and no PyTorch on the path.  The C host passes NO tables
(NULL pointers -> the library's built-in NumPy-evaluated copies) and takes its band
layout from pacx_default_bands, and its .pac body must equal the oracle's byte for byte."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_excerpt
from oracle import pac_oracle as po

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import audio_codec_amd as A
    A.load()                                  # makes sure libpacx.so belongs to this tree
    lib_dir = os.path.join(ROOT, "audio-codec_amd")
    out = str(tmp_path_factory.mktemp("cabi") / "cabi_demo")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
                           "-D__HIP_PLATFORM_AMD__", os.path.join(ROOT, "examples", "cabi_demo.c"),
                           "-L", lib_dir, "-lpacx", "-L", "/opt/rocm/lib", "-lamdhip64", "-lm",
                           "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-o", out])
    return out


def test_c_host_encodes_a_stream(exe):
    out = subprocess.run([exe, "48"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "96 channel-blocks" in out.stdout and "consistent" in out.stdout and "tables exact" in out.stdout


def _planar_stream(pcm, hop=1024):
    """what the reference's driver feeds the coder (coder/pacfile.py:716-743, 612-625): a prior
    block of zeros, the hops, the last hop a second time, the Close block of zeros"""
    n, n_ch = pcm.shape
    buf = np.zeros((n_ch, n + 3 * hop), dtype=np.int16)
    buf[:, hop:hop + n] = pcm.T
    buf[:, hop + n:2 * hop + n] = pcm[n - hop:].T
    return buf


@pytest.mark.parametrize("source", ["synthetic48", "harpsichord", "spmg"])
def test_c_host_bytes_equal_the_oracle(exe, tmp_path, source):
    import audio_codec_amd as A
    if source == "synthetic48":
        sr, pcm = 48000, A.synth.stream(24, 2)
        want = po.encode_stream(pcm, sr, 128, block_switching=False)
    else:
        # 64-hop excerpt of one of the reference's test WAVs: the bytes the REFERENCE wrote
        ex = load_excerpt(source)
        sr, pcm = int(ex["sr"]), ex["pcm"]
        if len(pcm) % 1024:
            pcm = np.concatenate((pcm, np.zeros((-len(pcm) % 1024, pcm.shape[1]), pcm.dtype)))
        want = bytes(ex["pac_long"])
    n_bands = int.from_bytes(want[26:30], "little")
    want_body = want[30 + 2 * n_bands:]                 # 'PAC ' + '<LHLLHHHH' + '<L nBands' + nBands x '<H'
    buf = _planar_stream(pcm)
    raw, body = str(tmp_path / "in.raw"), str(tmp_path / "body.bin")
    buf.tofile(raw)
    out = subprocess.run([exe, "--pcm", raw, str(buf.shape[0]), str(buf.shape[1]), "--out", body,
                          "--rate", str(sr), "--kbps", "128"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "tables exact" in out.stdout
    got = open(body, "rb").read()
    assert got == want_body, (len(got), len(want_body))
