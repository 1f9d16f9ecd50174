"""The C ABI from plain C (examples/cabi_demo.c): compiled with gcc against
include/pacx.h, linked to libpacx.so and the system HIP runtime, run as its own
process -- no Python and no PyTorch on the path."""
import os
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_c_host_encodes_a_stream(tmp_path):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    lib_dir = os.path.join(ROOT, "audio-codec_amd")
    assert os.path.exists(os.path.join(lib_dir, "libpacx.so"))
    exe = str(tmp_path / "cabi_demo")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
                           "-D__HIP_PLATFORM_AMD__", os.path.join(ROOT, "examples", "cabi_demo.c"),
                           "-L", lib_dir, "-lpacx", "-L", "/opt/rocm/lib", "-lamdhip64", "-lm",
                           "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    out = subprocess.run([exe, "48"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "96 channel-blocks" in out.stdout and "consistent" in out.stdout
