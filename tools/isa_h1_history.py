#!/usr/bin/env python3
"""Establishes the cause of round 1's MDCT data race from the code as it was, not from reruns:
compiles csrc/k_mdct2.hip as of the commit BEFORE the fix (60f6072^) and as of the fix (60f6072)
to gfx950 assembly and runs hazard check H1 of tests/test_build_isa.py on both -- every LDS-DMA
into a buffer must follow an `s_waitcnt lgkmcnt(0)` that follows the wave's last LDS access.

    python tools/isa_h1_history.py > profiles/r02_lds_dma_h1_before_after.txt      (needs the git history; CPU only)
"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import test_build_isa as T   # noqa: E402

FIX = "60f6072"
HEADERS = ["wave_fft.h", "pacx_dev.h", "pacx_exact.h", "pcm_stage.h"]


def asm_of(rev):
    d = tempfile.mkdtemp(prefix="h1_")
    for f in ["k_mdct2.hip"] + HEADERS:
        src = subprocess.run(["git", "-C", ROOT, "show", f"{rev}:audio-codec_amd/csrc/{f}"], capture_output=True, text=True)
        if src.returncode == 0:
            open(os.path.join(d, f), "w").write(src.stdout)
    out = os.path.join(d, "k.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off",
                           "--offload-device-only", "-S", os.path.join(d, "k_mdct2.hip"), "-o", out],
                          stderr=subprocess.DEVNULL)
    return open(out).read()


for tag, rev in (("before the fix", FIX + "^"), ("with the fix", FIX)):
    print(f"== k_mdct2.hip {tag} ({rev}) ==")
    for name, ins in T.functions(asm_of(rev)).items():
        if "k_mdct_long_v2" not in name:
            continue
        short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
        try:
            groups = T.check_h1(ins, name)
            print(f"  {short}: H1 holds ({len(groups)} LDS-DMA groups)")
        except AssertionError as e:
            msg = str(e).split(": ", 1)[1]
            print(f"  {short}: H1 VIOLATED -- {msg}")
