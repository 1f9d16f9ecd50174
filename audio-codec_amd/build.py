"""Builds libpacx.so (the HIP library behind include/pacx.h) in-tree for gfx950.

    python audio-codec_amd/build.py [--force]

hipcc cross-compiles without a GPU, so this also runs in the build container.
The .so is git-ignored but travels to the GPU box with the source snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libpacx.so")
OBJ = os.path.join(HERE, "build")
SOURCES = ["pacx_api.hip", "k_mdct.hip", "k_psy.hip", "k_quant.hip", "k_misc.hip", "k_mdct2.hip", "k_decode.hip",
           "k_vq.hip", "k_vq_dec.hip", "k_mdct3.hip"]
# -ffp-contract=off: integer codes must follow the reference's individually
# rounded double operations; FMAs are written explicitly where wanted.
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off",
         "-Wall", "-Wno-unused-function"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(HERE, "..", "include", "pacx.h"))
    os.makedirs(OBJ, exist_ok=True)
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s.replace(".hip", ".o"))
        if force or _stale(obj, [src] + headers):
            jobs.append([hipcc] + FLAGS + ["-c", src, "-o", obj])
    if jobs:
        def run(cmd):
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        with ThreadPoolExecutor(max_workers=4) as ex:
            list(ex.map(run, jobs))
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES]
    if jobs or force or _stale(OUT, objs):
        cmd = [hipcc, "-shared", "--offload-arch=gfx950", "-o", OUT] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT


def build_phase_debug():
    """libpacx_dbg.so: the same library with k_mdct3.hip and k_psy.hip compiled with
    -DPACX_MDCT_DEBUG / -DPACX_PSY_DEBUG / -DPACX_TAIL_DEBUG (in-kernel s_memtime stamps per phase, read by
    tools/mdct_phase_probe.py and tools/psy_phase_probe.py through PACX_LIB).  A
    measuring aid, never loaded by default."""
    build(verbose=False)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    dbg_srcs = {"k_mdct3.hip": "-DPACX_MDCT_DEBUG", "k_psy.hip": "-DPACX_PSY_DEBUG", "k_quant.hip": "-DPACX_TAIL_DEBUG", "k_vq.hip": "-DPACX_VQ_DEBUG"}
    dbg_objs = []
    for src, flag in dbg_srcs.items():
        obj = os.path.join(OBJ, src.replace(".hip", "_dbg.o"))
        subprocess.check_call([hipcc] + FLAGS + [flag, "-c", os.path.join(CSRC, src), "-o", obj])
        dbg_objs.append(obj)
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES if s not in dbg_srcs]
    out = os.path.join(HERE, "libpacx_dbg.so")
    subprocess.check_call([hipcc, "-shared", "--offload-arch=gfx950", "-o", out] + objs + dbg_objs)
    for obj in dbg_objs:
        os.remove(obj)
    return out


def build_tools():
    """tools/hbm_mix_probe.hip -> variants/hbm_mix_probe (stand-alone HIP program, a measuring aid)."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    out_dir = os.path.join(HERE, "variants")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, "hbm_mix_probe")
    subprocess.check_call([hipcc, "-O3", "--offload-arch=gfx950", "-Wno-unused-result",
                           os.path.join(HERE, "..", "tools", "hbm_mix_probe.hip"), "-o", out])
    return out


if __name__ == "__main__":
    if "--tools" in sys.argv:
        print(build_tools())
    elif "--phase-debug" in sys.argv:
        print(build_phase_debug())
    else:
        print(build(force="--force" in sys.argv))
