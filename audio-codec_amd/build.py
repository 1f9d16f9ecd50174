"""Builds libpacx.so (the HIP library behind include/pacx.h) in-tree for gfx950.

    python audio-codec_amd/build.py [--force]

hipcc cross-compiles without a GPU, so this also runs in the build container.
The .so is git-ignored but travels to the GPU box with the source snapshot.

Staleness is decided by CONTENT, not by mtimes: every object carries the sha256 of
(compiler flags + its source + every header it may include) in build/<name>.o.hash and the
library the hash of all of those in libpacx.so.hash, so a library that does not belong to
the sources next to it is never loaded silently (`_lib.load()` raises on a mismatch, or rebuilds with PACX_AUTOBUILD=1).

Every compile also records the compiler's per-kernel resource report (VGPRs, spills, scratch,
occupancy, LDS: -Rpass-analysis=kernel-resource-usage) in build/resources.json;
tests/test_build_resources.py gates the hot kernels on "no scratch, no spills" with it.
"""
import fcntl
import hashlib
import json
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libpacx.so")
OBJ = os.path.join(HERE, "build")
RESOURCES = os.path.join(OBJ, "resources.json")
SOURCES = ["pacx_api.hip", "k_mdct.hip", "k_psy.hip", "k_quant.hip", "k_misc.hip", "k_mdct2.hip", "k_decode.hip",
           "k_vq.hip", "k_vq_dec.hip", "k_mdct3.hip"]
# -ffp-contract=off: integer codes must follow the reference's individually
# rounded double operations; FMAs are written explicitly where wanted.
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off",
         "-Wall", "-Wno-unused-function", "-Rpass-analysis=kernel-resource-usage"]


def _hipcc():
    return os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _headers():
    hs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))
    hs.append(os.path.join(HERE, "..", "include", "pacx.h"))
    return hs


def _digest(paths, extra=()):
    h = hashlib.sha256()
    for e in extra:
        h.update(e.encode() + b"\0")
    for p in paths:
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def source_hashes(extra_flags=()):
    """{source name: sha256 of flags + source + headers}"""
    hs = _headers()
    return {s: _digest([os.path.join(CSRC, s)] + hs, FLAGS + list(extra_flags)) for s in SOURCES}


def library_hash():
    """hash a current libpacx.so must carry in libpacx.so.hash"""
    sh = source_hashes()
    return hashlib.sha256("".join(sh[s] for s in SOURCES).encode()).hexdigest()


def is_current():
    return os.path.exists(OUT) and _read(OUT + ".hash") == library_hash()


_REMARK = re.compile(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|"
                     r"Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]):\s+(\S+)")
_KEYS = {"TotalSGPRs": "sgprs", "VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch",
         "Occupancy [waves/SIMD]": "occupancy", "SGPRs Spill": "sgpr_spill", "VGPRs Spill": "vgpr_spill",
         "LDS Size [bytes/block]": "lds"}


def parse_resource_remarks(text):
    """compiler remarks -> {mangled kernel name: {vgprs, scratch, ...}}"""
    out, cur = {}, None
    for m in _REMARK.finditer(text):
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            cur = out.setdefault(v, {})
        elif cur is not None:
            cur[_KEYS[k]] = int(v)
    return out


def _demangle(names):
    if not names:
        return {}
    try:
        res = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
        return dict(zip(names, res))
    except (OSError, subprocess.CalledProcessError):
        return {n: n for n in names}


def _compile(src_name, obj, want_hash, extra_flags=(), verbose=True):
    cmd = [_hipcc()] + FLAGS + list(extra_flags) + ["-c", os.path.join(CSRC, src_name), "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    p = subprocess.run(cmd, capture_output=True, text=True)
    if p.returncode != 0:
        lines = p.stderr.split("\n")
        keep = [i for i, l in enumerate(lines) if "error:" in l]
        for i in keep[:20]:
            sys.stderr.write("\n".join(lines[i:i + 4]) + "\n")
        raise subprocess.CalledProcessError(p.returncode, cmd)
    # warnings stay visible; the remarks go to the resource report
    for line in p.stderr.split("\n"):
        if "warning:" in line or "error:" in line:
            sys.stderr.write(line + "\n")
    res = parse_resource_remarks(p.stderr)
    with open(obj + ".res.json", "w") as f:
        json.dump(res, f)
    with open(obj + ".hash", "w") as f:
        f.write(want_hash)


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    with open(os.path.join(OBJ, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)        # ranks of one node may all get here at once
        want = source_hashes()
        jobs = []
        for s in SOURCES:
            obj = os.path.join(OBJ, s.replace(".hip", ".o"))
            if force or not os.path.exists(obj) or _read(obj + ".hash") != want[s] \
                    or not os.path.exists(obj + ".res.json"):
                jobs.append((s, obj, want[s]))
        if jobs:
            with ThreadPoolExecutor(max_workers=4) as ex:
                list(ex.map(lambda j: _compile(*j, verbose=verbose), jobs))
        objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES]
        lib_hash = library_hash()
        if jobs or force or not os.path.exists(OUT) or _read(OUT + ".hash") != lib_hash:
            cmd = [_hipcc(), "-shared", "--offload-arch=gfx950", "-o", OUT] + objs
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            with open(OUT + ".hash", "w") as f:
                f.write(lib_hash)
        if jobs or not os.path.exists(RESOURCES):
            merged = {}
            for s in SOURCES:
                with open(os.path.join(OBJ, s.replace(".hip", ".o")) + ".res.json") as f:
                    res = json.load(f)
                names = _demangle(list(res))
                for k, v in res.items():
                    v["source"] = s
                    merged[names[k]] = v
            with open(RESOURCES, "w") as f:
                json.dump(merged, f, indent=1, sort_keys=True)
    return OUT


def ensure(verbose=False):
    """The library that belongs to the sources in this tree: rebuilt when its recorded content
    hash says it is not (hipcc is on the build container and on the GPU boxes alike)."""
    if is_current():
        return OUT
    return build(verbose=verbose)


def resources():
    """{demangled kernel name: {vgprs, agprs, scratch, vgpr_spill, sgpr_spill, occupancy, lds, source}}
    of the current build."""
    build(verbose=False)
    with open(RESOURCES) as f:
        return json.load(f)


def device_asm(source):
    """gfx950 assembly text of one csrc/*.hip (device side only, the library's flags), cached
    under build/ by content hash: what tests/test_build_isa.py checks the hand-counted
    s_waitcnt / LDS-DMA orderings on."""
    os.makedirs(OBJ, exist_ok=True)
    want = source_hashes()[source]
    out = os.path.join(OBJ, source.replace(".hip", ".s"))
    if not os.path.exists(out) or _read(out + ".hash") != want:
        flags = [f for f in FLAGS if not f.startswith("-Rpass")]
        subprocess.check_call([_hipcc()] + flags + ["--offload-device-only", "-S", os.path.join(CSRC, source), "-o", out],
                              stderr=subprocess.DEVNULL)
        with open(out + ".hash", "w") as f:
            f.write(want)
    with open(out) as f:
        return f.read()


def build_phase_debug():
    """libpacx_dbg.so: the same library with k_mdct3.hip and k_psy.hip compiled with
    -DPACX_MDCT_DEBUG / -DPACX_PSY_DEBUG / -DPACX_TAIL_DEBUG (in-kernel s_memtime stamps per phase, read by
    tools/mdct_phase_probe.py and tools/psy_phase_probe.py through PACX_LIB).  A
    measuring aid, never loaded by default."""
    build(verbose=False)
    dbg_srcs = {"k_mdct3.hip": "-DPACX_MDCT_DEBUG", "k_psy.hip": "-DPACX_PSY_DEBUG", "k_quant.hip": "-DPACX_TAIL_DEBUG", "k_vq.hip": "-DPACX_VQ_DEBUG", "k_vq_dec.hip": "-DPACX_VQD_DEBUG"}
    dbg_objs = []
    for src, flag in dbg_srcs.items():
        obj = os.path.join(OBJ, src.replace(".hip", "_dbg.o"))
        subprocess.check_call([_hipcc()] + FLAGS[:-1] + [flag, "-c", os.path.join(CSRC, src), "-o", obj])
        dbg_objs.append(obj)
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES if s not in dbg_srcs]
    out = os.path.join(HERE, "libpacx_dbg.so")
    subprocess.check_call([_hipcc(), "-shared", "--offload-arch=gfx950", "-o", out] + objs + dbg_objs)
    for obj in dbg_objs:
        os.remove(obj)
    with open(out + ".hash", "w") as f:          # the sources it belongs to (_lib.load checks PACX_LIB libraries too)
        f.write(library_hash())
    return out


def build_variant(name, defines):
    """variants/libpacx_<name>.so: the library with extra -D flags on some sources, for A/B
    measurements through PACX_LIB (never loaded by default).  defines: {source: [flags]}."""
    build(verbose=False)
    out_dir = os.path.join(HERE, "variants")
    os.makedirs(out_dir, exist_ok=True)
    objs = []
    for s in SOURCES:
        if s in defines:
            obj = os.path.join(out_dir, s.replace(".hip", f"_{name}.o"))
            flags = [f for f in FLAGS if not f.startswith("-Rpass")]
            subprocess.check_call([_hipcc()] + flags + list(defines[s]) + ["-c", os.path.join(CSRC, s), "-o", obj],
                                  stderr=subprocess.DEVNULL)
            objs.append(obj)
        else:
            objs.append(os.path.join(OBJ, s.replace(".hip", ".o")))
    out = os.path.join(out_dir, f"libpacx_{name}.so")
    subprocess.check_call([_hipcc(), "-shared", "--offload-arch=gfx950", "-o", out] + objs)
    with open(out + ".hash", "w") as f:
        f.write(library_hash())
    return out


def build_tools():
    """tools/*.hip -> variants/<name> (stand-alone HIP programs, measuring aids)."""
    out_dir = os.path.join(HERE, "variants")
    os.makedirs(out_dir, exist_ok=True)
    tools_dir = os.path.join(HERE, "..", "tools")
    outs = []
    for f in sorted(os.listdir(tools_dir)):
        if f.endswith(".hip"):
            out = os.path.join(out_dir, f[:-4])
            subprocess.check_call([_hipcc(), "-O3", "--offload-arch=gfx950", "-Wno-unused-result",
                                   os.path.join(tools_dir, f), "-o", out])
            outs.append(out)
    return outs


if __name__ == "__main__":
    if "--variant" in sys.argv:            # --variant name source.hip -DX=1 [source2.hip -DY ...]
        i = sys.argv.index("--variant")
        name, rest, defs, cur = sys.argv[i + 1], sys.argv[i + 2:], {}, None
        for a in rest:
            if a.endswith(".hip"):
                cur = defs.setdefault(a, [])
            else:
                cur.append(a)
        print(build_variant(name, defs))
    elif "--tools" in sys.argv:
        print(build_tools())
    elif "--phase-debug" in sys.argv:
        print(build_phase_debug())
    elif "--resources" in sys.argv:
        for name, r in sorted(resources().items()):
            print(f"{r['vgprs']:4d} v {r['agprs']:3d} a  scratch {r['scratch']:4d}  vspill {r['vgpr_spill']:3d}  "
                  f"occ {r['occupancy']}  lds {r['lds']:6d}  {name[:110]}")
    else:
        print(build(force="--force" in sys.argv))
