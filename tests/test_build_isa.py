"""Static check of the LDS-DMA orderings the MDCT kernels rely on, on the COMPILED code
(gfx950 assembly of csrc/k_mdct3.hip and k_mdct2.hip, device side only).

Round 1 found a data race here by a failing test and accepted its fix because the failure
stopped recurring.  The hazards are now stated ("LDS-DMA hazard table") and the
instruction stream is checked for each of them at build time:

 H1  LDS-DMA (`global_load_lds_dwordx4`) writes LDS from the vector-memory side and is NOT
     ordered with the issuing wave's own `ds_read`s: every DMA into a buffer must come after an
     `s_waitcnt lgkmcnt(0)` that follows the last LDS read of that buffer.
 H2  k_mdct_long_x2p issues its DMA from inline asm (invisible to the compiler's waitcnt
     pass) and waits for it with a hand-counted `s_waitcnt vmcnt(2*EPI_STORES)`: vector-memory
     operations retire in issue order, so the count is safe only if at least that many
     vector-memory operations are issued AFTER the DMA in every iteration that is followed by
     another one -- the 2 x 8 `global_store_dwordx4` of the two epilogues.
 H3  No LDS read of a landing buffer may be issued between the loop head and that wait.
"""
import importlib
import re

import pytest


@pytest.fixture(scope="module")
def B():
    import audio_codec_amd  # noqa: F401
    return importlib.import_module("audio_codec_amd.build")


def functions(asm):
    """{mangled name: [instruction / label lines]} of every kernel in an assembly file"""
    out, cur = {}, None
    for line in asm.split("\n"):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = out.setdefault(m.group(1), [])
            continue
        if cur is None:
            continue
        t = line.split(";")[0].rstrip() if not line.lstrip().startswith(";;#") else line.strip()
        lm = re.match(r"^(\.LBB\d+_\d+):(.*)$", line)
        if lm:                                   # label, tagged with the compiler's loop annotation
            note = lm.group(2)
            tag = " LOOPHEAD" if "Loop Header" in note and "Depth=1" in note else (" INLOOP" if "Loop" in note else "")
            cur.append(lm.group(1) + ":" + tag)
        elif t.startswith("\t") and t.strip() and not t.strip().startswith("."):
            cur.append(t.strip())
        if t.strip() == "s_endpgm":
            cur = None
    return out


def is_dma(i):
    return i.startswith("global_load_lds_dwordx4")


def is_lds_access(i):
    return i.startswith("ds_") and not i.startswith("ds_nop")


def lgkm0(i):
    return i.startswith("s_waitcnt") and "lgkmcnt(0)" in i


def dma_groups(ins):
    groups, k = [], 0
    while k < len(ins):
        if is_dma(ins[k]):
            j = k
            while j + 1 < len(ins) and (is_dma(ins[j + 1]) or not re.match(r"^(ds_|global_|buffer_|flat_|s_waitcnt)", ins[j + 1])) \
                    and any(is_dma(x) for x in ins[j + 1:j + 12]):
                j += 1
                if not any(is_dma(x) for x in ins[j:j + 12]):
                    break
            while not is_dma(ins[j]):
                j -= 1
            groups.append((k, j))
            k = j + 1
        else:
            k += 1
    return groups


def check_h1(ins, name):
    """walking back from every DMA group: an s_waitcnt lgkmcnt(0) before any LDS access"""
    gs = dma_groups(ins)
    assert gs, f"{name}: no LDS-DMA found"
    for (a, _) in gs:
        k = a - 1
        while k >= 0 and not is_lds_access(ins[k]):
            if lgkm0(ins[k]):
                break
            k -= 1
        assert k < 0 or lgkm0(ins[k]), f"{name}: LDS access `{ins[k]}` before the DMA at #{a} with no s_waitcnt lgkmcnt(0) between"
    return gs


def kernel(fns, pattern):
    hits = [k for k in fns if re.search(pattern, k)]
    assert len(hits) == 1, (pattern, hits)
    return hits[0], fns[hits[0]]


@pytest.mark.parametrize("step", ["Lb0E", "Lb1E"])      # the stand-alone instantiation and the in-step one
def test_x2p_dma_and_counted_wait(B, step):
    src = open(B.CSRC + "/k_mdct3.hip").read()
    m = re.search(r"constexpr int EPI_STORES = (\d+);", src)
    assert m, "k_mdct3.hip must define EPI_STORES (line stores per epilogue)"
    epi = int(m.group(1))
    name, ins = kernel(functions(B.device_asm("k_mdct3.hip")), r"k_mdct_long_x2pILi8ELi2E%sE" % step)
    groups = check_h1(ins, name)                                        # H1
    assert all(b - a == 3 or sum(map(is_dma, ins[a:b + 1])) == 4 for a, b in groups), "4 DMA instructions per 4 KB frame"
    # the counted wait, once, inside the loop
    waits = [k for k, i in enumerate(ins) if re.fullmatch(r"s_waitcnt vmcnt\(%d\)" % (2 * epi), i)]
    assert len(waits) == 1, f"{name}: expected exactly one s_waitcnt vmcnt({2 * epi}), found {len(waits)}"
    w = waits[0]
    head = max(k for k in range(w) if ins[k].endswith("LOOPHEAD"))      # the frame loop's header block
    in_loop = [g for g in groups if g[0] > w]
    assert len(in_loop) == 2 and len(groups) == 4, "two prologue DMA groups (frames A, B) and two in the loop"
    last_dma = in_loop[-1][1]
    # H2: the vector-memory operations issued after the loop's last DMA, up to the end of the loop body
    end = next(k for k in range(last_dma, len(ins)) if re.match(r"^\.LBB\d+_\d+:$", ins[k]))   # first block outside the loop
    younger = [i for i in ins[last_dma + 1:end + 1] if re.match(r"^(global|buffer|flat)_", i)]
    stores16 = [i for i in younger if i.startswith("global_store_dwordx4")]
    assert len(stores16) == 2 * epi, f"{name}: {len(stores16)} line stores after the DMA, the wait counts {2 * epi}"
    assert not any(i.startswith(("global_load", "buffer_load", "flat_")) for i in younger), \
        "no loads between the DMA and the wait (they would be waited for as well, harmless, but unexpected)"
    # H3: between the loop head and the first LDS read, nothing but the two explicit waits touches vmcnt
    first_ds = next(k for k in range(head, len(ins)) if is_lds_access(ins[k]))
    assert w < first_ds, "the counted wait comes before the first landing-buffer read"
    between = [i for i in ins[head:first_ds] if i.startswith("s_waitcnt") and "vmcnt" in i]
    assert between == ["s_waitcnt vmcnt(%d)" % (2 * epi), "s_waitcnt vmcnt(0)"], between
    assert not any(re.match(r"^(global|buffer|flat)_", i) for i in ins[head:first_ds])


@pytest.mark.parametrize("anywin", ["Lb0E", "Lb1E"])
def test_v2_dma_after_lds_reads_drained(B, anywin):
    name, ins = kernel(functions(B.device_asm("k_mdct2.hip")), r"k_mdct_long_v2I%sE" % anywin)
    groups = check_h1(ins, name)                                        # H1
    assert sum(b - a >= 3 for a, b in groups) >= 2                      # prologue + in-loop stagings
    # the frame's PCM is waited for with vmcnt(0) before its first LDS read in the loop
    loop_heads = [k for k, i in enumerate(ins) if i.endswith("LOOPHEAD")]
    k0 = next(k for k, i in enumerate(ins) if i == "s_barrier")
    first_ds = next(k for k in range(k0, len(ins)) if is_lds_access(ins[k]) and "read" in ins[k])
    assert any(i.startswith("s_waitcnt") and "vmcnt(0)" in i for i in ins[k0:first_ds]), \
        f"{name}: no s_waitcnt vmcnt(0) between the table barrier and the first tile read"
    assert loop_heads


def test_x2_two_tile_variant(B):
    name, ins = kernel(functions(B.device_asm("k_mdct3.hip")), r"k_mdct_long_x2ILi8ELi2E")
    check_h1(ins, name)


def test_checker_catches_the_round1_race():
    """the failing shape of round 1 (tile reads followed by the next frame's DMA with nothing
    between) and a wait that does not cover LGKM must be rejected; the fixed shape accepted"""
    dma = ["global_load_lds_dwordx4 v[2:3], off", "global_load_lds_dwordx4 v[2:3], off offset:1024",
           "global_load_lds_dwordx4 v[2:3], off offset:2048", "global_load_lds_dwordx4 v[2:3], off offset:3072"]
    bad1 = ["ds_read_b128 v[0:3], v9", "v_add_f64 v[4:5], v[0:1], v[2:3]"] + dma
    bad2 = ["ds_read_b128 v[0:3], v9", "s_waitcnt vmcnt(0)"] + dma
    good = ["ds_read_b128 v[0:3], v9", "s_waitcnt lgkmcnt(0)", "s_mov_b32 m0, s5"] + dma
    for bad in (bad1, bad2):
        with pytest.raises(AssertionError):
            check_h1(bad, "synthetic")
    assert check_h1(good, "synthetic") == [(3, 6)]
