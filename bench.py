#!/usr/bin/env python3
"""Benchmark of the MI355X encode path (BASELINE.json metric: audio channel-frames/s
encode, 48 kHz, 1024-line long blocks, 128 kb/s/ch; MDCT HBM GB/s vs roofline).

    python bench.py --gpus N --steps K --warmup W

N > 1 from the plain command: this process starts `python -m torch.distributed.run --nproc-per-node N`
on this very file as a CHILD process (before anything here touches the GPU; it never does itself),
relays rank 0's JSON line and the exit code -- the way the reference starts its own workers from one
command (coder/pacfile.py:771-781, Pool(8).starmap).  Launched under torch.distributed.run already
(WORLD_SIZE set), it is a rank and runs the bench.

A step = one pass of the whole hot path over one device-resident batch of
BASELINE configs[1]: 4096 synthetic 48 kHz stereo frames (8192 channel-frames) per
GPU -> window, MDCT, psychoacoustic SMR, bit allocation, scale factors +
mantissas, .pac bit packing and body assembly; with N > 1 the packed bitstream
of every rank is then gathered to rank 0 over RCCL.  Weak scaling: every rank
encodes its own 4096-frame shard.  Prints ONE JSON line on rank 0.

Timing: after W warm-up steps the K-step region (barrier + synchronize on both sides,
maximum over ranks) is timed --repeats times; `value` and `ms_per_step` come from the
MEDIAN region, the spread is reported in config.  After the timing, outside it, 32
channel-frames spread over the batch are re-encoded by the oracle (the CPU restatement of
the reference) and their payload bytes compared with what the timed kernels left in the
output buffers: `verified_cf`.  The body must also have fitted its buffer.

--corpus: BASELINE configs[4]: the config-2 stream tiled, tile t scaled
by 0.5 + 0.5 (t mod 16)/16, 1 048 576 stereo frames in all (--corpus-frames), sharded into
contiguous hop ranges with a one-hop halo: STRONG scaling (total work fixed), `value` with
the final gather inside the step and, in config, without it.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("PACX_AUTOBUILD", "1")     # a stale libpacx.so is rebuilt (and reported on stderr), never loaded

FRAMES_PER_GPU = 4096          # stereo frames per step and GPU (configs[1])
N_CH = 2
SAMPLE_RATE = 48000
KBPS = 128
MDCT_BYTES_PER_CF = 1024 * 2 + 1024 * 8      # int16 hop in + float64 lines out
HBM_PEAK_GBS = 8000.0
CORPUS_FRAMES = 1 << 20        # configs[4]: 1 048 576 stereo frames
# the reference ITSELF (its own Python, imported in the build container where /root/reference exists; it
# cannot travel to the GPU box): BASELINE.md section 3.1, tools/ref_cpu_timing.py -- quoted, not measured here
REFERENCE_CPU = {"one_process": 32.2, "pool8": 234.0, "unit": "channel-frames/s", "cores": 8,
                 "sample": "the reference's own PCMFile -> PACFile loop (coder/pacfile.py:674-757) on the first 120 "
                           "hops of test_signals/harpsichord.wav, config 1 settings; one process, and eight the way "
                           "its driver parallelises (Pool(8), coder/pacfile.py:780-781), 8-core build container",
                 "source": "BASELINE.md section 3.1 (tools/ref_cpu_timing.py); quoted, not measured by this run"}


def step_kernels(workload, n_cf):
    """per-kernel figures of the step from the COMMITTED profiles of this command (rocprofv3 kernel trace + SQ
    counter passes, tools/step_kernels.py -> profiles/r03_step_kernels.json): average microseconds, VALU-busy
    fraction, PMC bytes.  Quoted with their source; a bench run by itself has no access to the counters."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r03_step_kernels.json")))
        e = d.get(workload)
        if e and e.get("cf_per_step") == n_cf:
            return dict(e, source="profiles/r03_step_kernels.json (committed profile of this command; not measured by this run)")
    except Exception:
        pass
    return None


# ------------------------------------------------------------------ CPU baseline
def _cpu_worker(job):
    """one process: encode `per` stereo frames of a synthetic stream (seed offset i) with the
    oracle.  Only per + 1 hops are generated: a worker's memory stays at a few tens of MB."""
    i, per, vq_kbps = job
    from oracle import pac_oracle as po
    import audio_codec_amd as A
    pcm = A.synth.stream(per, N_CH, seed=A.synth.SEED + i)
    halo = np.concatenate((np.zeros((1024, N_CH), np.int16), pcm))
    if vq_kbps:
        from oracle import pac_oracle_vq as pv
        p = pv.make_params_vq(SAMPLE_RATE, N_CH, vq_kbps)
        fn = pv.encode_channel_sbr_vq if p.useSBR else pv.encode_channel_vq
    else:
        p = po.make_params(SAMPLE_RATE, N_CH, KBPS)
        fn = po.encode_channel
    t0 = time.perf_counter()
    for f in range(per):
        for ch in range(N_CH):
            fn(po.pcm16_to_fraction(halo[f * 1024:f * 1024 + 2048, ch]), p)
    return time.perf_counter() - t0


def usable_cores():
    """host cores this process may run on: the scheduler affinity (a GPU box hands a job a
    share of its cores), never more than os.cpu_count(), capped at 32 workers"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, os.cpu_count() or 1, 32))


def cpu_baseline(vq_kbps=None, frames_per_core=None):
    """The oracle (NumPy restatement of the reference, kind 'port') on the same synthetic
    workload: ALL usable host cores (one process per core, each its own run of hops -- the
    reference parallelises the same way, over independent encodes with Pool(8),
    coder/pacfile.py:780-781), one core alone, and BASELINE configs[0]'s excerpt
    (harpsichord, 44.1 kHz, the committed 64-hop fixture) through the oracle's file loop."""
    import subprocess
    from oracle import pac_oracle as po
    cores = usable_cores()
    per = frames_per_core or (96 if vq_kbps else 160)
    what = "oracle/pac_oracle_vq.py" if vq_kbps else "oracle/pac_oracle.py encode_channel"
    dt1 = _cpu_worker((0, per, vq_kbps))
    # one plain child process per core (no fork of a process that holds the GPU, no
    # multiprocessing start-method subtleties): each imports this file and runs _cpu_worker
    code = ("import sys; sys.path.insert(0, %r); import bench; "
            "print(bench._cpu_worker((int(sys.argv[1]), %d, %r)))" % (ROOT, per, vq_kbps))
    t0 = time.perf_counter()
    procs = [subprocess.Popen([sys.executable, "-c", code, str(i)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True) for i in range(cores)]
    for pr in procs:
        so, se = pr.communicate(timeout=600)
        if pr.returncode != 0:
            raise RuntimeError("cpu_baseline worker failed: " + se[-400:])
    wall = time.perf_counter() - t0
    out = {"value": cores * per * N_CH / wall, "unit": "channel-frames/s", "cores": cores, "kind": "port",
           "sample": f"{cores} processes (usable cores: {cores}, os.cpu_count() = {os.cpu_count()}) x {per} stereo "
                     f"frames of the same kind of synthetic stream, {what}, {wall:.1f} s wall including process start-up",
           "one_core": {"value": per * N_CH / dt1, "cores": 1,
                        "sample": f"{per} stereo frames ({per * N_CH} cf), {dt1:.1f} s"},
           "reference": REFERENCE_CPU}
    if not vq_kbps:
        ex = np.load(os.path.join(ROOT, "tests", "golden", "excerpt_harpsichord.npz"))
        pcm = ex["pcm"][:64 * 1024]
        t0 = time.perf_counter()
        po.encode_stream(pcm, int(ex["sr"]), KBPS, block_switching=False)
        dtc = time.perf_counter() - t0
        out["config1_excerpt"] = {"value": (len(pcm) // 1024 + 2) * pcm.shape[1] / dtc, "cores": 1,
                                  "sample": "BASELINE configs[0]: 64-hop excerpt of harpsichord.wav (44.1 kHz "
                                            f"stereo, tests/golden), long blocks, 128 kb/s, whole file loop, {dtc:.1f} s"}
    return out


def cpu_baseline_bs(pcm, sample_rate, n_hops_cpu=192, vq_kbps=None):
    """block-switched workloads: the oracle's whole file loop on the first hops of the same stream (scalar
    mantissas, or with vq_kbps the reference driver's own settings: gain-shape coder, SBR below 128 kb/s)"""
    t0 = time.perf_counter()
    if vq_kbps:
        from oracle import pac_oracle_vq as pv
        n_hops_cpu = 96
        pv.encode_stream_vq(pcm[:n_hops_cpu * 1024], sample_rate, vq_kbps)
        what = "oracle/pac_oracle_vq.py encode_stream_vq (gain-shape coder, block switching on)"
    else:
        from oracle import pac_oracle as po
        po.encode_stream(pcm[:n_hops_cpu * 1024], sample_rate, KBPS, block_switching=True)
        what = "oracle/pac_oracle.py encode_stream (block switching on)"
    dtc = time.perf_counter() - t0
    return {"value": (n_hops_cpu + 2) * N_CH / dtc, "unit": "channel-frames/s", "cores": 1, "kind": "port",
            "sample": f"first {n_hops_cpu} hops of the same tiled stream through {what}, {dtc:.1f} s"}


# ------------------------------------------------------------------ verification
def verify_against_oracle(pcm, sample_rate, kbps, vq_kbps, flags, payload_rows, n_bytes, picks, halo=None):
    """The hops `picks` of the batch (2 channel-frames each) through the oracle; their .pac
    payload bytes must equal what the timed run left in its slots.  pcm: the rank's int16
    stream [n_hops*1024, nCh]; halo: int16 [1024, nCh], the hop before it (None = zeros, the
    start of a file); flags: uint8 per frame or None; payload_rows: {cf: uint8 slot}.
    Returns the number of channel-frames verified."""
    from oracle import pac_oracle as po
    if vq_kbps:
        from oracle import pac_oracle_vq as pv
        p = pv.make_params_vq(sample_rate, pcm.shape[1], vq_kbps)
    else:
        p = po.make_params(sample_rate, pcm.shape[1], kbps)
    n_ch = pcm.shape[1]
    ok = 0
    for f in picks:
        before = pcm[(f - 1) * 1024:f * 1024] if f else (halo if halo is not None else np.zeros((1024, n_ch), np.int16))
        prior = [po.pcm16_to_fraction(before[:, ch]) for ch in range(n_ch)]
        hop = [po.pcm16_to_fraction(pcm[f * 1024:(f + 1) * 1024, ch]) for ch in range(n_ch)]
        fl = int(flags[f]) if flags is not None else 0
        fl3 = (fl & 1, (fl >> 1) & 1, (fl >> 2) & 1)
        if flags is not None:
            # the detector's own decision for this hop (coder/detect_transients.py:5-23): next flag of frame f
            look = np.concatenate((np.array(hop), np.zeros((n_ch, 1024))), axis=1)
            assert int(bool(po.transient_detect(look))) == fl3[2], f"transient flag of hop {f}"
        parts = (pv.encode_hop_vq if vq_kbps else po.encode_hop)(p, prior, hop, fl3)
        for ch in range(n_ch):
            i = f * n_ch + ch
            n = int(n_bytes[i])
            if parts is None:                      # hop the reference drops (all-zero sub-block)
                assert n == 0, f"cf {i}: dropped hop has a payload"
                continue
            nb, want = (pv.pack_channel_block_vq if vq_kbps else po.pack_channel_block)(p, fl3, parts[ch])
            assert n == nb, f"cf {i}: {n} payload bytes, oracle {nb}"
            got = payload_rows[i][:n].tobytes()
            assert got == want, f"cf {i}: payload bytes differ from the oracle's"
            ok += 1
    return ok


# ------------------------------------------------------------------ N ranks from one command
def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(n, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: start the N ranks as
    children (torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1 at a free port),
    pass their stderr through, print rank 0's ONE JSON line, return the launcher's exit code.
    This process never initialises the GPU (no torch import here): the ranks are child
    processes, not an exec of a process that holds the device."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.abspath(__file__)] + list(argv)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env, cwd=ROOT)
    lines = []
    for ln in proc.stdout:
        if ln.startswith('{"metric"'):
            lines.append(ln.rstrip("\n"))
        else:                                   # anything else a rank printed: to stderr, the JSON line stays alone
            sys.stderr.write(ln)
    rc = proc.wait()
    if rc == 0 and len(lines) != 1:
        sys.stderr.write(f"bench: expected one JSON line from rank 0, got {len(lines)}\n")
        rc = 1
    if lines:
        print(lines[-1], flush=True)
    return rc


def launcher_selftest(args):
    """--launcher-selftest: what a rank does when only the launch itself is under test (the CPU test of
    `python bench.py --gpus 2` in a container without a GPU): join the process group over gloo, see every
    rank, rank 0 prints one line.  No encode, no value."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but the launcher started {world} ranks"
    dist.init_process_group("gloo")
    seen = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(seen, torch.tensor([rank], dtype=torch.int64))
    dist.barrier()
    if rank == 0:
        print(json.dumps({"metric": "launcher self-test (no encode)", "value": None, "unit": "channel-frames/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ranks_seen": [int(t.item()) for t in seen]}), flush=True)
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--repeats", type=int, default=10, help="timed K-step regions (at least); value = median")
    ap.add_argument("--no-decode-leg", action="store_true",
                    help="skip the decode figure (config.decode_cf_per_s: the timed run's payloads back to PCM; never `value`)")
    ap.add_argument("--gate-us", type=float, default=None,
                    help="two steps in flight: hold both pipelines' streams for this long at the start of every timed region "
                         "(INSIDE it) so that their first steps start together and the pipelines stay in phase "
                         "(engine.EncoderPool.align; DESIGN.md 5.0); 0 = off; default: 20 for all-long scalar batches "
                         "(the only workload with two timing modes), 0 otherwise")
    ap.add_argument("--min-seconds", type=float, default=1.5,
                    help="keep timing K-step regions until this much wall time has gone into them (and --repeats regions "
                         "are done): the median then describes the card under sustained load, clocks settled")
    ap.add_argument("--frames", type=int, default=None, help="stereo frames per GPU and step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--mdct-launches", type=int, default=50)
    ap.add_argument("--graph", dest="graph", action="store_true", default=None,
                    help="the step's kernel launches are captured once into a hipGraph and the timed regions replay "
                         "it: the launch gaps between the step's kernels go.  Default: on only with --pipeline 1 and the "
                         "scalar coder (scalar128 +2.6 %%, bs128 +3.4 %% there); with two steps in flight direct launches "
                         "overlap better, and the gain-shape coder's two launches on two streams do not overlap when "
                         "replayed (shipped128: 0.82 against 0.69 ms)")
    ap.add_argument("--no-graph", dest="graph", action="store_false",
                    help="launch the step's kernels one by one")
    ap.add_argument("--launcher-selftest", action="store_true",
                    help="ranks only join the process group (gloo) and rank 0 prints one line: tests the "
                         "self-launch of `bench.py --gpus N` where there is no GPU")
    ap.add_argument("--workload", choices=["scalar128", "vq128", "vq96", "bs128", "shipped128", "shipped96"],
                    default="scalar128",
                    help="scalar128 = BASELINE configs[1] (the headline); vq128 / vq96 = the gain-shape "
                         "coder of configs[3] (vq96 with SBR) on the same synthetic stream; bs128 = "
                         "configs[2]: block switching on, the castanet stream (the whole file as the "
                         "reference's driver reads it, tests/golden/full_castanet.npz) tiled to --frames "
                         "hops (44.1 kHz), transient detector inside the step; shipped128 / shipped96 = the "
                         "settings the reference's own driver uses (coder/pacfile.py:699-707: gain-shape coder, "
                         "block switching, SBR below 128 kb/s) on that castanet stream")
    ap.add_argument("--corpus", action="store_true",
                    help="BASELINE configs[4]: the tiled, level-scaled corpus sharded over the ranks "
                         "(strong scaling); --corpus-frames stereo frames in all")
    ap.add_argument("--corpus-frames", type=int, default=CORPUS_FRAMES)
    ap.add_argument("--pipeline", type=int, default=2,
                    help="steps in flight: consecutive steps (independent batches) alternate between this many "
                         "handles, each with its own stream, workspaces, output buffers and hipGraph -- the kernels of "
                         "step i+1 fill the vector-issue slots those of step i leave idle (every kernel of the step is "
                         "latency- or issue-bound at 50-60 %% VALU utilisation: DESIGN.md 5.3).  Every step is still one "
                         "full pass over one batch and all K of them are inside the timed region.  1 = one step at a "
                         "time (round 2's figure; reported as config.value_one_step_in_flight either way)")
    ap.add_argument("--host-stream-frames", type=int, default=65536,
                    help="stereo frames per chunk of the host-memory-to-host-memory measurement reported as "
                         "config.host_to_host_cf_per_s (pinned PCM in, packed bodies out, copies overlapped with the "
                         "kernels: audio-codec_amd/streaming.py); 0 = skip.  Never `value`.")
    args = ap.parse_args()

    if (args.gpus > 1 or os.environ.get("PACX_BENCH_FORCE_DIST")) and "WORLD_SIZE" not in os.environ:
        # the plain command: start the N ranks ourselves, as children (nothing above has touched the GPU)
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    if args.launcher_selftest:
        return launcher_selftest(args)

    import torch
    import torch.distributed as dist
    import audio_codec_amd as A

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # PACX_BENCH_FORCE_DIST=1: take the N > 1 path (process group over RCCL, barriers, max over ranks,
    # asynchronous gather of the bodies) with whatever world size the launcher gave, 1 included -- the
    # rehearsal of that control flow on the one GPU of a test box (tests/test_gpu_rccl.py)
    multi = world > 1 or bool(os.environ.get("PACX_BENCH_FORCE_DIST"))
    if args.gpus > 1 or multi:
        assert world == args.gpus, f"--gpus {args.gpus} but the launcher started {world} ranks"
        rehearsal = bool(os.environ.get("PACX_BENCH_ONE_GPU"))
        if rehearsal:
            # rehearsal of the N > 1 control flow on a 1-GPU box: every rank on cuda:0, the
            # gather over gloo with host staging (RCCL refuses two ranks on one device)
            local = 0
        torch.cuda.set_device(local)
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())

    # ---- workload: device-resident before any timing
    vq_kbps = {"scalar128": None, "vq128": 128, "vq96": 96, "bs128": None, "shipped128": 128,
               "shipped96": 96}[args.workload]
    kbps = vq_kbps or KBPS
    block_switched = args.workload in ("bs128", "shipped128", "shipped96")
    if args.gate_us is None:
        args.gate_us = float(os.environ.get("PACX_POOL_GATE_US", "0" if (block_switched or vq_kbps) else "20"))
    sample_rate = SAMPLE_RATE
    corpus = args.corpus
    if corpus:
        assert not block_switched, "--corpus is the synthetic 48 kHz corpus"
        # this rank's contiguous hop range of the corpus, with its one-hop halo (no exchange)
        lo, hi = A.dist.shard_bounds(args.corpus_frames, world, rank)
        n_frames = hi - lo
        shard = A.synth.corpus_shard(lo, hi)                       # planar [nCh, (n_frames+1)*1024]
        pcm = None
    else:
        n_frames = args.frames or FRAMES_PER_GPU
        if block_switched:
            full = np.load(os.path.join(ROOT, "tests", "golden", "full_castanet.npz"))
            sample_rate = int(full["sr"])
            src = full["pcm"]
            src = src[:len(src) // 1024 * 1024]
            reps = -(-n_frames * 1024 // len(src))
            pcm = np.ascontiguousarray(np.tile(src, (reps, 1))[:n_frames * 1024])
        else:
            pcm = A.synth.stream(n_frames, N_CH, seed=A.synth.SEED + rank)
        shard = A.synth.planar_with_halo(pcm)
    P = max(1, args.pipeline)
    if multi:
        P = 2                                  # the gather's two send buffers
    # the product's pool of handles and streams (engine.EncoderPool): consecutive steps alternate between its slots
    pool = A.engine.EncoderPool(P, sample_rate, kbps / (sample_rate / 1000), use_vq=bool(vq_kbps),
                                use_sbr=bool(vq_kbps and vq_kbps < 128))
    encs = pool.encs
    enc = encs[0]
    planar = torch.as_tensor(shard, device=dev)
    view = A.engine.PcmView.stream(planar)
    n_cf = view.n_cf
    hop_view = None
    if block_switched:      # the hops as the transient detector sees them (hop h = planar hop h+1)
        hop_view = A._lib.PacxPcm(planar.data_ptr() + 2 * 1024, A._lib.PCM_I16, N_CH, n_frames, 1024,
                                  planar.shape[1], 1)
    import ctypes
    from audio_codec_amd.engine import _ptr
    slot = A.dist.slot_bytes(n_cf, kbps / (sample_rate / 1000))      # bound on one rank's body
    gather = None
    if multi:
        # fixed-slot asynchronous gather of the packed bodies to rank 0 (RCCL): two send
        # buffers alternate, the gather of step i overlaps the encode of step i+1, no host
        # synchronisation and no size exchange inside a step
        on_host = dist.get_backend() == "gloo"          # rehearsal only
        gather = A.dist.BitstreamGather(slot, torch.device("cpu") if on_host else dev)
        bodies = [gather.body(0), gather.body(1)]
        if on_host:
            host_bodies, bodies = bodies, [torch.empty_like(b, device=dev) for b in bodies]
    else:
        bodies = [torch.empty(slot, dtype=torch.uint8, device=dev) for _ in range(P)]
    cap = int(bodies[0].numel())
    # one pipeline = one handle with its workspaces, its own output buffers, body, stream (and hipGraph)
    pipes = []
    for q in range(P):
        e = encs[q]
        e.reserve(n_cf)
        o = e.alloc_outputs(n_cf, with_payload=True)
        if corpus:
            o["mantissa"] = None               # 1 GB per 131 072 frames nobody reads
        pipes.append({"enc": e, "out": o, "body": bodies[q], "total": torch.zeros(1, dtype=torch.int64, device=dev),
                      "vq_out": {k: o[k] for k in ("overall", "bit_alloc", "status", "payload", "n_bytes")} if vq_kbps else None,
                      "tr": torch.empty(n_frames, dtype=torch.uint8, device=dev) if block_switched else None,
                      "fl": torch.empty(n_frames + 2, dtype=torch.uint8, device=dev) if block_switched else None,
                      "stream": pool.streams[q], "graph": None})
    out, total = pipes[0]["out"], pipes[0]["total"]
    fl_buf = pipes[0]["fl"]
    step_no = [0]
    do_gather = [True]
    in_flight = [P]

    def encode_part(q):
        """the kernels of one step on pipeline q: what a hipGraph of the step holds"""
        pp = pipes[q]
        e, o = pp["enc"], pp["out"]
        if block_switched:
            e._call("pacx_transient_flags", ctypes.byref(hop_view), _ptr(pp["tr"]), _ptr(pp["fl"]), e._stream())
        flags = pp["fl"][:n_frames] if block_switched else None
        if vq_kbps:
            e.encode_vq(view, flags, pp["vq_out"])
        else:
            e.encode_pack(view, flags, o)
        e._call("pacx_gather_body", ctypes.c_int64(n_cf), _ptr(o["payload"]), _ptr(o["n_bytes"]),
                _ptr(pp["body"]), ctypes.c_int64(cap), _ptr(pp["total"]), e._stream())

    # the kernel launches of a step captured once into a hipGraph (one per pipeline) and replayed.
    # The gather of the bodies (RCCL) stays outside the graph.
    if args.graph is None:                 # measured: with two steps in flight direct launches overlap better than two replayed
        args.graph = False                 # graphs (50.6 against 46.6 M cf/s); with one, the step on ONE stream needs no graph
                                           # either (42.6 M direct, 41.5 M replayed; the forked step 39.1 / 40.3 M)
    torch.cuda.synchronize()
    if args.graph:
        for q in range(P):
            with torch.cuda.stream(pipes[q]["stream"]):
                encode_part(q)
        torch.cuda.synchronize()
        try:
            for q in range(P):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=pipes[q]["stream"]):
                    encode_part(q)
                pipes[q]["graph"] = g
        except Exception as e:                      # capture not possible: plain launches
            print(f"bench: hipGraph capture failed ({e}); launching kernels directly", file=sys.stderr)
            for pp in pipes:
                pp["graph"] = None
        torch.cuda.synchronize()
    graphs = [pp["graph"] for pp in pipes]
    dump_regions = os.environ.get("PACX_BENCH_DUMP_REGIONS")      # every region's ms/step, one per line (tools/bench_regions.py)

    def step():
        q = step_no[0] % in_flight[0]
        pp = pipes[q]
        with torch.cuda.stream(pp["stream"]):
            if gather is not None:
                gather.wait(q)                      # the gather that last used this buffer (stream-level wait)
            if pp["graph"] is not None:
                pp["graph"].replay()
            else:
                encode_part(q)
            if gather is not None and do_gather[0]:
                if dist.get_backend() == "gloo":            # rehearsal: stage through the host
                    host_bodies[q].copy_(pp["body"])
                    gather.launch(q, pp["total"].cpu())
                else:
                    gather.launch(q, pp["total"])
        step_no[0] += 1

    def sync_all():
        if gather is not None:
            for k in range(len(bodies)):
                gather.wait(k)
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    enqueue_s = []

    def timed_region():
        sync_all()
        t0 = time.perf_counter()
        if in_flight[0] > 1 and args.gate_us > 0:
            pool.align(args.gate_us)                # inside the timed region: both pipelines' first steps start together
        for _ in range(args.steps):
            step()
        enqueue_s.append(time.perf_counter() - t0)     # the host's share: all K steps queued (nothing waited for yet)
        sync_all()
        dt = time.perf_counter() - t0
        if multi:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    for _ in range(args.warmup):
        step()
    regions, t_begin = [], time.perf_counter()
    while len(regions) < max(1, args.repeats) or (time.perf_counter() - t_begin < args.min_seconds and len(regions) < 20000):
        regions.append(timed_region())
        if multi and len(regions) >= max(1, args.repeats):      # ranks must agree on when to stop
            flag = torch.tensor([1.0 if time.perf_counter() - t_begin < args.min_seconds else 0.0], dtype=torch.float64,
                                device=torch.device("cpu") if dist.get_backend() == "gloo" else dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if float(flag.item()) == 0.0:
                break
    dt = float(np.median(regions))
    if dump_regions and rank == 0:
        with open(dump_regions, "w") as f:
            f.write("\n".join(f"{r / args.steps * 1e3:.5f}" for r in regions) + "\n")
    regions_one = None
    if P > 1 and not multi:                # the same K steps with one step in flight at a time (round 2's figure)
        in_flight[0] = 1
        step_no[0] = 0
        one_graph = None
        pipes[0]["enc"].set_side_fork(False)            # that mode's best form: the whole step on one stream (no fork across
        torch.cuda.synchronize()                        # hardware queues), direct launches
        regions_one = [timed_region() for _ in range(max(3, args.repeats // 2))]
        if one_graph is not None:
            pipes[0]["graph"] = None
        pipes[0]["enc"].set_side_fork(True)
        in_flight[0] = P
    regions_nogather = None
    if corpus and multi:                   # with and without the gather
        do_gather[0] = False
        regions_nogather = [timed_region() for _ in range(max(1, args.repeats))]
        do_gather[0] = True

    # ---- what was timed, checked (outside the timed regions)
    torch.cuda.synchronize()
    if gather is not None:
        for k in range(len(bodies)):
            gather.check(k)                    # every rank: its own body fitted its slot
    body_bytes = int(total.item())
    assert 0 < body_bytes <= cap, f"body of {body_bytes} bytes does not fit its {cap}-byte buffer"
    n_bytes_h = out["n_bytes"].cpu().numpy()
    assert body_bytes == int(n_bytes_h.astype(np.int64).sum()) + 4 * int(np.count_nonzero(n_bytes_h)), \
        "body size is not the sum of its records"
    verified = 0
    if not args.no_verify:                 # EVERY rank checks its own output (its own stream or shard)
        picks = sorted(set(np.linspace(0, n_frames - 1, 16).astype(int).tolist()))
        rows = {f * N_CH + ch: out["payload"][f * N_CH + ch].cpu().numpy() for f in picks for ch in range(N_CH)}
        flags_h = fl_buf[:n_frames].cpu().numpy() if block_switched else None
        stream_h = pcm if pcm is not None else np.ascontiguousarray(shard[:, 1024:].T)
        halo_h = None if pcm is not None else np.ascontiguousarray(shard[:, :1024].T)
        verified = verify_against_oracle(stream_h, sample_rate, kbps, vq_kbps, flags_h, rows, n_bytes_h, picks, halo_h)
        for pp in pipes[1:]:               # every pipeline encoded the same batch: same records
            nb_q = pp["out"]["n_bytes"].cpu().numpy()
            assert np.array_equal(nb_q, n_bytes_h), "pipelines disagree on the record lengths"
            for i, row in rows.items():
                assert np.array_equal(pp["out"]["payload"][i].cpu().numpy()[:int(nb_q[i])], row[:int(nb_q[i])]), \
                    f"cf {i}: the pipelines' payloads differ"
    verified_per_rank = [verified]
    if multi:                              # rank 0 reports every rank's count; a rank whose check fails has raised
        cnt = torch.tensor([verified], dtype=torch.int64,
                           device=torch.device("cpu") if dist.get_backend() == "gloo" else dev)
        cnts = [torch.zeros_like(cnt) for _ in range(world)]
        dist.all_gather(cnts, cnt)
        verified_per_rank = [int(c.item()) for c in cnts]
        if not args.no_verify:
            assert min(verified_per_rank) > 0, f"a rank verified nothing: {verified_per_rank}"
        verified = sum(verified_per_rank)

    # ---- MDCT kernel alone: HIP events on the stream the kernel runs on
    mdct_cf = min(n_cf, 2 * FRAMES_PER_GPU) if corpus else n_cf
    mview = view
    if mdct_cf != n_cf:
        mview = A.engine.PcmView.stream(planar[:, :(mdct_cf // N_CH + 1) * 1024].contiguous())
    lines = torch.empty((mdct_cf, 1024), dtype=torch.float64, device=dev)
    scale = torch.empty((mdct_cf,), dtype=torch.int32, device=dev)

    def mdct_once():
        enc._call("pacx_mdct_batch", ctypes.byref(mview.c), None, 0, _ptr(lines), _ptr(scale), enc._stream())
    for _ in range(5):
        mdct_once()
    torch.cuda.synchronize()
    # HIP events on the launch stream around trains of back-to-back launches:
    # (train time / launches) = kernel duration + the ~1 us kernel boundary, with
    # the host launch latency hidden behind the previous kernel
    train = 10
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(max(1, args.mdct_launches // train))]
    for a, b in ev:
        a.record()
        for _ in range(train):
            mdct_once()
        b.record()
    torch.cuda.synchronize()
    trains = [a.elapsed_time(b) / train for a, b in ev]
    mdct_ms = float(np.median(trains))
    mdct_gbs = mdct_cf * MDCT_BYTES_PER_CF / (mdct_ms * 1e-3) / 1e9

    # HBM bytes of the MDCT kernel: NOT measured by this run (PMC counters need rocprofv3);
    # the figure of the committed PMC passes of this same command is quoted, with its
    # source, when the launch geometry matches
    traffic, traffic_src = None, None
    for name in ("r03_mdct_pmc.json", "r02_mdct_pmc.json", "r01_mdct_pmc.json"):
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", name)))
            if pmc["cf_per_launch"] == mdct_cf:
                traffic, traffic_src = pmc["hbm_bytes_per_launch"], f"profiles/{name}"
                break
        except Exception:
            pass

    # ---- host memory to host memory (never `value`): the same coder fed from pinned host buffers in chunks, the
    # PCIe copies of neighbouring chunks overlapping the kernels (streaming.HostStreamEncoder)
    host_stream = None
    if rank == 0 and not multi and not corpus and args.host_stream_frames > 0:
        hf = args.host_stream_frames
        hs = A.streaming.HostStreamEncoder(enc, N_CH, hf, depth=2, block_switching=block_switched)
        src_planar = shard[:, 1024:]                      # the workload's own stream, repeated to fill a chunk
        reps = -(-hf * 1024 // src_planar.shape[1])
        chunk = np.ascontiguousarray(np.tile(src_planar, (1, reps))[:, :hf * 1024])
        for k in range(2):
            hs.input(k)[:] = chunk
        for i in range(4):                                 # warm-up, and the bodies' lengths get known
            k = i & 1
            if i >= 2:
                hs.result(k)
            hs.submit(k)
        hs.result(0), hs.result(1)
        torch.cuda.synchronize()
        n_chunks = 10
        t0 = time.perf_counter()
        out_bytes = 0
        for i in range(n_chunks):
            k = i & 1
            if i >= 2:
                out_bytes += len(hs.result(k))
            hs.submit(k)
        out_bytes += len(hs.result(0)) + len(hs.result(1))
        dt_h = time.perf_counter() - t0
        host_stream = {"cf_per_s": n_chunks * hf * N_CH / dt_h, "chunk_cf": hf * N_CH, "chunks": n_chunks,
                       "ms_per_chunk": dt_h / n_chunks * 1e3, "pcm_bytes_per_chunk": int(chunk.nbytes),
                       "body_bytes_per_chunk": out_bytes // n_chunks,
                       "how": "pinned int16 PCM -> HBM, encode + pack + body, body -> pinned host memory; three streams, "
                              "two buffers of everything, no host synchronisation inside the loop but taking a finished body"}
        del hs

    # ---- the other direction (never `value`): the timed run's payloads of pipeline 0 back to 16-bit PCM
    # (unpack / gain-shape decode, IMDCT, overlap-and-add), one step in flight on the current stream
    decode_leg = None
    if rank == 0 and not multi and not corpus and not args.no_decode_leg:
        o = pipes[0]["out"]
        e0 = pipes[0]["enc"]

        def decode_step():
            if vq_kbps:
                return e0.decode_vq(o["payload"], o["n_bytes"], N_CH)["pcm"]
            return e0.decode(e0.unpack(o["payload"], o["n_bytes"]), N_CH)
        for _ in range(2):
            pcm_dec = decode_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n_dec = 5
        for _ in range(n_dec):
            pcm_dec = decode_step()
        torch.cuda.synchronize()
        dt_d = (time.perf_counter() - t0) / n_dec
        decode_leg = {"cf_per_s": n_cf / dt_d, "ms_per_step": dt_d * 1e3, "pcm_samples": int(pcm_dec.shape[0]),
                      "what": "payload slots of the timed run -> " + ("gain-shape decode" if vq_kbps else "unpack + dequantise")
                              + " -> IMDCT -> overlap-and-add -> int16 PCM, one step in flight"}
        del pcm_dec
        if P > 1:                                   # the same with two decode steps in flight (the pool's two handles and streams)
            def decode_on(q):
                oq, eq = pipes[q]["out"], pipes[q]["enc"]
                with pool.slot(q):
                    if vq_kbps:
                        return eq.decode_vq(oq["payload"], oq["n_bytes"], N_CH)["pcm"]
                    return eq.decode(eq.unpack(oq["payload"], oq["n_bytes"]), N_CH)
            keep = [decode_on(q % 2) for q in range(4)]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n_dec2 = 10
            keep = [decode_on(q % 2) for q in range(n_dec2)]
            torch.cuda.synchronize()
            dt_d2 = (time.perf_counter() - t0) / n_dec2
            decode_leg["cf_per_s_two_in_flight"] = n_cf / dt_d2
            del keep

    if rank == 0:
        total_cf = world * n_cf if not corpus else N_CH * args.corpus_frames
        what = ("scalar mantissas" if not vq_kbps else "gain-shape PVQ" + (" + SBR" if kbps < 128 else ""))
        if corpus:
            wl = (f"BASELINE configs[4]: corpus of {args.corpus_frames} synthetic 48 kHz stereo frames "
                  f"({total_cf} channel-frames: the configs[1] stream tiled, tile t scaled by 0.5+0.5(t mod 16)/16), "
                  f"{world} contiguous hop shards with a one-hop halo, {n_cf} channel-frames on rank 0, ")
        elif block_switched:
            wl = (f"{n_frames} stereo frames per GPU of castanet.wav as the reference's driver reads it, tiled "
                  f"({n_cf} channel-frames, 44.1 kHz), long + short blocks by the transient detector (inside the step), ")
        else:
            wl = (f"{n_frames} synthetic 48 kHz stereo frames per GPU ({n_cf} channel-frames), N=1024 long blocks, ")
        res = {
            "metric": "audio channel-frames/s encode (48 kHz, 1024-line long blocks, 128 kb/s/ch)"
                      if not (vq_kbps or block_switched)
                      else f"audio channel-frames/s encode, the reference driver's settings: gain-shape PVQ"
                           f"{' + SBR' if vq_kbps < 128 else ''} + block switching (castanet.wav tiled, 44.1 kHz, "
                           f"{vq_kbps} kb/s/ch)" if (vq_kbps and block_switched)
                      else "audio channel-frames/s encode, block switching on (castanet.wav tiled, 44.1 kHz, "
                           "128 kb/s/ch)" if block_switched
                      else f"audio channel-frames/s encode, gain-shape PVQ{' + SBR' if vq_kbps < 128 else ''} "
                           f"(48 kHz, 1024-line long blocks, {vq_kbps} kb/s/ch)",
            "value": total_cf * args.steps / dt,
            "unit": "channel-frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if corpus else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "castanet.wav PCM of tests/golden, tiled" if block_switched else "synthetic",
            "verified_cf": verified,
            "config": {"workload": wl + f"{kbps} kb/s/ch, {what}, int16 PCM resident in HBM; step = encode + "
                                        ".pac bit packing + body assembly" +
                                   (" + asynchronous RCCL gather of the bodies to rank 0" if multi else ""),
                       "stereo_frames_per_s": total_cf / N_CH * args.steps / dt,
                       "timing": f"median of {len(regions)} regions of {args.steps} steps",
                       "value_min": total_cf * args.steps / max(regions),
                       "value_max": total_cf * args.steps / min(regions),
                       "regions": len(regions),
                       "ms_per_step_regions": [r / args.steps * 1e3 for r in (regions if len(regions) <= 24 else
                                                                                regions[:8] + regions[-16:])],
                       "host_enqueue_ms_per_step": float(np.median(enqueue_s)) / args.steps * 1e3 if enqueue_s else None,
                       "host_enqueue_note": "median over the regions of the time the host takes to queue the K steps of a region "
                                            "(Python + HIP launches), per step; the regions' own time is ms_per_step: where the "
                                            "two are close the host, not the GPU, sets the rate",
                       "start_gate_us": args.gate_us if P > 1 else 0.0,
                       "start_gate_note": "inside every timed region: both pipelines' streams wait this long on a gate event so "
                                          "that their first steps start together (EncoderPool.align): started together the two "
                                          "pipelines stay in phase, free-running they settle per region into one of two relations "
                                          "(47.0 or 51.4 M cf/s on the headline batch, DESIGN.md 5.0)",
                       "ms_per_step_regions_note": "all regions" if len(regions) <= 24 else "the first 8 and the last 16 regions",
                       # two steps in flight settle, region by region, into one of two phase relations of the pipelines
                       # (DESIGN.md 5.0): the spread says how the regions of THIS run fell
                       "ms_per_step_quantiles": {q: sorted(regions)[min(len(regions) - 1, int(len(regions) * f))] / args.steps * 1e3
                                                 for q, f in (("p10", 0.10), ("p25", 0.25), ("p50", 0.50), ("p75", 0.75), ("p90", 0.90))},
                       "body_bytes_per_step": body_bytes,
                       "verified": f"{verified} channel-frames of the timed run's output re-encoded by the oracle, "
                                   "payload bytes equal" + (f" (per rank: {verified_per_rank})" if multi else "")
                                   if verified else "not verified",
                       "verified_per_rank": verified_per_rank,
                       "launch": ("hipGraph replay of the captured step" if graphs[0] is not None else "direct launches") +
                                 (f"; {P} steps in flight: consecutive steps alternate between {P} handles, each with its own "
                                  "stream, workspaces and output buffers" if P > 1 else "; one step in flight"),
                       "steps_in_flight": P,
                       "sharding": f"{world} x frame-range shards, no data-path collective; process group: "
                                   + (f"{dist.get_backend()} with {dist.get_world_size()} ranks" if multi
                                      else "none (one process)")},
            "roofline": {"kernel": "k_mdct_long_x2p<8, 2, false> (the stand-alone launches of the kernel; the step launches the <8, 2, true> instantiation, which also initialises status words: window + MDCT, int16 in, float64 lines out; two frames per wave alternating on one FFT tile, PCM prefetched a whole iteration ahead)",
                         "bound": "hbm", "achieved": mdct_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": mdct_gbs / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_note": ("HBM bytes per launch from the COMMITTED rocprofv3 PMC passes of this "
                                          f"command (2*FETCH_SIZE + WRITE_SIZE), {traffic_src}; not measured by this run")
                                         if traffic else "no committed PMC profile for this launch geometry",
                         "algorithmic_bytes_per_launch": mdct_cf * MDCT_BYTES_PER_CF,
                         "launch_ms": mdct_ms, "launch_ms_min": float(min(trains)), "launch_ms_max": float(max(trains)),
                         "timing": f"median of {len(trains)} HIP-event trains of {train} back-to-back launches",
                         "bytes_per_cf": MDCT_BYTES_PER_CF, "cf_per_launch": mdct_cf,
                         "mdct_cf_per_s": mdct_cf / (mdct_ms * 1e-3)},
        }
        if regions_one is not None:
            d1 = float(np.median(regions_one))
            res["config"]["value_one_step_in_flight"] = total_cf * args.steps / d1
            res["config"]["ms_per_step_one_step_in_flight"] = d1 / args.steps * 1e3
        if host_stream:
            res["config"]["host_to_host_cf_per_s"] = host_stream["cf_per_s"]
            res["config"]["host_to_host"] = host_stream
        if decode_leg:
            res["config"]["decode_cf_per_s"] = decode_leg["cf_per_s"]
            res["config"]["decode"] = decode_leg
        sk = step_kernels(args.workload, n_cf)
        if sk:
            res["config"]["step_kernels"] = sk
        if regions_nogather is not None:
            d2 = float(np.median(regions_nogather))
            res["config"]["value_without_gather"] = total_cf * args.steps / d2
            res["config"]["ms_per_step_without_gather"] = d2 / args.steps * 1e3
        if not args.no_cpu_baseline and not multi:
            if block_switched:
                res["cpu_baseline"] = cpu_baseline_bs(pcm, sample_rate, vq_kbps=vq_kbps)
            else:
                res["cpu_baseline"] = cpu_baseline(vq_kbps)
        print(json.dumps(res), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
