#!/usr/bin/env python3
"""Benchmark of the MI355X encode path (BASELINE.json metric: audio channel-frames/s
encode, 48 kHz, 1024-line long blocks, 128 kb/s/ch; MDCT HBM GB/s vs roofline).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A step = one pass of the whole hot path over one device-resident batch of
BASELINE configs[1]: 4096 synthetic 48 kHz stereo frames (8192 channel-frames) per
GPU -> window, MDCT, psychoacoustic SMR, bit allocation, scale factors +
mantissas, .pac bit packing and body assembly; with N > 1 the packed bitstream
of every rank is then gathered to rank 0 over RCCL.  Weak scaling: every rank
encodes its own 4096-frame shard.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FRAMES_PER_GPU = 4096          # stereo frames per step and GPU (configs[1])
N_CH = 2
SAMPLE_RATE = 48000
KBPS = 128
MDCT_BYTES_PER_CF = 1024 * 2 + 1024 * 8      # int16 hop in + float64 lines out
HBM_PEAK_GBS = 8000.0


def cpu_baseline(n_frames=768, vq_kbps=None):
    """The oracle (NumPy restatement of the reference, kind 'port') on the first
    n_frames stereo frames of the same workload, one host core."""
    from oracle import pac_oracle as po
    import audio_codec_amd as A
    pcm = A.synth.stream(n_frames, N_CH)
    halo = np.concatenate((np.zeros((1024, N_CH), np.int16), pcm))
    if vq_kbps:
        from oracle import pac_oracle_vq as pv
        p = pv.make_params_vq(SAMPLE_RATE, N_CH, vq_kbps)
        fn = pv.encode_channel_sbr_vq if p.useSBR else pv.encode_channel_vq
        what = "oracle/pac_oracle_vq.py " + fn.__name__
    else:
        p = po.make_params(SAMPLE_RATE, N_CH, KBPS)
        fn = po.encode_channel
        what = "oracle/pac_oracle.py encode_channel"
    t0 = time.perf_counter()
    for f in range(n_frames):
        for ch in range(N_CH):
            fn(po.pcm16_to_fraction(halo[f * 1024:f * 1024 + 2048, ch]), p)
    dt = time.perf_counter() - t0
    return {"value": n_frames * N_CH / dt, "unit": "channel-frames/s", "cores": 1, "kind": "port",
            "sample": f"first {n_frames} stereo frames ({n_frames * N_CH} cf) of the same synthetic stream, "
                      f"{what}, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=FRAMES_PER_GPU, help="stereo frames per GPU and step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mdct-launches", type=int, default=50)
    ap.add_argument("--graph", action="store_true",
                    help="replay a captured hipGraph of the step instead of launching its kernels one by one "
                         "(measured slower on ROCm 7.2: 0.408 vs 0.397 ms/step -- the queue is GPU-bound)")
    ap.add_argument("--workload", choices=["scalar128", "vq128", "vq96", "bs128"], default="scalar128",
                    help="scalar128 = BASELINE configs[1] (the headline); vq128 / vq96 = the gain-shape "
                         "coder of configs[3] (vq96 with SBR) on the same synthetic stream; bs128 = "
                         "configs[2]: block switching on, the castanet excerpt of tests/golden tiled to "
                         "--frames hops (44.1 kHz), transient detector inside the step")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import audio_codec_amd as A

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 or world > 1:
        assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node N"
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
    n_frames = args.frames
    vq_kbps = {"scalar128": None, "vq128": 128, "vq96": 96, "bs128": None}[args.workload]
    kbps = vq_kbps or KBPS
    block_switched = args.workload == "bs128"
    sample_rate = SAMPLE_RATE
    if block_switched:
        ex = np.load(os.path.join(ROOT, "tests", "golden", "excerpt_castanet.npz"))
        sample_rate = int(ex["sr"])
        reps = -(-n_frames * 1024 // len(ex["pcm"]))
        pcm = np.ascontiguousarray(np.tile(ex["pcm"], (reps, 1))[:n_frames * 1024])
    else:
        pcm = A.synth.stream(n_frames, N_CH, seed=A.synth.SEED + rank)
    enc = A.engine.Encoder(sample_rate, kbps / (sample_rate / 1000), use_vq=bool(vq_kbps),
                           use_sbr=bool(vq_kbps and vq_kbps < 128))
    planar = torch.as_tensor(A.synth.planar_with_halo(pcm), device=dev)
    view = A.engine.PcmView.stream(planar)
    hop_view = None
    if block_switched:      # the hops as the transient detector sees them (hop h = planar hop h+1)
        hop_view = A._lib.PacxPcm(planar.data_ptr() + 2 * 1024, A._lib.PCM_I16, N_CH, n_frames, 1024,
                                  planar.shape[1], 1)
        tr_buf = torch.empty(n_frames, dtype=torch.uint8, device=dev)
        fl_buf = torch.empty(n_frames + 2, dtype=torch.uint8, device=dev)
    n_cf = view.n_cf
    enc.reserve(n_cf)
    out = enc.alloc_outputs(n_cf, with_payload=True)
    total = torch.zeros(1, dtype=torch.int64, device=dev)
    gather = None
    if world > 1:
        # fixed-slot asynchronous gather of the packed bodies to rank 0 (RCCL): two send
        # buffers alternate, the gather of step i overlaps the encode of step i+1, no host
        # synchronisation and no size exchange inside a step
        on_host = dist.get_backend() == "gloo"          # rehearsal only
        gather = A.dist.BitstreamGather(A.dist.slot_bytes(n_cf, kbps / (sample_rate / 1000)),
                                        torch.device("cpu") if on_host else dev)
        bodies = [gather.body(0), gather.body(1)]
        if on_host:
            host_bodies, bodies = bodies, [torch.empty_like(b, device=dev) for b in bodies]
    else:
        bodies = [torch.empty(n_cf * 512, dtype=torch.uint8, device=dev)]
    cap = int(bodies[0].numel())
    step_no = [0]
    import ctypes
    from audio_codec_amd.engine import _ptr

    vq_out = None
    if vq_kbps:
        vq_out = {k: out[k] for k in ("overall", "bit_alloc", "status", "payload", "n_bytes")}

    def device_step():
        k = step_no[0] % len(bodies)
        if gather is not None:
            gather.wait(k)                      # the gather that last used this buffer (stream-level wait)
        if vq_kbps:
            enc.encode_vq(view, None, vq_out)
        elif block_switched:
            enc._call("pacx_transient_flags", ctypes.byref(hop_view), _ptr(tr_buf), _ptr(fl_buf), enc._stream())
            enc.encode_pack(view, fl_buf[:n_frames], out)
        else:
            enc.encode_pack(view, None, out)
        enc._call("pacx_gather_body", ctypes.c_int64(n_cf), _ptr(out["payload"]), _ptr(out["n_bytes"]),
                  _ptr(bodies[k]), ctypes.c_int64(cap), _ptr(total), enc._stream())
        if gather is not None:
            if dist.get_backend() == "gloo":            # rehearsal: stage through the host
                host_bodies[k].copy_(bodies[k])
                gather.launch(k, total.cpu())
            else:
                gather.launch(k, total)
        step_no[0] += 1

    # optional: the ~15 kernel launches of a step captured once into a hipGraph and replayed
    graph = None
    if args.graph and world == 1:
        device_step()
        torch.cuda.synchronize()
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                device_step()
            graph = g
        except Exception as e:                      # capture not possible: plain launches
            print(f"bench: hipGraph capture failed ({e}); launching kernels directly", file=sys.stderr)
            graph = None
        torch.cuda.synchronize()

    def step():
        if graph is not None:
            graph.replay()
        else:
            device_step()

    def sync_all():
        if gather is not None:
            for k in range(len(bodies)):
                gather.wait(k)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- MDCT kernel alone: HIP events on the stream the kernel runs on
    lines = torch.empty((n_cf, 1024), dtype=torch.float64, device=dev)
    scale = torch.empty((n_cf,), dtype=torch.int32, device=dev)

    def mdct_once():
        enc._call("pacx_mdct_batch", ctypes.byref(view.c), None, 0, _ptr(lines), _ptr(scale), enc._stream())
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
    mdct_ms = float(np.mean([a.elapsed_time(b) for a, b in ev])) / train
    mdct_gbs = n_cf * MDCT_BYTES_PER_CF / (mdct_ms * 1e-3) / 1e9

    # HBM bytes of the MDCT kernel from the PMC passes of this same command
    # (profiles/r01_mdct_pmc.json: separate FETCH_SIZE / WRITE_SIZE runs, gfx950
    # correction applied); only quoted when the launch geometry matches
    traffic = None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_mdct_pmc.json")))
        if pmc["cf_per_launch"] == n_cf:
            traffic = pmc["hbm_bytes_per_launch"]
    except Exception:
        pass

    if rank == 0:
        res = {
            "metric": "audio channel-frames/s encode (48 kHz, 1024-line long blocks, 128 kb/s/ch)"
                      if not (vq_kbps or block_switched)
                      else "audio channel-frames/s encode, block switching on (castanet excerpt tiled, 44.1 kHz, "
                           "128 kb/s/ch)" if block_switched
                      else f"audio channel-frames/s encode, gain-shape PVQ{' + SBR' if vq_kbps < 128 else ''} "
                           f"(48 kHz, 1024-line long blocks, {vq_kbps} kb/s/ch)",
            "value": world * n_cf * args.steps / dt,
            "unit": "channel-frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "castanet excerpt of tests/golden, tiled" if block_switched else "synthetic",
            "config": {"workload": (f"{n_frames} stereo frames per GPU of the castanet excerpt tiled "
                                    f"({n_cf} channel-frames, 44.1 kHz), long + short blocks by the transient "
                                    "detector (inside the step), " if block_switched else
                                    f"{n_frames} synthetic 48 kHz stereo frames per GPU ({n_cf} channel-frames), "
                                    "N=1024 long blocks, ") + f"{kbps} kb/s/ch, "
                                   f"{'gain-shape PVQ' + (' + SBR' if kbps < 128 else '') if vq_kbps else 'scalar mantissas'}, "
                                   "int16 PCM resident in HBM; step = encode + "
                                   ".pac bit packing + body assembly" +
                                   (" + asynchronous RCCL gather of the bodies to rank 0" if world > 1 else ""),
                       "stereo_frames_per_s": world * n_frames * args.steps / dt,
                       "launch": "hipGraph replay of the captured step" if graph is not None else "direct launches",
                       "sharding": f"{world} x frame-range shards, no data-path collective"},
            "roofline": {"kernel": "k_mdct_long_x2p (window + MDCT, int16 in, float64 lines out; two frames per wave alternating on one FFT tile, PCM prefetched a whole iteration ahead)",
                         "bound": "hbm", "achieved": mdct_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": mdct_gbs / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_note": "HBM bytes per launch, rocprofv3 PMC (2*FETCH_SIZE + WRITE_SIZE), "
                                         "profiles/r01_mdct_pmc.json",
                         "algorithmic_bytes_per_launch": n_cf * MDCT_BYTES_PER_CF,
                         "launch_ms": mdct_ms, "bytes_per_cf": MDCT_BYTES_PER_CF, "cf_per_launch": n_cf,
                         "mdct_cf_per_s": n_cf / (mdct_ms * 1e-3)},
        }
        if not args.no_cpu_baseline and world == 1:
            if block_switched:
                from oracle import pac_oracle as po
                n_hops_cpu = 192
                t0 = time.perf_counter()
                po.encode_stream(pcm[:n_hops_cpu * 1024], sample_rate, KBPS, block_switching=True)
                dtc = time.perf_counter() - t0
                res["cpu_baseline"] = {"value": (n_hops_cpu + 2) * N_CH / dtc, "unit": "channel-frames/s",
                                       "cores": 1, "kind": "port",
                                       "sample": f"first {n_hops_cpu} hops of the same tiled stream through "
                                                 f"oracle/pac_oracle.py encode_stream (block switching on), {dtc:.1f} s"}
            else:
                res["cpu_baseline"] = cpu_baseline(256, vq_kbps) if vq_kbps else cpu_baseline()
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
