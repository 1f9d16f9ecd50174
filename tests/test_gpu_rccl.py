"""The RCCL side of the multi-GPU path on the one GPU a test box has: BitstreamGather (what
`bench.py --gpus N` uses to bring every rank's packed body to rank 0) over the 'nccl' backend with a
single rank -- device-resident slots, asynchronous gather, header + body round trip.  The world_size-2
control flow is covered on CPU over gloo (tests/test_dist_gloo.py); this run shows that the very calls
the bench makes (init_process_group("nccl", device_id=...), dist.gather(async_op=True) on device
tensors, wait, unpack) execute on ROCm.  Runs in a child process with a timeout so that a collective
library that cannot initialise on a box ends in a skip, not in a hung suite."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import numpy as np
    import torch
    import torch.distributed as dist
    import audio_codec_amd as A
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    try:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    except Exception as e:                                       # noqa: BLE001
        print("INIT_FAILED", repr(e))
        sys.exit(3)
    enc = A.engine.Encoder(48000, 128 / 48.0)
    pcm = A.synth.stream(64, 2)
    planar = torch.as_tensor(A.synth.planar_with_halo(pcm), device=dev)
    view = A.engine.PcmView.stream(planar)
    out = enc.encode_pack(view)
    slot = A.dist.slot_bytes(view.n_cf, 128 / 48.0)
    g = A.dist.BitstreamGather(slot, dev)
    want, total = enc.gather_body(out["payload"], out["n_bytes"])
    n = int(total.item())
    for k in (0, 1, 0):                                          # both send buffers, one reused
        body, tot = enc.gather_body(out["payload"], out["n_bytes"], out=g.body(k))
        g.launch(k, tot)
        g.wait(k)
        g.check(k)
        got = g.unpack(k)
        if got.numel() != n or not torch.equal(got, want[:n]):
            print("MISMATCH", k, got.numel(), n)
            sys.exit(4)
    dist.barrier()
    dist.destroy_process_group()
    print("RCCL_GATHER_OK", n)
""") % ROOT


def test_bitstream_gather_over_rccl_single_rank():
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, timeout=240, env=env)
    except subprocess.TimeoutExpired:
        pytest.skip("RCCL single-rank run did not finish within 240 s on this box")
    if r.returncode == 3 or "INIT_FAILED" in r.stdout:
        pytest.skip("RCCL could not initialise on this box: " + r.stdout.strip()[-300:])
    assert r.returncode == 0, (r.stdout[-800:], r.stderr[-1500:])
    assert "RCCL_GATHER_OK" in r.stdout


def test_bench_multi_gpu_flow_over_rccl_single_rank():
    """bench.py launched the way the driver launches it for N > 1 (torch.distributed.run, one rank per
    GPU, RCCL) with N = 1 and PACX_BENCH_FORCE_DIST=1: process group over 'nccl', barrier-bracketed timed
    regions, max over ranks, the asynchronous fixed-slot gather of the bodies overlapping the next step,
    slot check on the sending rank, oracle check of the timed run's output."""
    import json
    env = dict(os.environ)
    env["PACX_BENCH_FORCE_DIST"] = "1"
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", "29543", os.path.join(ROOT, "bench.py"),
           "--gpus", "1", "--steps", "4", "--warmup", "2", "--repeats", "2", "--frames", "512",
           "--no-cpu-baseline"]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=400, env=env, cwd=ROOT)
    except subprocess.TimeoutExpired:
        pytest.skip("single-rank RCCL bench did not finish within 400 s on this box")
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert r.returncode == 0 and lines, (r.stdout[-800:], r.stderr[-1500:])
    d = json.loads(lines[-1])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["verified_cf"] > 0
    assert "RCCL gather" in d["config"]["workload"]
