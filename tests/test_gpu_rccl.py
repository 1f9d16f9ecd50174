"""The RCCL side of the multi-GPU path on the one GPU a test box has: BitstreamGather (what
`bench.py --gpus N` uses to bring every rank's packed body to rank 0) over the 'nccl' backend with a
single rank -- device-resident slots, asynchronous gather, header + body round trip.  The world_size-2
control flow is covered on CPU over gloo (tests/test_dist_gloo.py); this run shows that the very calls
the bench makes (init_process_group("nccl", device_id=...), dist.gather(async_op=True) on device
tensors, wait, unpack) execute on ROCm.  Runs in a fresh child process with a timeout.  A child that does
not finish is a FAILURE with its output and the last progress marker it printed attached (a hung
collective or kernel is a defect to diagnose from that record); the only skip is the explicit
INIT_FAILED path, RCCL refusing to initialise on the box."""
import os
import socket
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
    def mark(what):
        print("STEP", what, flush=True)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    mark("init_process_group")
    try:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    except Exception as e:                                       # noqa: BLE001
        print("INIT_FAILED", repr(e))
        sys.exit(3)
    mark("encoder")
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
        mark(f"gather buffer {k}")
        body, tot = enc.gather_body(out["payload"], out["n_bytes"], out=g.body(k))
        g.launch(k, tot)
        g.wait(k)
        g.check(k)
        got = g.unpack(k)
        if got.numel() != n or not torch.equal(got, want[:n]):
            print("MISMATCH", k, got.numel(), n)
            sys.exit(4)
    mark("barrier")
    dist.barrier()
    dist.destroy_process_group()
    print("RCCL_GATHER_OK", n)
""") % ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(cmd, env, timeout, cwd=None):
    """one fresh child process; a timeout is a failure that carries what the child had printed"""
    try:
        return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=cwd)
    except subprocess.TimeoutExpired as e:
        out = e.stdout.decode(errors="replace") if isinstance(e.stdout, bytes) else (e.stdout or "")
        err = e.stderr.decode(errors="replace") if isinstance(e.stderr, bytes) else (e.stderr or "")
        steps = [ln for ln in out.splitlines() if ln.startswith("STEP")]
        pytest.fail(f"child did not finish within {timeout} s; last progress marker: "
                    f"{steps[-1] if steps else 'none'}\n--- stdout tail ---\n{out[-1500:]}\n--- stderr tail ---\n{err[-2500:]}")


def test_bitstream_gather_over_rccl_single_rank():
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_PORT"] = str(_free_port())
    r = _run([sys.executable, "-c", CHILD], env, 240)
    if r.returncode == 3 or "INIT_FAILED" in r.stdout:
        pytest.skip("RCCL could not initialise on this box: " + r.stdout.strip()[-300:])
    assert r.returncode == 0, (r.stdout[-800:], r.stderr[-1500:])
    assert "RCCL_GATHER_OK" in r.stdout


def _bench_line(r):
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-800:], r.stderr[-2500:])
    import json
    return json.loads(lines[0])


def test_bench_multi_gpu_flow_over_rccl_single_rank():
    """The PLAIN command (`python bench.py --gpus 1`, no torch.distributed.run in front) with
    PACX_BENCH_FORCE_DIST=1: bench.py starts its own rank(s) as child processes; inside, the N > 1 flow:
    process group over 'nccl', barrier-bracketed timed regions, max over ranks, two steps in flight
    with the asynchronous fixed-slot gather of the bodies overlapping the next step, slot check on the sending
    rank, oracle check of the timed run's output on every rank."""
    env = dict(os.environ)
    env["PACX_BENCH_FORCE_DIST"] = "1"
    env.pop("WORLD_SIZE", None)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2",
           "--repeats", "2", "--min-seconds", "0", "--frames", "512", "--no-cpu-baseline"]
    d = _bench_line(_run(cmd, env, 400, cwd=ROOT))
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["verified_cf"] > 0
    assert "RCCL gather" in d["config"]["workload"]
    assert "nccl with 1 ranks" in d["config"]["sharding"]
    assert d["config"]["steps_in_flight"] == 2          # the gather's two send buffers = two steps in flight


def test_bench_plain_command_starts_two_ranks():
    """`python bench.py --gpus 2` as the driver types it.  A test box has one GPU, so PACX_BENCH_ONE_GPU=1 puts
    both ranks on cuda:0 and the gather on gloo (RCCL refuses two ranks on one device); everything else is the
    N = 2 path: self-launch, two processes, each its own stream, barriers, max over ranks, every rank's output
    checked against the oracle and the counts gathered to rank 0."""
    env = dict(os.environ)
    env["PACX_BENCH_ONE_GPU"] = "1"
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "PACX_BENCH_FORCE_DIST"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--repeats", "2", "--min-seconds", "0", "--frames", "256", "--no-cpu-baseline"]
    d = _bench_line(_run(cmd, env, 400, cwd=ROOT))
    assert d["n_gpus"] == 2 and d["value"] > 0
    assert len(d["config"]["verified_per_rank"]) == 2 and min(d["config"]["verified_per_rank"]) > 0
    assert d["verified_cf"] == sum(d["config"]["verified_per_rank"])
    assert "gloo with 2 ranks" in d["config"]["sharding"]
