"""`python bench.py --gpus N` must start its own N ranks (the reference starts its workers from one
command too, coder/pacfile.py:771-781).  CPU tests of the launcher alone: the ranks join a gloo group and rank 0
prints one line (--launcher-selftest: no encode -- the encode path has no CPU implementation), and a launch whose
ranks fail (no GPU here) ends with a non-zero exit code and no JSON line, not with a hang or a made-up value."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "PACX_BENCH_FORCE_DIST"):
        env.pop(k, None)
    return env


@pytest.mark.timeout(300)
def test_plain_command_starts_two_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--frames", "64", "--steps", "2", "--no-cpu-baseline",
                        "--launcher-selftest"], capture_output=True, text=True, timeout=280, env=_env(), cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-1500:])
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                       # ONE line on stdout: rank 0's
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == [0, 1] and d["steps"] == 2


@pytest.mark.timeout(300)
def test_failing_ranks_give_nonzero_exit_and_no_line():
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU: here the ranks would run the bench")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--frames", "64", "--steps", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=280, env=_env(), cwd=ROOT)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]


def test_parent_never_imports_torch_before_launching():
    """the launching process must not initialise the GPU: the launch decision sits before `import torch`"""
    import ast
    src = open(BENCH).read()
    tree = ast.parse(src)
    main = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "main")
    launch_line = next(n.lineno for n in ast.walk(main) if isinstance(n, ast.Call)
                       and getattr(n.func, "id", "") == "self_launch")
    torch_line = min(n.lineno for n in ast.walk(main) if isinstance(n, ast.Import)
                     and any(a.name.startswith("torch") for a in n.names))
    assert launch_line < torch_line
    top = [a.name for n in tree.body if isinstance(n, ast.Import) for a in n.names]
    assert not any(t.startswith("torch") for t in top)
    launcher = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "self_launch")
    assert not any(isinstance(n, (ast.Import, ast.ImportFrom)) and "torch" in ast.dump(n) for n in ast.walk(launcher))
