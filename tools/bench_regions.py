#!/usr/bin/env python3
"""Spread of the K-step regions of one bench.py run (two steps in flight settle into one of two phase relations
between the pipelines; this shows which, region by region):  python tools/bench_regions.py label -- <bench args>"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
i = sys.argv.index("--")
label, args = " ".join(sys.argv[1:i]), sys.argv[i + 1:]
r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-verify"] + args,
                   capture_output=True, text=True)
lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
if r.returncode or not lines:
    print(label, "FAILED", r.stderr[-600:])
    sys.exit(1)
d = json.loads(lines[-1])
reg = np.array(d["config"]["ms_per_step_regions"])
q = np.percentile(reg, [0, 10, 50, 90, 100])
hist, edges = np.histogram(reg, bins=8)
print(f"{label:24s} {d['value'] / 1e6:7.2f} M  regions {len(reg)}  min/p10/med/p90/max "
      + " ".join(f"{x:.4f}" for x in q) + "  hist " + " ".join(f"{e:.3f}:{h}" for h, e in zip(hist, edges)), flush=True)
step = max(len(reg) // 24, 1)
print("   every %d-th region: " % step + " ".join(f"{x:.3f}" for x in reg[::step]))
