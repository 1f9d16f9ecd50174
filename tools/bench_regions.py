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
dump = "/tmp/pacx_regions_%d.txt" % os.getpid()
r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-verify", "--no-decode-leg"] + args,
                   capture_output=True, text=True, env=dict(os.environ, PACX_BENCH_DUMP_REGIONS=dump))
lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
if r.returncode or not lines:
    print(label, "FAILED", r.stderr[-600:])
    sys.exit(1)
d = json.loads(lines[-1])
reg = np.loadtxt(dump)
os.remove(dump)
q = np.percentile(reg, [0, 10, 50, 90, 100])
hist, edges = np.histogram(reg, bins=8)
print(f"{label:24s} {d['value'] / 1e6:7.2f} M  regions {len(reg)}  min/p10/med/p90/max "
      + " ".join(f"{x:.4f}" for x in q) + "  hist " + " ".join(f"{e:.3f}:{h}" for h, e in zip(hist, edges)), flush=True)
fast = reg < 0.5 * (q[0] + q[4] if q[4] < 1.3 * q[0] else 2.15 * q[0])
runs, cur = [], 1
for a, b in zip(fast[:-1], fast[1:]):
    if a == b:
        cur += 1
    else:
        runs.append(cur)
        cur = 1
runs.append(cur)
print("   fast regions %.0f %%; lengths of runs of equal kind: %s" % (100.0 * fast.mean(), " ".join(map(str, runs[:60]))))
