#!/usr/bin/env python3
"""One line per bench run: workload, M cf/s, ms/step (helper for A/B runs on the GPU box):
    python tools/bench_value.py label -- <bench.py args>"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
i = sys.argv.index("--")
label, args = " ".join(sys.argv[1:i]), sys.argv[i + 1:]
r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-decode-leg", "--min-seconds", "0.4"] + args, capture_output=True, text=True)
lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
if r.returncode or not lines:
    print(label, "FAILED", r.stderr[-600:])
    sys.exit(1)
d = json.loads(lines[-1])
c = d["config"]
print(f"{label:36s} {d['value'] / 1e6:8.2f} M cf/s  {d['ms_per_step']:.4f} ms/step  verified {d['verified_cf']}  "
      f"[{c['launch'][:8]} x{c.get('steps_in_flight', 1)}; one in flight {c.get('value_one_step_in_flight', 0) / 1e6:.2f} M]  "
      f"mdct {d['roofline']['launch_ms'] * 1e3:.1f} us  host {c.get('host_enqueue_ms_per_step') or 0:.4f} ms/step", flush=True)
