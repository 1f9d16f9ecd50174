#!/usr/bin/env python3
"""From a rocprofv3 kernel trace of bench.py with two steps in flight: for every train of mask kernels (a timed region), the
time per step and how the two pipelines' mask kernels lie relative to one another (offset of B's start inside A's period)."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
name_key = "Kernel_Name" if "Kernel_Name" in rows[0] else "Name"
k = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r[name_key], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")) for r in rows]
k.sort()
masks = [x for x in k if x[2].startswith("void k_mask<1024")]
print("mask kernels:", len(masks), "queues:", sorted({m[3] for m in masks}), "streams:", sorted({m[4] for m in masks}))
# split into trains: a gap of > 1 ms between consecutive mask starts ends a region
trains, cur = [], [masks[0]]
for m in masks[1:]:
    if m[0] - cur[-1][0] > 1_000_000:
        trains.append(cur)
        cur = []
    cur.append(m)
trains.append(cur)
trains = [t for t in trains if len(t) >= 12]
print("regions:", len(trains))
out = []
for t in trains:
    by = defaultdict(list)
    for m in t:
        by[m[4] if m[4] != "?" else m[3]].append(m)
    ids = sorted(by)
    if len(ids) != 2:
        continue
    a, b = by[ids[0]], by[ids[1]]
    per_step = (t[-1][1] - t[0][0]) / len(t) / 1e3
    pa = (a[-1][0] - a[0][0]) / max(len(a) - 1, 1)
    offs = []
    for mb in b[2:-2]:
        prev = [ma for ma in a if ma[0] <= mb[0]]
        if prev:
            offs.append(((mb[0] - prev[-1][0]) % pa) / pa)
    dur = sum(m[1] - m[0] for m in t) / len(t) / 1e3
    # concurrency: fraction of a's mask time during which a b mask runs
    ov = 0
    for ma in a:
        for mb in b:
            ov += max(0, min(ma[1], mb[1]) - max(ma[0], mb[0]))
    out.append((per_step, sum(offs) / max(len(offs), 1), dur, ov / max(sum(m[1] - m[0] for m in a), 1)))
out.sort()
print("us/step  mean phase of B in A's period  mean mask duration us  share of A's mask time overlapped by a B mask")
for o in out[:: max(len(out) // 30, 1)]:
    print("%7.1f  %5.2f  %6.1f  %5.2f" % o)
