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

# windows of 24 consecutive mask kernels that hold both pipelines: the fastest and the slowest, every kernel with its queue
def timeline(w, label):
    t0, t1 = w[8][0], w[12][0]
    print("--- %s: kernels starting in four steps from the middle of the window (offset us, duration us, stream, queue, name)" % label)
    for x in k:
        if t0 - 5_000 <= x[0] < t1 + 5_000:
            print("  %8.1f %7.1f  s%-3s q%-3s %s" % ((x[0] - t0) / 1e3, (x[1] - x[0]) / 1e3, x[4], x[3], x[2][:36]))


wins = []
for i in range(0, len(masks) - 24, 6):
    w = masks[i:i + 24]
    if len({m[4] for m in w}) == 2 and max(b[0] - a[0] for a, b in zip(w[:-1], w[1:])) < 400_000:
        wins.append(((w[-1][0] - w[0][0]) / 23.0, i))
wins.sort()
if wins:
    print("windows: %d, us/step fastest %.1f median %.1f slowest %.1f" % (len(wins), wins[0][0] / 1e3, wins[len(wins) // 2][0] / 1e3, wins[-1][0] / 1e3))
    timeline(masks[wins[0][1]:wins[0][1] + 24], "fastest window (%.1f us/step)" % (wins[0][0] / 1e3))
    timeline(masks[wins[-1][1]:wins[-1][1] + 24], "slowest window (%.1f us/step)" % (wins[-1][0] / 1e3))
