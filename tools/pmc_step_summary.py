#!/usr/bin/env python3
"""Per-kernel means of the counters collected by tools/pmc_step.sh."""
import collections, csv, glob, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/pmcs/*/out_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if not any(t in k for t in ("k_mask", "k_side", "k_mdct", "k_tail", "k_vq")):
        continue
    m = {c: sum(v) / len(v) for c, v in d.items()}
    print(k)
    print("  " + "  ".join("%s=%.3g" % (c, v) for c, v in sorted(m.items())))
    wc = m.get("SQ_WAVE_CYCLES")
    if wc:
        for c in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS"):
            if c in m:
                print("    %-22s %5.1f %% of wave cycles" % (c, 100.0 * m[c] / wc))
