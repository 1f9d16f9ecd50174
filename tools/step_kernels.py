#!/usr/bin/env python3
"""profiles/rNN_step_kernels.json: per kernel of the step -- microseconds (rocprofv3 --kernel-trace --stats of the bench
command with one step in flight and with the default two), VALU-busy fraction (SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE
/ 8 x 1024 SIMDs), separate --pmc passes, kernels serialised by the profiler) and PMC bytes per launch against the
kernel's algorithmic bytes.  bench.py quotes it as config.step_kernels.
    python tools/step_kernels.py <dir with the collected files> <out.json>"""
import csv, json, os, re, sys

d, out_path = sys.argv[1], sys.argv[2]


def stats(path):
    out = {}
    if not os.path.exists(path):
        return out
    for r in csv.DictReader(open(path)):
        name = re.sub(r"\(.*", "", r["Name"]).strip()
        if name.startswith(("k_", "void k_")):
            out[name.replace("void ", "")] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3}
    return out


def counters(path):
    out, cur = {}, None
    if not os.path.exists(path):
        return out
    for line in open(path):
        if not line.startswith(" "):
            cur = line.strip().replace("void ", "")
            out[cur] = {}
        else:
            for m in re.finditer(r"(\w+)=([0-9.e+]+)", line):
                out[cur][m.group(1)] = float(m.group(2))
    return out


res = {}
for wl, cf in (("scalar128", 8192), ("bs128", 8192), ("vq128", 8192), ("shipped128", 8192)):
    alone = stats(os.path.join(d, f"{wl}_kernel_stats_one_in_flight.csv"))
    piped = stats(os.path.join(d, f"{wl}_kernel_stats.csv"))
    sq = counters(os.path.join(d, f"sq_counters_{wl}.txt"))
    traffic = {}
    tp = os.path.join(d, "step_traffic.json")
    if wl == "scalar128" and os.path.exists(tp):
        for row in json.load(open(tp)).get("kernels", []):
            traffic[row["kernel"].replace("void ", "")] = row
    kernels = {}
    for name, a in sorted(alone.items(), key=lambda kv: -kv[1]["avg_us"]):
        if a["calls"] < 20:
            continue
        e = {"us_one_step_in_flight": round(a["avg_us"], 2)}
        if name in piped:
            e["us_two_steps_in_flight"] = round(piped[name]["avg_us"], 2)
        c = next((v for k, v in sq.items() if name.startswith(k) or k.startswith(name[:40])), None)
        if c and c.get("GRBM_GUI_ACTIVE") and c.get("SQ_ACTIVE_INST_VALU"):
            e["valu_busy"] = round(c["SQ_ACTIVE_INST_VALU"] * 4 / (c["GRBM_GUI_ACTIVE"] / 8 * 1024), 3)
            if c.get("SQ_WAVE_CYCLES"):
                e["wait_any_of_wave_cycles"] = round(c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"], 3)
        t = next((v for k, v in traffic.items() if name.startswith(k[:30])), None)
        if t:
            e.update({kk: t[kk] for kk in t if kk in ("pmc_bytes_per_launch", "algorithmic_bytes_per_launch")})
        kernels[name] = e
    if kernels:
        res[wl] = {"cf_per_step": cf, "kernels": kernels}
json.dump(res, open(out_path, "w"), indent=1)
print(out_path, {k: len(v["kernels"]) for k, v in res.items()})
