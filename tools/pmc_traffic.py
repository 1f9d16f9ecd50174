"""HBM traffic of the bench step from two rocprofv3 --pmc passes of bench.py:
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d A -- python3 bench.py --steps 10 --warmup 2 --repeats 1 --no-cpu-baseline --no-verify
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d B -- python3 bench.py --steps 10 --warmup 2 --repeats 1 --no-cpu-baseline --no-verify
    python tools/pmc_traffic.py A B profiles/rNN_mdct_pmc.json [profiles/rNN_step_traffic.json]
FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 wide coalesced reads are counted as
64-byte requests although 128 bytes move (MI355X_MICROARCH.md, HBM section):
read bytes = 2 x FETCH_SIZE.

The second output lists, for every kernel of one step, the PMC bytes per launch next to the
ALGORITHMIC bytes of that kernel's interface (what it must read and write once), so that
re-reads and spills show as a ratio above 1."""
import collections, csv, glob, json, re, sys

N_CF = 8192


def load(d, counter):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            name = re.sub(r"\(.*", "", r["Kernel_Name"]).strip()
            acc[name].append(float(r["Counter_Value"]))
    return acc


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
per = {}
for k in fetch:
    per[k] = {"FETCH_SIZE": {"launches": len(fetch[k]), "mean": sum(fetch[k]) / len(fetch[k])},
              "WRITE_SIZE": {"launches": len(write.get(k, [])),
                             "mean": sum(write.get(k, [0])) / max(1, len(write.get(k, [])))}}
cands = [k for k in per if "k_mdct_long_x2" in k or "k_mdct_long_v2" in k]
mdct = ([k for k in cands if "false" in k] or cands)[0]       # the stand-alone instantiation (the bench's roofline trains) where both ran
f_kb, w_kb = per[mdct]["FETCH_SIZE"]["mean"], per[mdct]["WRITE_SIZE"]["mean"]
out = {
    "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 bench.py --steps 10 --warmup 2 --repeats 1 --no-cpu-baseline --no-verify",
    "kernel": mdct, "cf_per_launch": N_CF,
    "FETCH_SIZE_KB_raw": f_kb, "WRITE_SIZE_KB": w_kb,
    "correction": "gfx950: FETCH_SIZE counts 128-B requests as 64 B for wide coalesced reads (MI355X_MICROARCH.md, HBM): read bytes = 2 x FETCH_SIZE; WRITE_SIZE exact for 16-B/lane stores",
    "hbm_bytes_per_launch": (2 * f_kb + w_kb) * 1024,
    "algorithmic_bytes_per_launch": N_CF * 10240,
    "per_kernel": per,
}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(mdct, "HBM bytes/launch", out["hbm_bytes_per_launch"], "algorithmic", out["algorithmic_bytes_per_launch"])

if len(sys.argv) > 4:
    def pmc_bytes(k):
        return (2 * per[k]["FETCH_SIZE"]["mean"] + per[k]["WRITE_SIZE"]["mean"]) * 1024

    def find(tag):
        hits = [k for k in per if tag in k]
        return hits[0] if hits else None
    side = find("k_side_long")
    peaks = per[side]["WRITE_SIZE"]["mean"] * 1024 if side else 0.0     # the masker lists: written once, read once
    codes = N_CF * (17 * 4 * 2 + 32 + 4 + 4)                             # scale factors, allocation, overall[8], status, n_bytes
    payload = None
    g = find("k_gather_small") or find("k_copy_body")
    if g:
        payload = per[g]["WRITE_SIZE"]["mean"] * 1024                    # the .pac body: what the step is for
    algo = {
        "k_mdct_long_x2": ("int16 hop in, float64 lines out", N_CF * 10240),
        "k_side_long": ("int16 hop in, kept maskers out", N_CF * 2048 + peaks),
        "k_mask_tail": ("lines + maskers in, codes + payload out (mask, BitAlloc, quantise, pack fused)",
                        N_CF * 8192 + peaks + codes + (payload or 0)),
        "k_mask": ("lines + maskers in, SMR out", N_CF * 8192 + peaks + N_CF * 17 * 8),
        "k_tail_long": ("lines + SMR in, codes + payload out", N_CF * 8192 + N_CF * 17 * 8 + codes + (payload or 0)),
        "k_gather_small": ("payload slots in, .pac body out", 2 * (payload or 0) + N_CF * 4),
    }
    rows, tot_pmc, tot_algo = [], 0.0, 0.0
    for k in sorted(per):
        if not k.split("<")[0].replace("void ", "").startswith("k_"):
            continue
        tag = next((t for t in sorted(algo, key=len, reverse=True) if t in k), None)
        b = pmc_bytes(k)
        a = algo[tag][1] if tag else None
        rows.append({"kernel": k, "launches_profiled": per[k]["FETCH_SIZE"]["launches"],
                     "pmc_bytes_per_launch": b, "read_bytes": 2 * per[k]["FETCH_SIZE"]["mean"] * 1024,
                     "write_bytes": per[k]["WRITE_SIZE"]["mean"] * 1024,
                     "algorithmic_bytes_per_launch": a, "what": algo[tag][0] if tag else None,
                     "ratio": (b / a) if a else None})
        if "k_mdct_long_x2p<8, 2, false>" in k:
            continue                       # the stand-alone MDCT trains of the roofline measurement are not part of the step
        tot_pmc += b
        tot_algo += a or 0.0
    step = {"command": out["command"], "cf_per_step": N_CF,
            "correction": out["correction"],
            "kernels": rows,
            "step_pmc_bytes": tot_pmc, "step_algorithmic_bytes": tot_algo,
            "compulsory_bytes": N_CF * 2048 + (payload or 0),
            "compulsory_note": "what a single fused kernel would have to move: the int16 PCM once (2048 B per channel-frame) and the .pac body",
            "pmc_over_compulsory": tot_pmc / (N_CF * 2048 + (payload or 1))}
    json.dump(step, open(sys.argv[4], "w"), indent=1)
    print("step: PMC %.1f MB, algorithmic %.1f MB, compulsory %.1f MB" %
          (tot_pmc / 1e6, tot_algo / 1e6, step["compulsory_bytes"] / 1e6))
