"""profiles/r01_mdct_pmc.json from two rocprofv3 --pmc passes of bench.py:
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d A -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d B -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline
    python tools/pmc_traffic.py A B profiles/r01_mdct_pmc.json
FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 wide coalesced reads are counted as
64-byte requests although 128 bytes move (MI355X_MICROARCH.md, HBM section):
read bytes = 2 x FETCH_SIZE."""
import collections, csv, glob, json, re, sys

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
mdct = [k for k in per if "k_mdct_long_x2" in k or "k_mdct_long_v2" in k][0]
f_kb, w_kb = per[mdct]["FETCH_SIZE"]["mean"], per[mdct]["WRITE_SIZE"]["mean"]
out = {
    "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline",
    "kernel": mdct, "cf_per_launch": 8192,
    "FETCH_SIZE_KB_raw": f_kb, "WRITE_SIZE_KB": w_kb,
    "correction": "gfx950: FETCH_SIZE counts 128-B requests as 64 B for wide coalesced reads (MI355X_MICROARCH.md, HBM): read bytes = 2 x FETCH_SIZE; WRITE_SIZE exact for 16-B/lane stores",
    "hbm_bytes_per_launch": (2 * f_kb + w_kb) * 1024,
    "algorithmic_bytes_per_launch": 8192 * 10240,
    "per_kernel": per,
}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(mdct, "HBM bytes/launch", out["hbm_bytes_per_launch"], "algorithmic", out["algorithmic_bytes_per_launch"])
