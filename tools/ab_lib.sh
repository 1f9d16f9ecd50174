# same-box A/B of the default library against variant libraries: bash tools/ab_lib.sh <workload> <kernel-prefix> <variant> [<variant> ...]
mkdir -p /root/repo/gpurun_out/ab; cd /tmp && export TMPDIR=/tmp
W=$1; K=$2; shift 2
for rep in 1 2; do
for v in "" "$@"; do
  if [ -n "$v" ]; then export PACX_LIB=/root/repo/audio-codec_amd/variants/libpacx_$v.so; else unset PACX_LIB; fi
  d=/root/repo/gpurun_out/ab/${v:-base}_$rep
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $d -o s --output-format csv -- python3 /root/repo/bench.py --workload $W --no-cpu-baseline --no-verify --no-decode-leg --steps 10 --repeats 3 > $d.log 2>&1 || exit 1
  python3 - <<PY
import csv,glob
f=glob.glob("$d/**/s_kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if r["Name"].startswith("$K"): print("${v:-base} rep $rep:", r["Name"][:24], round(float(r["AverageNs"])/1e3,1), "us")
PY
done; done
