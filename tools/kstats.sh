#!/bin/bash
# kernel-stats summary of one bench workload:  bash tools/kstats.sh <tag> <bench args...>  -> gpurun_out/<tag>/
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/st -o s --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-verify --no-decode-leg "$@" > $OUT/bench.log 2>&1 || { tail -5 $OUT/bench.log; exit 1; }
grep '^{"metric' $OUT/bench.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$TAG', round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],4), 'ms')"
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/st/**/s_kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(r["Name"][:44].ljust(44), r["Calls"].rjust(5), round(float(r["AverageNs"])/1e3,1))
PY
cp $(find $OUT/st -name s_kernel_stats.csv | head -1) $OUT/kernel_stats.csv; rm -rf $OUT/st
