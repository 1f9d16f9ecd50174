#!/bin/bash
# quick GPU round: tests, headline bench line, kernel stats and the step's PMC traffic
#   bash tools/r2_quick.sh <tag>   -> gpurun_out/<tag>/
TAG=${1:-r2q}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
if [ -z "$SKIP_TESTS" ]; then
  timeout -k 10 900 python3 -m pytest tests -m gpu -q -x > $OUT/tests.log 2>&1
  rc=$?
  tail -4 $OUT/tests.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
fi
timeout -k 10 300 python3 bench.py > $OUT/bench_scalar128.log 2>&1 || { tail -20 $OUT/bench_scalar128.log; exit 1; }
tail -1 $OUT/bench_scalar128.log > $OUT/bench_scalar128.json
python3 -c "import json,sys; d=json.load(open('$OUT/bench_scalar128.json')); print('scalar128', d['value'], d['ms_per_step'], d['verified_cf'], d['roofline']['frac'], d.get('cpu_baseline',{}).get('value'))"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o s --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-verify > /dev/null 2>&1 || exit 1
cp $OUT/stats/*/s_kernel_stats.csv $OUT/scalar128_kernel_stats.csv 2>/dev/null || cp $OUT/stats/s_kernel_stats.csv $OUT/scalar128_kernel_stats.csv
head -12 $OUT/scalar128_kernel_stats.csv | cut -c1-150
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 10 --warmup 2 --repeats 1 --no-cpu-baseline --no-verify > /dev/null 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 10 --warmup 2 --repeats 1 --no-cpu-baseline --no-verify > /dev/null 2>&1 || exit 1
python3 $R/tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write $OUT/mdct_pmc.json $OUT/step_traffic.json
rm -rf $OUT/stats $OUT/pmc_fetch $OUT/pmc_write
