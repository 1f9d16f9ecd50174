#!/bin/bash
# kernel trace of the default bench (two steps in flight) for tools/phase_trace.py:  bash tools/phase_trace.sh <tag> [bench args]
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $OUT/tr -o t --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-verify --no-decode-leg --min-seconds 0.25 --host-stream-frames 0 "$@" > $OUT/bench.log 2>&1 || { tail -5 $OUT/bench.log; exit 1; }
f=$(find $OUT/tr -name 't_kernel_trace.csv' | head -1)
python3 $R/tools/phase_trace.py $f > $OUT/phase.txt; head -30 $OUT/phase.txt
gzip -c $f > $OUT/kernel_trace.csv.gz; rm -rf $OUT/tr
