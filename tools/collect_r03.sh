#!/bin/bash
# Round-3 artefacts kept under profiles/ (run on the GPU box from the repo root): bash tools/collect_r03.sh
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# bench lines of the default command per workload; kernel-trace stats of the same command (two steps in flight)
# and with one step in flight and direct launches (what a kernel takes when the step runs by itself)
for w in scalar128 bs128 vq128 vq96 shipped128 shipped96; do
  extra="--workload $w"; [ $w = scalar128 ] || extra="$extra --host-stream-frames 0"
  python3 $R/bench.py $extra > $OUT/bench_$w.log 2>&1
  tail -1 $OUT/bench_$w.log > $OUT/bench_$w.json
  rocprofv3 --kernel-trace --stats -d $OUT/stats_$w -o s --output-format csv -- python3 $R/bench.py --workload $w --no-cpu-baseline --no-verify --no-decode-leg --host-stream-frames 0 --min-seconds 0 > /dev/null 2>&1
  cp $OUT/stats_$w/s_kernel_stats.csv $OUT/${w}_kernel_stats.csv
  rocprofv3 --kernel-trace --stats -d $OUT/stats1_$w -o s --output-format csv -- python3 $R/bench.py --workload $w --no-cpu-baseline --no-verify --no-decode-leg --host-stream-frames 0 --min-seconds 0 --pipeline 1 --no-graph > /dev/null 2>&1
  cp $OUT/stats1_$w/s_kernel_stats.csv $OUT/${w}_kernel_stats_one_in_flight.csv
  echo "done $w"
done
# HBM bytes (separate FETCH_SIZE / WRITE_SIZE passes; the profiler serialises the kernels)
A="--min-seconds 0 --steps 10 --warmup 2 --repeats 1 --no-cpu-baseline --no-verify --no-decode-leg --pipeline 1 --no-graph --host-stream-frames 0"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py $A > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py $A > /dev/null 2>&1
python3 $R/tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write $OUT/mdct_pmc.json $OUT/step_traffic.json
echo "done pmc traffic"
cd $R
for w in scalar128 bs128 vq128 shipped128; do
  rm -rf gpurun_out/pmcs
  PMC_BENCH_ARGS="--workload $w" bash tools/pmc_step.sh && python3 tools/pmc_step_summary.py > $OUT/sq_counters_$w.txt
  echo "done sq $w"
done
rm -rf gpurun_out/pmcs gpurun_out/pmcs_p*.log
python3 tools/step_kernels.py $OUT $OUT/step_kernels.json
# large batches
python3 bench.py --frames 131072 --steps 5 --warmup 2 --repeats 5 --no-cpu-baseline --host-stream-frames 0 > $OUT/bench_scalar128_262144.log 2>&1 && tail -1 $OUT/bench_scalar128_262144.log > $OUT/bench_scalar128_262144.json
python3 bench.py --workload bs128 --frames 862000 --steps 5 --warmup 2 --repeats 5 --no-cpu-baseline --host-stream-frames 0 > $OUT/bench_bs128_x1000.log 2>&1 && tail -1 $OUT/bench_bs128_x1000.log > $OUT/bench_bs128_x1000.json
python3 bench.py --corpus --corpus-frames 131072 --steps 5 --warmup 2 --repeats 5 --no-cpu-baseline > $OUT/bench_corpus_eighth.log 2>&1 && tail -1 $OUT/bench_corpus_eighth.log > $OUT/bench_corpus_eighth.json
python3 tools/mdct_sweep.py 8192 16384 65536 262144 > $OUT/mdct_sweep.txt 2>&1
python3 tools/decode_probe.py 2>&1 | grep -v amdgpu > $OUT/decode_probe.txt
python3 tools/overlap2_probe.py 4096 2>&1 | grep -v amdgpu > $OUT/overlap2_probe.txt
rm -rf $OUT/stats_* $OUT/stats1_* $OUT/pmc_fetch $OUT/pmc_write $OUT/*.log
ls $OUT
