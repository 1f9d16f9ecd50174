#!/bin/bash
# Collects the artefacts kept under profiles/ (run on the GPU box from the repo root):
#   bash tools/collect_profiles.sh <tag>      -> gpurun_out/<tag>/...
# bench lines and rocprofv3 --kernel-trace --stats summaries of the four workloads, the
# FETCH_SIZE / WRITE_SIZE passes of the MDCT kernel, SQ counters of the step's kernels,
# in-kernel phase stamps (needs libpacx_dbg.so from `build.py --phase-debug`) and the
# traffic-mix ceiling probe.
set -e
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for w in scalar128 vq128 vq96 bs128 shipped128 shipped96; do
  python3 $R/bench.py --workload $w > $OUT/bench_$w.log 2>&1
  tail -1 $OUT/bench_$w.log > $OUT/bench_$w.json
  rocprofv3 --kernel-trace --stats -d $OUT/stats_$w -o s --output-format csv -- python3 $R/bench.py --workload $w --no-cpu-baseline --no-verify > /dev/null 2>&1
  cp $OUT/stats_$w/s_kernel_stats.csv $OUT/${w}_kernel_stats.csv
  echo "done $w"
done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 10 --warmup 2 --repeats 1 --no-cpu-baseline --no-verify > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 10 --warmup 2 --repeats 1 --no-cpu-baseline --no-verify > /dev/null 2>&1
python3 $R/tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write $OUT/mdct_pmc.json $OUT/step_traffic.json
cp $(ls $OUT/pmc_fetch/*/*counter_collection.csv | head -1) $OUT/pmc_fetch_size.csv
cp $(ls $OUT/pmc_write/*/*counter_collection.csv | head -1) $OUT/pmc_write_size.csv
echo "done pmc traffic"
cd $R
rm -rf gpurun_out/pmcs
bash tools/pmc_step.sh && python3 tools/pmc_step_summary.py > $OUT/sq_counters_scalar128.txt
echo "done sq counters"
rm -rf gpurun_out/pmcs
PMC_BENCH_ARGS="--workload vq128" bash tools/pmc_step.sh && python3 tools/pmc_step_summary.py > $OUT/sq_counters_vq128.txt
rm -rf gpurun_out/pmcs
echo "done sq counters vq128"
if [ -f audio-codec_amd/libpacx_dbg.so ]; then
  PACX_LIB=$R/audio-codec_amd/libpacx_dbg.so python3 tools/psy_phase_probe.py 4096 > $OUT/phases_side_mask_tail.txt 2>&1
  PACX_LIB=$R/audio-codec_amd/libpacx_dbg.so python3 tools/mdct_phase_probe.py 8192 > $OUT/phases_mdct.txt 2>&1
  PACX_LIB=$R/audio-codec_amd/libpacx_dbg.so python3 tools/mdct_phase_probe.py 262144 >> $OUT/phases_mdct.txt 2>&1
  PACX_LIB=$R/audio-codec_amd/libpacx_dbg.so python3 tools/vq_phase_probe.py 128 > $OUT/phases_vq.txt 2>&1
  PACX_LIB=$R/audio-codec_amd/libpacx_dbg.so python3 tools/vq_phase_probe.py 96 >> $OUT/phases_vq.txt 2>&1
fi
python3 tools/mdct_sweep.py 8192 16384 65536 262144 > $OUT/mdct_sweep.txt 2>&1
python3 tools/decode_probe.py 2>&1 | grep -v amdgpu > $OUT/decode_probe.txt
if [ -f audio-codec_amd/libpacx_dbg.so ]; then PACX_LIB=$R/audio-codec_amd/libpacx_dbg.so python3 tools/vqd_phase_probe.py 128 2>&1 | grep -v amdgpu > $OUT/phases_vq_dec.txt; fi
# the tail fused into the mask kernel: the same traffic passes with PACX_FUSE_TAIL=1
cd /tmp
PACX_FUSE_TAIL=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_f -- python3 $R/bench.py --steps 10 --warmup 2 --repeats 1 --no-cpu-baseline --no-verify > /dev/null 2>&1
PACX_FUSE_TAIL=1 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_f -- python3 $R/bench.py --steps 10 --warmup 2 --repeats 1 --no-cpu-baseline --no-verify > /dev/null 2>&1
python3 $R/tools/pmc_traffic.py $OUT/pmc_fetch_f $OUT/pmc_write_f $OUT/mdct_pmc_fused.json $OUT/step_traffic_fused.json
rm -rf $OUT/pmc_fetch_f $OUT/pmc_write_f
cd $R
# the step replayed from a hipGraph (not the default)
python3 bench.py --graph --no-cpu-baseline > $OUT/bench_scalar128_graph.log 2>&1 && tail -1 $OUT/bench_scalar128_graph.log > $OUT/bench_scalar128_graph.json
python3 bench.py --graph --workload bs128 --no-cpu-baseline > $OUT/bench_bs128_graph.log 2>&1 && tail -1 $OUT/bench_bs128_graph.log > $OUT/bench_bs128_graph.json
# large batches: one rank's share of BASELINE configs[4] and configs[2] at full size
python3 bench.py --frames 131072 --steps 5 --warmup 2 --repeats 5 --no-cpu-baseline > $OUT/bench_scalar128_262144.log 2>&1 && tail -1 $OUT/bench_scalar128_262144.log > $OUT/bench_scalar128_262144.json
python3 bench.py --workload bs128 --frames 862000 --steps 5 --warmup 2 --repeats 5 --no-cpu-baseline > $OUT/bench_bs128_x1000.log 2>&1 && tail -1 $OUT/bench_bs128_x1000.log > $OUT/bench_bs128_x1000.json
python3 bench.py --corpus --corpus-frames 131072 --steps 5 --warmup 2 --repeats 5 --no-cpu-baseline > $OUT/bench_corpus_eighth.log 2>&1 && tail -1 $OUT/bench_corpus_eighth.log > $OUT/bench_corpus_eighth.json
if [ -x audio-codec_amd/variants/ds_max_u64_probe ]; then ./audio-codec_amd/variants/ds_max_u64_probe > $OUT/ds_max_u64_probe.txt 2>&1; fi
if [ -x audio-codec_amd/variants/hbm_mix_probe ]; then
  for n in 8192 65536 262144; do ./audio-codec_amd/variants/hbm_mix_probe $n 2; done > $OUT/hbm_mix_ceiling.txt 2>&1
  ./audio-codec_amd/variants/hbm_mix_probe 262144 0 >> $OUT/hbm_mix_ceiling.txt 2>&1
fi
rm -rf $OUT/stats_* $OUT/pmc_fetch $OUT/pmc_write
ls $OUT
