# SQ counters of the kernels of one bench step (separate --pmc passes, kernel trace only).
# usage (on the GPU box): [PMC_BENCH_ARGS="--workload vq128"] bash tools/pmc_step.sh ; python tools/pmc_step_summary.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pmcs/p$i -o out --output-format csv -- python3 $R/bench.py --min-seconds 0 --steps 5 --warmup 2 --repeats 1 --no-cpu-baseline --no-verify --no-decode-leg --pipeline 1 --no-graph --host-stream-frames 0 $PMC_BENCH_ARGS > $R/gpurun_out/pmcs_p$i.log 2>&1 || exit 1
done
