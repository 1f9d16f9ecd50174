cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pmcm/$tag -o out --output-format csv -- python3 $R/tools/mdct_sweep.py 262144 > $R/gpurun_out/pmcm_$tag.log 2>&1 || exit 1
done
