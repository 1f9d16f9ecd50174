# A/B of k_vq with the enumeration lookups and/or the pulse ranking stubbed out (wrong output, timing only):
#   python audio-codec_amd/build.py --variant vqterm k_vq.hip -DVQ_STUB_TERM   (likewise vqrank, vqboth), then on the GPU box: bash tools/vq_stub_ab.sh
cd /tmp && export TMPDIR=/tmp
for v in "" vqterm vqrank vqboth; do
  if [ -n "$v" ]; then export PACX_LIB=/root/repo/audio-codec_amd/variants/libpacx_$v.so; else unset PACX_LIB; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/vqab/$v -o s --output-format csv -- python3 /root/repo/bench.py --workload vq128 --no-cpu-baseline --no-verify --steps 10 --repeats 3 > /root/repo/gpurun_out/vqab_$v.log 2>&1 || exit 1
  python3 - <<PY
import csv,glob
f=glob.glob("/root/repo/gpurun_out/vqab/$v/**/s_kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if r["Name"].startswith("k_vq("): print("variant '$v'", round(float(r["AverageNs"])/1e3,1))
PY
done
