# k_vq time against the shape-bit threshold of the level-by-level walk (PACX_VQ_BFS; 0 = depth first only)
cd /tmp && export TMPDIR=/tmp
for t in 0 1 48 64 96 128 192 256; do
  export PACX_VQ_BFS=$t
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/vqbfs/$t -o s --output-format csv -- python3 /root/repo/bench.py --workload ${1:-vq128} --no-cpu-baseline --no-verify --steps 10 --repeats 3 > /root/repo/gpurun_out/vqbfs_$t.log 2>&1 || exit 1
  python3 - <<PY
import csv,glob
f=glob.glob("/root/repo/gpurun_out/vqbfs/$t/**/s_kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if r["Name"].startswith("k_vq("): print("threshold $t:", round(float(r["AverageNs"])/1e3,1), "us")
PY
done
