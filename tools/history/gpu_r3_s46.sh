#!/bin/bash
# round-3 session 46: per-kernel times with the up-sampler inside the first layer and without (rocprofv3 kernel stats of the replayed graph)
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for f in 1 0; do
  OUT=$R/gpurun_out/prof_up$f; rm -rf $OUT; mkdir -p $OUT
  cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o kt -- python $R/bench.py --steps 5 --warmup 1 --no-roofline --no-cpu-baseline --no-fp32-pipe --no-host-io --no-b1 --opt fuse_up=$f > $OUT/run.log 2>&1; cd $R
  rm -f $OUT/kt_kernel_trace.csv
  python - $OUT/kt_kernel_stats.csv $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("fuse_up =", sys.argv[2])
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:22]:
    n = r["Name"]
    if "k_" not in n: continue
    print(f"  {float(r['AverageNs'])/1e3:8.1f} us x {int(r['Calls']):4d}  {float(r['Percentage']):5.1f} %  {n[:110]}")
PY
done > gpurun_out/up_kernel_stats.txt 2>&1
cat gpurun_out/up_kernel_stats.txt
