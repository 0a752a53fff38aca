#!/bin/bash
# round-2 session 10: per-kernel times of the LVC operator (training side) at the reference's training shape
set -u
R=$PWD; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_lvcop -o lvcop -- python $R/tools/lvc_op_profile.py > $R/gpurun_out/lvcop_prof.log 2>&1
echo "rocprof rc=$?"
f=$(find $R/gpurun_out/prof_lvcop -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY' | tee $R/gpurun_out/lvc_op_kernels.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
# per kernel name, in order of appearance: calls come in three groups (hop 8, 64, 256), 5 repetitions each
seq = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "lvc" not in n: continue
    seq[n.split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in seq.items():
    print(f"{n[:70]:70s} calls {len(v):3d}  us: " + " ".join(f"{x:7.1f}" for x in v[-15:]))
PY
