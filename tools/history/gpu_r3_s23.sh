#!/bin/bash
# round-3 session 23: where a wave's issue cycles go in the hop-256 LVC layer (more SQ counters)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u | tr '\n' ' ' > gpurun_out/sq_counters.txt
OUT=$R/gpurun_out/pmc2; rm -rf $OUT; mkdir -p $OUT; cd /tmp
CMD="python $R/bench.py --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --no-graph --no-fp32-pipe --no-host-io --no-b1"
pass() { name=$1; shift; timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT -o $name -- $CMD > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
pass q1 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass q2 SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES
cd $R
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc2/*counter_collection.csv')):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        k = 'h256' if 'k_lvc_h2<256' in n else 'h64' if 'k_lvc_h2<64' in n else 'gemm' if 'k_kp_gemm_h2' in n else 'h8' if 'k_lvc_h8m' in n else None
        if k: acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in acc.items():
        print(f.split('/')[-1], k, {c: f"{sum(x)/len(x):.3e}" for c, x in sorted(v.items())})
PY
find $OUT -name '*.csv' -size +8M -delete
