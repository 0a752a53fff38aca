#!/bin/bash
# round-3 session 6: B=1 timeline (gaps between the kernels of the replayed graph), fixed hand-over test, config 4 with the numpy packing,
# CU-mask sanity (half the CUs must show in the time), PMC traffic passes of this round
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
echo "== B=1 gaps"; bash tools/gap_probe.sh --batch 1 --no-b1 --no-fp32-pipe; cp gpurun_out/gaps.txt gpurun_out/gaps_b1.txt
python - <<'PY' | tee gpurun_out/b1_kernel_table.txt
import csv, glob, re, collections
rows = []
for f in glob.glob('/tmp/gap/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
idx = max(i for i, r in enumerate(rows) if 'k_embed_mlp' in r[2])
call = rows[idx:]
end = max(i for i, r in enumerate(call) if 'k_advance' in r[2])
call = call[:end + 1]
acc = collections.OrderedDict()
for s, e, n in call:
    m = re.search(r'::(k_\w+(<[^>]*>)?)', n)
    k = m.group(1) if m else n[:30]
    a = acc.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3
tot = sum(v[1] for v in acc.values())
print("one B=1 sample call (N=4) in the replayed graph: kernel, launches, total us, share of busy time")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:34s} {v[0]:4d} {v[1]:9.1f} {100*v[1]/tot:6.1f} %")
PY
echo "== hand-over test"; timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "hands_over" 2>&1 | tail -3 | cut -c1-300
echo "== config4"; timeout 600 python bench.py --workload config4 --steps 5 --warmup 2 > gpurun_out/bench_config4.log 2>&1 ; python - <<'PY'
import json
for line in open('gpurun_out/bench_config4.log'):
    if line.startswith('{'):
        d = json.loads(line); print(d['value'], d['ms_per_step'], d['projection'])
PY
echo "== CU mask sanity: B=8 on half the CUs"
HSA_CU_MASK="0:0-127" timeout 300 python tools/ab_opts.py --batch 8 --reps 1 "" 2>&1 | grep "^B="
timeout 300 python tools/ab_opts.py --batch 8 --reps 1 "" 2>&1 | grep "^B="
echo "== PMC"
OUT=$R/gpurun_out/pmc; rm -rf $OUT; mkdir -p $OUT; cd /tmp
CMD="python $R/bench.py --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --no-graph --no-fp32-pipe --no-host-io --no-b1"
pass() { name=$1; shift; timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT -o $name -- $CMD > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
pass p2 FETCH_SIZE
pass p3 WRITE_SIZE
pass p4 SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_MFMA
pass p1 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
cd $R; PMC_SOURCE="round 3, rocprofv3 --pmc passes of bench.py --steps 2 --warmup 1 --no-graph (tools/history/gpu_r3_s6.sh); HBM bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch" python tools/pmc_summary.py $OUT --json $OUT/pmc_traffic.json > $OUT/summary.txt 2>&1; cat $OUT/summary.txt | grep -v "at::native\|rocclr" | head -40
find $OUT -name '*.csv' -size +8M -delete
