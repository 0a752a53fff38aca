#!/bin/bash
# Idle time between consecutive kernels of the replayed step graph: tools/gap_probe.sh [bench args]   (writes gpurun_out/gaps.txt)
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/gap -o g -- python $R/bench.py --steps 5 --warmup 2 --no-roofline --no-cpu-baseline --no-host-io "$@" > /tmp/gap.log 2>&1
cd $R; python - <<'PY' | tee gpurun_out/gaps.txt
import csv, glob, re
rows = []
for f in glob.glob('/tmp/gap/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
# the last sample call: from the last k_embed_mlp on
idx = max(i for i, r in enumerate(rows) if 'k_embed_mlp' in r[2])
call = rows[idx:]
end = max(i for i, r in enumerate(call) if 'k_advance' in r[2])
call = call[:end + 1]
busy = sum(e - s for s, e, _ in call)
span = call[-1][1] - call[0][0]
gaps = [(call[i + 1][0] - call[i][1], call[i][2], call[i + 1][2]) for i in range(len(call) - 1)]
print(f"kernels {len(call)}  span {span/1e3:.1f} us  busy {busy/1e3:.1f} us  idle {100*(span-busy)/span:.1f} %")
short = lambda n: (re.search(r'(k_\w+(<[^>]*>)?)', n) or [n[:30]])[0] if re.search(r'(k_\w+)', n) else n[:30]
pos = [g for g in gaps if g[0] > 0]
print(f"mean gap {sum(g[0] for g in pos)/max(len(pos),1)/1e3:.2f} us over {len(pos)} gaps; overlapped pairs {len(gaps)-len(pos)}")
for g in sorted(gaps, reverse=True)[:12]:
    print(f"  {g[0]/1e3:7.2f} us after {short(g[1])} before {short(g[2])}")
PY
