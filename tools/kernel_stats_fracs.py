"""Recompute bench.py's per-kernel roofline fractions from a rocprofv3 `--kernel-trace --stats` summary (kernel_stats.csv).

    python tools/kernel_stats_fracs.py profiles/r04_kernel_stats_bench_B8_T864_N4.csv [--batch 8] [--frames 864] [--bench-json line.json]

Maps every kernel name to bench.py's row (`tools/pmc_summary.py: fam` -- one row per instantiation kind of k_lvc_h2: plain, FINAL, UP),
takes the AverageNs of the CSV and bench.py's own byte model (`bench.kernel_model`): frac = bytes / avg / 8 TB/s.  With --bench-json
the `kernels` table of a bench line (the JSON line bench.py printed) is laid beside it.  Only rows whose launches all have the bench
shape are meaningful: profile a command without the b1 / fp32 / parity legs (tools/gpu_round.sh does).
"""
import csv
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
sys.argv_saved, sys.argv = sys.argv, [sys.argv[0]]
spec.loader.exec_module(bench)
sys.argv = sys.argv_saved

fam = bench.rocprof_row      # kernel name -> bench.py row (one per instantiation kind of k_lvc_h2: plain, FINAL, UP)


def main():
    args = sys.argv[1:]
    path = args[0]
    B = int(args[args.index("--batch") + 1]) if "--batch" in args else 8
    T = int(args[args.index("--frames") + 1]) if "--frames" in args else 864
    line = None
    if "--bench-json" in args:
        for ln in open(args[args.index("--bench-json") + 1]):
            if ln.startswith("{"):
                line = json.loads(ln)
    acc = {}
    for row in csv.DictReader(open(path)):
        k = fam(row["Name"])
        if k is None:
            continue
        a = acc.setdefault(k, [0, 0.0])
        a[0] += int(row["Calls"])
        a[1] += float(row["TotalDurationNs"])
    total = sum(v[1] for v in acc.values())
    print(f"{'row':18s} {'calls':>6s} {'avg_us':>9s} {'share':>7s} {'MB':>8s} {'GB/s':>8s} {'frac':>6s}   bench.py line (avg_us / frac)")
    lvc_keys = [k for k in acc if k.startswith(("lvc_layer_h", "lvc_final_h", "lvc_up_h"))]
    for k, (n, ns) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        avg = ns / n
        _, nbytes, _ = bench.kernel_model(k, B, T)
        s = f"{k:18s} {n:6d} {avg / 1e3:9.1f} {ns / total:7.3f}"
        if nbytes:
            s += f" {nbytes / 1e6:8.1f} {nbytes / avg:8.1f} {nbytes / avg / bench.HBM_PEAK_GBS:6.3f}"
        else:
            s += " " * 25
        if line and k in line.get("kernels", {}):
            e = line["kernels"][k]
            s += f"   {e['avg_us']:9.1f} / {e.get('hbm_frac', float('nan')):.3f}"
        print(s)
    if lvc_keys:
        ns = sum(acc[k][1] for k in lvc_keys)
        for label, fn in (("minimal bytes", lambda k: bench.kernel_model(k, B, T)[1]), ("unfused ops' bytes", lambda k: bench.unfused_bytes(k, B, T))):
            by = sum(fn(k) * acc[k][0] for k in lvc_keys)
            print(f"all LVC launches, time-weighted, {label}: {by / ns:.1f} GB/s = {by / ns / bench.HBM_PEAK_GBS:.3f} of 8 TB/s")
        if line and "lvc_all_12_launches" in line.get("roofline", {}):
            print("bench.py line:", line["roofline"]["lvc_all_12_launches"])


if __name__ == "__main__":
    main()
