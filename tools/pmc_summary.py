"""Aggregate rocprofv3 counter_collection CSVs per kernel family (mean per dispatch) and derive HBM traffic.

    python tools/pmc_summary.py <dir with pN_counter_collection.csv> [--json out.json]

HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE reports half of the bytes of a wide
coalesced read stream (MI355X_MICROARCH.md, HBM section).  Calibration in our own access pattern: final_conv_update reads
32 channels + x (233.6 MB at B=8,T=864) and FETCH_SIZE*2*1024 = 243 MB; first_conv writes exactly 226.5 MB and
WRITE_SIZE*1024 = 226.5 MB.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

out = sys.argv[1]
BENCH_NAME = {"k_kp_gemm": "kp_gemm_fp32_fallback", "k_kp_gemm_h2": "kp_gemm_f16x2", "k_kp_gemm_w": "kp_gemm_f16x2", "k_h_wino": "h_wino", "k_h_split": "h_split", "k_final": "final_conv_fallback", "k_final_acc": "final_update",
              "k_first_conv": "first_conv", "k_embed_mlp": "embed", "k_embed_fct": "embed_fct", "k_dblock_h2": "dblock", "k_dblock": "dblock_fp32_fallback",
              "k_convt_h2": "convt", "k_convt": "convt_fp32_fallback", "k_kp_front_h2": "kp_front", "k_kp_front": "kp_front_fp32_fallback",
              "k_advance": "advance_step", "k_init_noise": "init_noise"}


def fam(name):
    m = re.search(r"k_lvc_h2<(\d+), *\d+, *(\w+), *(\d+)", name)   # the fp16-pipe LVC layer (hop 64, 256): <HOP, DIL, FINAL, UP = the fused up-sampler's ratio>
    if m:                                                            # one row per instantiation kind, as in bench.py's `kernels` table
        if int(m.group(3)) > 0:
            return "lvc_up_h" + m.group(1)
        return ("lvc_final_h" if m.group(2) in ("true", "1") else "lvc_layer_h") + m.group(1)
    m = re.search(r"k_lvc_h2<(\d+)", name)
    if m:
        return "lvc_layer_h" + m.group(1)
    if "k_lvc_h8m<" in name:                          # the hop-8 layer on 16x16x32 fp16 matrix tiles
        return "lvc_layer_h8"
    if "k_lvc_h8<" in name:                           # its all-VALU fp32 twin: an early-exit fallback launch in the default build
        return "lvc_h8_fp32_fallback"
    if "k_lvc_layer<" in name:                        # fp32 kernel for hop 64 / 256: an early-exit fallback launch
        return "lvc_fp32_fallback"
    m = re.search(r"::(k_\w+)", name)
    return BENCH_NAME.get(m.group(1), m.group(1)) if m else name[:40]


acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in sorted(glob.glob(os.path.join(out, "*counter_collection.csv"))):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = fam(row["Kernel_Name"])
            a = acc[k][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"]); a[1] += 1
dur = defaultdict(lambda: [0.0, 0])
for f in sorted(glob.glob(os.path.join(out, "p1_kernel_trace.csv"))):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = fam(row["Kernel_Name"])
            dur[k][0] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"]); dur[k][1] += 1
traffic = {}
for k in sorted(acc, key=lambda k: -dur[k][0]):
    d = dur[k][0] / max(dur[k][1], 1)
    c = {n: v[0] / v[1] for n, v in acc[k].items()}
    line = f"{k:20s} n={dur[k][1]:4d} avg_us={d/1e3:9.1f}"
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        hbm = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
        traffic[k] = round(hbm)
        line += f" hbm_MB={hbm/1e6:8.1f} (fetch*2={2*c['FETCH_SIZE']*1024/1e6:.1f} write={c['WRITE_SIZE']*1024/1e6:.1f})"
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and c.get("GRBM_GUI_ACTIVE"):
        line += f" mfma_busy={c['SQ_VALU_MFMA_BUSY_CYCLES']/(c['GRBM_GUI_ACTIVE']/8*1024):.3f}"
    for n in ("SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_LDS_BANK_CONFLICT"):
        if n in c:
            line += f" {n[3:]}={c[n]:.3g}"
    print(line)
if "--json" in sys.argv:
    traffic["_source"] = os.environ.get("PMC_SOURCE", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py --no-graph, HBM bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch")
    with open(sys.argv[sys.argv.index("--json") + 1], "w") as fh:
        json.dump(traffic, fh, indent=1, sort_keys=True)
