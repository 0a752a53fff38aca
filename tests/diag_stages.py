"""Stage-isolation diagnostic (run on the GPU box): all-naive vs oracle, then one fast stage at a time.
Writes gpurun_out/stage_report.json.   python tests/diag_stages.py [B T]"""
import json
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import synth  # noqa: E402
from oracle import Oracle  # noqa: E402
import gpu_common as gc  # noqa: E402


def main():
    cases = [(1, 4), (2, 37)] if len(sys.argv) < 3 else [(int(sys.argv[1]), int(sys.argv[2]))]
    report = {}
    o = Oracle("f64")
    o.set_weights(synth.synth_state_dict(1234))
    m = gc.make_model()
    m.set_option("taps", "1")
    for B, T in cases:
        mel = synth.synth_mel(7, B, T)
        audio = synth.synth_audio(7, B, T)
        steps = np.linspace(3.25, 480.5, B).astype(np.float32)
        y_ref, taps_ref = o.forward(audio, mel, steps, taps=True)
        for mode in ["naive"] + gc.STAGES + ["fast"]:
            key = f"B{B}_T{T}_{mode}"
            try:
                if mode in ("naive", "fast"):
                    m.set_option("kernels", mode)
                else:
                    m.set_option("kernels", "naive")
                    m.set_option("kernels." + mode, "fast")
                y = gc.run_forward(m, audio, mel, steps)
                taps = gc.read_taps(m, B, T)
                entry = {"out": gc.maxdiff(y, y_ref)}
                for k in taps:
                    entry[k] = gc.maxdiff(taps[k], taps_ref[k])
                entry["nan"] = bool(np.isnan(y).any())
                report[key] = entry
            except Exception as e:   # keep going: the point is to see every stage
                report[key] = {"error": repr(e), "trace": traceback.format_exc()[-800:]}
            print(key, json.dumps(report[key])[:400], flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "stage_report.json"), "w") as f:
        json.dump(report, f, indent=1)


if __name__ == "__main__":
    main()
