"""What an utterance costs a rank: fd_sample + epilogue calls of B = 1..16 utterances at 200..864 frames, N = 6 (BASELINE configs[3]'s
schedule), device-resident, fitted as ms(call) = c0 + c1 * B + c2 * sum(T_i).  c1 / c2 is shard.UTTERANCE_OVERHEAD_FRAMES: the
per-utterance constant of the partition's cost, in frames.   python tools/cost_model.py > profiles/r06_cost_model.json"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import fastdiff_amd  # noqa: E402
from fastdiff_amd import infer  # noqa: E402


def main():
    torch.manual_seed(1234)
    model = fastdiff_amd.FastDiff().cuda().eval()
    N, rows = 6, None
    g = torch.Generator().manual_seed(5)
    pts = []
    for B in (1, 2, 4, 8, 12, 16):
        for lo, hi in ((200, 200), (400, 400), (864, 864), (200, 864), (500, 864)):
            lens = sorted(torch.randint(lo, hi + 1, (B,), generator=g).tolist(), reverse=True)
            items = [{"item_name": str(i), "mel": (torch.rand(t, 80, generator=g) * 7.5 - 6.0).cuda(), "len": t, "uid": i} for i, t in enumerate(lens)]
            fn = lambda: infer.synthesize(model, items, N, 16, 1234, drop_last_frame=False, return_device=True)      # noqa: E731
            fn(); fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
            pts.append({"B": B, "frames": sum(lens), "t_max": lens[0], "ms": (time.perf_counter() - t0) / 10 * 1e3})
    A = np.array([[1.0, p["B"], p["frames"]] for p in pts])
    y = np.array([p["ms"] for p in pts])
    c, *_ = np.linalg.lstsq(A, y, rcond=None)
    res = y - A @ c
    print(json.dumps({"model": "ms(call) = c0 + c1 * B + c2 * sum(T_i), N = 6, device-resident, micro-batch <= 16", "c0_ms": round(float(c[0]), 4),
                      "c1_ms_per_utterance": round(float(c[1]), 4), "c2_ms_per_frame": round(float(c[2]), 6),
                      "overhead_frames_per_utterance": round(float(c[1] / c[2]), 1), "per_call_overhead_frames": round(float(c[0] / c[2]), 1),
                      "rms_residual_ms": round(float(np.sqrt((res ** 2).mean())), 4), "max_residual_ms": round(float(np.abs(res).max()), 4), "points": pts}))


if __name__ == "__main__":
    main()
