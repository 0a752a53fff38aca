"""The rare mismatch of tests/test_sharded_synthesis.py shows only in the FIRST call of a FRESH process that shares the GPU with
another starting process.  Driver: a reference from one solo worker, then ROUNDS rounds of K concurrent fresh worker processes, each
vocoding the job's first micro-batch (the 39- and the 32-frame utterance) exactly as rank 0 of the test does; their PCM is compared
bit for bit with the reference.  Usage: python tools/fresh_proc_hunt.py [K] [ROUNDS] [key=value,...library options]
(worker: python tools/fresh_proc_hunt.py --worker OUT.npy [options])"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(out_path, opts):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import numpy as np
    import torch
    import gpu_common
    from fastdiff_amd import infer
    g = torch.Generator().manual_seed(11)
    lens = [40, 12, 33, 7, 25, 18, 40, 3, 29]
    items = [{"item_name": f"utt{i:02d}.npy", "mel": torch.rand(t, 80, generator=g) * 7.5 - 6.0, "len": t, "uid": i} for i, t in enumerate(lens)]
    model = gpu_common.make_model()
    for kv in opts:
        model.set_option(*kv.split("=", 1))
    out = infer.synthesize(model, [items[0], items[2]], n_steps=4, max_batch=2, seed=77, drop_last_frame=True)
    np.save(out_path, np.concatenate([out["utt00.npy"], out["utt02.npy"]]))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        return worker(sys.argv[2], [a for a in sys.argv[3:] if a])
    import numpy as np
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    opts = sys.argv[3].split(",") if len(sys.argv) > 3 and sys.argv[3] else []
    tmp = os.path.join(ROOT, "gpurun_out", "fresh")
    os.makedirs(tmp, exist_ok=True)
    cmd = [sys.executable, os.path.abspath(__file__), "--worker"]
    ref_path = os.path.join(tmp, "ref.npy")
    subprocess.run(cmd + [ref_path] + opts, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    ref = np.load(ref_path)
    bad = total = 0
    for r in range(rounds):
        paths = [os.path.join(tmp, f"w{r}_{k}.npy") for k in range(K)]
        procs = [subprocess.Popen(cmd + [p] + opts, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for p in paths]
        for pr in procs:
            pr.wait()
        for p in paths:
            total += 1
            if not os.path.exists(p):
                print(f"round {r}: worker produced nothing ({p})", flush=True)
                bad += 1
                continue
            a = np.load(p)
            if np.array_equal(a, ref):
                os.remove(p)
                continue
            bad += 1
            d = np.abs(a.astype(np.int32) - ref.astype(np.int32))
            idx = np.nonzero(d)[0]
            big = np.nonzero(d > 8)[0]
            print(f"round {r}: {os.path.basename(p)}: {idx.size} samples differ, max |d| {d.max()}, |d| > 8 in [{big[0] if big.size else -1}, {big[-1] if big.size else -1}]", flush=True)
    print(f"K={K} rounds={rounds} options={opts}: {bad} of {total} fresh processes differ from the solo reference", flush=True)


if __name__ == "__main__":
    main()
