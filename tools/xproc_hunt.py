"""Round 3: which activity of a SECOND process disturbs a vocoding process on the same GPU?

Round 2 left one loose end (profiles/r02/s12_s19_two_processes_one_gpu.txt): with two ranks vocoding on ONE GPU, ~3 % of runs gave one
utterance with a few hundred wrong samples -- 5 of 5 times an utterance of rank 0, 4 of 5 times in its FIRST micro-batch, i.e. while
rank 1 (which receives its mels a moment later) goes through the first-call work of a fresh process: pinned allocations, the
workspace hipMalloc + hipMemset, code-object loading, graph instantiation.  Two long-lived processes next to each other never
showed it.  This tool separates the candidates: ONE long-lived victim (the job's first micro-batch in a loop on one handle, every
result compared bit for bit with the first) next to an aggressor process that does exactly one class of operation in a loop.

    python tools/xproc_hunt.py [seconds_per_mode] [modes,comma,separated] [victim: fd|plain]

modes: idle (no aggressor), torchinit / torchops / model / chost / firstcall_nograph (fresh processes that stop earlier on the way to a
first call, or make it without torch -- examples/c_host -- or without a graph), kernels (a resident process that only launches fill kernels), hostalloc (hipHostMalloc / hipHostRegister churn),
devalloc (hipMalloc + hipMemset + hipFree churn), procs (short-lived processes: context creation, one allocation, one memset, exit),
firstcall (fresh Python processes that build the model and vocode once -- what rank 1 of the test does).
victim plain = a torch-only victim (no kernel of this repo: y = x * 2 + 1 over 64 MB, checksum compared), to tell a platform effect
from a fault of this library's kernels.
(aggressor entry: python tools/xproc_hunt.py --aggressor MODE)"""
import ctypes as ct
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def aggressor(mode):
    if mode == "procs":
        child = [sys.executable, os.path.abspath(__file__), "--aggressor", "oneshot"]
        while True:
            ps = [subprocess.Popen(child, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for _ in range(3)]
            for p in ps:
                p.wait()
    fresh = {      # fresh processes, two at a time, each doing one thing and exiting
        "firstcall": [sys.executable, os.path.join(ROOT, "tools", "fresh_proc_hunt.py"), "--worker", "/tmp/xproc_firstcall_%d.npy"],
        "firstcall_nograph": [sys.executable, os.path.join(ROOT, "tools", "fresh_proc_hunt.py"), "--worker", "/tmp/xproc_firstcall_%d.npy", "graph=0"],
        "torchinit": [sys.executable, "-c", "import torch; torch.zeros(1, device='cuda'); torch.cuda.synchronize()  # %d"],
        "torchops": [sys.executable, "-c", "import torch; x = torch.rand(1 << 24, device='cuda'); [x.mul_(1.0001) for _ in range(200)]; "
                     "y = torch.nn.functional.conv1d(x.view(1, 1, -1), torch.ones(32, 1, 7, device='cuda')); torch.cuda.synchronize()  # %d"],
        "model": [sys.executable, "-c", f"import sys; sys.path[:0] = [{ROOT!r}, {os.path.join(ROOT, 'oracle')!r}, {os.path.join(ROOT, 'tests')!r}]; "
                  "import torch, gpu_common; m = gpu_common.make_model(); m._ready(torch.device('cuda', 0)); torch.cuda.synchronize()  # %d"],
        "chost": [os.path.join(ROOT, "examples", "c_host"), "/tmp/xproc_job.bin", "/tmp/xproc_chost_%d.f32"],      # the library without Python or torch
        "chost_pad": [os.path.join(ROOT, "examples", "c_host"), "/tmp/xproc_job.bin", "/tmp/xproc_chost_%d.f32", "1536"],      # the same, every address moved by 1.5 GiB
        # the library's code object loaded and ONE small kernel of it launched (the int16 epilogue on 1 MB): no workspace, no sampler
        "modload": [sys.executable, os.path.abspath(__file__), "--aggressor", "modload_once"],
        # a different code object (a stand-alone HIP binary of tools/ubench), fresh process each time
        "othermod": [os.path.join(ROOT, "tools", "ubench", "copy_mix_probe")],
        # round 4: a generic kernel of the hop-256 layer's resource shape (72 KB LDS, 2 workgroups per CU, matrix instructions on LDS
        # operands), none of this library's code: tools/ubench/xproc_repro.hip, 40 launches per process
        "generic": [os.path.join(ROOT, "tools", "ubench", "xproc_repro"), "aggressor", "40", "%d"],
    }
    if mode in ("fdloop", "fdloop_pad"):      # a second long-lived sampler loop of the same shapes (its first call is ~8 s after its start)
        env = dict(os.environ, FD_HUNT_PAD_MB="1536") if mode == "fdloop_pad" else dict(os.environ)
        os.execve(sys.executable, [sys.executable, os.path.abspath(__file__), "100000", "idle", "fd"], env)
    if mode in fresh:
        while True:
            ps = [subprocess.Popen([a.replace("%d", str(i)) for a in fresh[mode]], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for i in range(2)]
            for p in ps:
                p.wait()
    hip = ct.CDLL("libamdhip64.so")
    hip.hipSetDevice(0)
    if mode == "modload_once":
        lib = ct.CDLL(os.path.join(ROOT, "fastdiff_amd", "lib", "libfastdiff_hip.so"))
        cfg = (ct.c_int * 64)()
        lib.fd_default_config(cfg)
        h, d, p = ct.c_void_p(), ct.c_void_p(), ct.c_void_p()
        assert lib.fd_create(cfg, 0, ct.byref(h)) == 0
        hip.hipMalloc(ct.byref(d), ct.c_size_t(1 << 20))
        hip.hipMemset(d, 0x3c, ct.c_size_t(1 << 20))
        hip.hipMalloc(ct.byref(p), ct.c_size_t(1 << 19))
        lib.fd_peak_normalize_int16.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_int, ct.c_int64, ct.c_void_p, ct.c_void_p]
        assert lib.fd_peak_normalize_int16(h, d, 1, 1 << 18, p, None) == 0
        hip.hipDeviceSynchronize()
        return
    if mode == "oneshot":
        d = ct.c_void_p()
        if hip.hipMalloc(ct.byref(d), ct.c_size_t(64 << 20)) == 0:
            hip.hipMemset(d, 1, ct.c_size_t(64 << 20))
            hip.hipDeviceSynchronize()
            hip.hipFree(d)
        return
    k = 0
    if mode == "kernels":
        d = ct.c_void_p()
        hip.hipMalloc(ct.byref(d), ct.c_size_t(256 << 20))
        while True:
            hip.hipMemsetAsync(d, k & 255, ct.c_size_t(256 << 20), None)
            if k % 8 == 7:
                hip.hipDeviceSynchronize()
            k += 1
    while True:
        size = (1 + k % 7) << 20
        if mode == "hostalloc":
            p = ct.c_void_p()
            if hip.hipHostMalloc(ct.byref(p), ct.c_size_t(size), 0) == 0:
                ct.memset(p, k & 255, 4096)
                hip.hipHostFree(p)
            a = (ct.c_char * size)()
            if hip.hipHostRegister(a, ct.c_size_t(size), 0) == 0:
                hip.hipHostUnregister(a)
        elif mode == "devalloc":
            d = ct.c_void_p()
            if hip.hipMalloc(ct.byref(d), ct.c_size_t(size * 16)) == 0:
                hip.hipMemset(d, 0, ct.c_size_t(size * 16))
                hip.hipDeviceSynchronize()
                hip.hipFree(d)
        else:
            raise SystemExit(f"unknown aggressor mode {mode}")
        k += 1


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--aggressor":
        return aggressor(sys.argv[2])
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
    modes = (sys.argv[2] if len(sys.argv) > 2 else "idle,kernels,hostalloc,devalloc,procs,firstcall").split(",")
    victim = sys.argv[3] if len(sys.argv) > 3 else "fd"
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import numpy as np
    import torch
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    if victim == "fd":
        import gpu_common
        from fastdiff_amd import schedules
        from fastdiff_amd.sampler import InferenceSchedule
        g = torch.Generator().manual_seed(11)
        lens_all = [40, 12, 33, 7, 25, 18, 40, 3, 29]
        mels_all = [torch.rand(t, 80, generator=g) * 7.5 - 6.0 for t in lens_all]
        pick = [0, 2]
        lens = [lens_all[i] - 1 for i in pick]
        T = max(lens)
        mel = torch.zeros(len(pick), 80, T)
        for b, i in enumerate(pick):
            mel[b, :, : lens[b]] = mels_all[i][: lens[b]].T
        rows = InferenceSchedule(schedules.training_hyperparams(), schedules.noise_schedule_for(4), verbose=False).rows()
        if os.environ.get("FD_HUNT_PAD_MB"):      # moves the addresses of everything allocated afterwards
            pad = torch.empty(int(os.environ["FD_HUNT_PAD_MB"]) << 20, dtype=torch.uint8, device="cuda")
        model = gpu_common.make_model()
        for kv in filter(None, os.environ.get("FD_HUNT_OPTS", "").split(",")):
            model.set_option(*kv.split("=", 1))
        mel_d = mel.cuda()
        if "chost" in modes:      # job file of examples/c_host.c: the state_dict, the schedule, this micro-batch, injected noise
            import struct
            import synth
            sd = synth.synth_state_dict(1234)
            with open("/tmp/xproc_job.bin", "wb") as f:
                f.write(struct.pack("<i", len(sd)))
                for name, a in sd.items():
                    a = np.ascontiguousarray(a, np.float32)
                    f.write(struct.pack("<i", len(name)) + name.encode() + struct.pack("<i", a.ndim) + struct.pack("<%dq" % a.ndim, *a.shape))
                    f.write(a.tobytes())
                f.write(struct.pack("<4i", 2, T, 4, 0))
                for r in rows:
                    f.write(struct.pack("<7fi", r["t"], r["c_eps"], r["c_div"], r["sigma"], r["c1"], r["c2"], r["c3"], r["add_noise"]))
                f.write(mel.numpy().astype(np.float32).tobytes())
                f.write(synth.hash_normal(78, 1, 2 * T * 256).astype(np.float32).tobytes())
                f.write(np.stack([synth.hash_normal(78, 2 + k, 2 * T * 256) for k in range(4)]).astype(np.float32).tobytes())

        def once():
            with torch.no_grad():
                # FD_HUNT_NOLENS=1: the padded batch as it is (the naive kernel set refuses ragged batches)
                wav = model.sample(mel_d, rows, ddim=False, seed=77, lens=None if os.environ.get("FD_HUNT_NOLENS") == "1" else lens, stream_ids=pick)
            return wav.cpu().numpy()
    else:
        x = torch.arange(16 << 20, dtype=torch.float32, device="cuda") * 1e-3

        def once():
            y = x * 2.0 + 1.0
            z = (y.view(-1, 64).sum(1))          # per-256-byte checksums
            return z.cpu().numpy()
    ref = once()
    for _ in range(20):
        assert np.array_equal(once(), ref), "victim is not reproducible on its own"
    summary = []
    for mode in modes:
        ag = None
        if mode != "idle":
            env = dict(os.environ)
            for kv in filter(None, os.environ.get("XPROC_AGG_ENV", "").split(";")):      # e.g. a CU mask for the aggressor only
                env[kv.split("=", 1)[0]] = kv.split("=", 1)[1]
            env.pop("HSA_CU_MASK", None) if "HSA_CU_MASK" not in os.environ.get("XPROC_AGG_ENV", "") else None
            ag = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--aggressor", mode], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                  start_new_session=True, env=env)
            time.sleep(1.0)
        t0, it, bad = time.time(), 0, 0
        while time.time() - t0 < secs:
            r = once()
            it += 1
            if not np.array_equal(r, ref):
                bad += 1
                d = np.abs(r.astype(np.float64) - ref).reshape(-1)
                nz = np.nonzero(d)[0]
                print(f"  {mode}: t = {time.time() - t0:.1f} s, iteration {it}: {nz.size} values differ, max {d.max():.3e}, first {nz[0]}, last {nz[-1]}", flush=True)
                if bad <= 4:
                    np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"xproc_{victim}_{mode}_{it}.npz"), got=r, ref=ref)
        if ag is not None:
            os.killpg(ag.pid, 9)          # the aggressor's own process group (start_new_session): exactly what this script started
            ag.wait()
        line = (f"victim {victim} [{os.environ.get('FD_HUNT_OPTS', '')}] [victim mask {os.environ.get('HSA_CU_MASK', '-')}; aggressor env "
                f"{os.environ.get('XPROC_AGG_ENV', '-')}], aggressor {mode}: {it} iterations in {time.time() - t0:.1f} s, {bad} mismatches")
        print(line, flush=True)
        summary.append(line)
    with open(os.path.join(ROOT, "gpurun_out", f"xproc_hunt_{victim}.txt"), "a") as f:
        f.write("\n".join(summary) + "\n")


if __name__ == "__main__":
    main()
