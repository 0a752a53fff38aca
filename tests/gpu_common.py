"""Helpers shared by the GPU parity tests (imported only under -m gpu)."""
import os

import numpy as np
import torch

import fastdiff_amd
import synth

STAGES = ["first", "dblock", "kp_front", "kp_gemm", "convt", "lvc", "final"]

_perm = None
_bperm = None


def kernel_perm():
    """reference kernel_conv row ((layer*32+in)*64+out)*3+tap -> position in the packed frame record."""
    global _perm
    if _perm is None:
        p = np.empty(24576, np.int64)
        for layer in range(4):
            for i in range(32):
                for o in range(64):
                    for k in range(3):
                        p[((layer * 32 + i) * 64 + o) * 3 + k] = fastdiff_amd.FastDiff.kernel_index(layer, i, o, k)
        _perm = p
    return _perm


def bias_perm():
    global _bperm
    if _bperm is None:
        _bperm = np.array([fastdiff_amd.FastDiff.bias_index(l, o) for l in range(4) for o in range(64)], np.int64)
    return _bperm


def unpack_kpack(kpack, B, T):
    """packed [B,T,24832] -> (kernels [B,24576,T], bias [B,256,T]) in the reference's conv-output layout."""
    rec = kpack.reshape(B, T, 24832)
    kern = rec[:, :, kernel_perm()].transpose(0, 2, 1)
    bias = rec[:, :, bias_perm()].transpose(0, 2, 1)
    return kern, bias


def make_model(seed=1234, device="cuda", contractive=False):
    m = fastdiff_amd.FastDiff()
    sd = {k: torch.from_numpy(v.copy()) for k, v in synth.synth_state_dict(seed, contractive=contractive).items()}
    m.load_state_dict(sd, strict=True)
    m = m.to(device).eval()
    for kv in filter(None, os.environ.get("FD_TEST_OPTS", "").split(",")):      # bisecting aid: library options for every test model
        m.set_option(*kv.split("=", 1))
    return m


def run_forward(m, audio, mel, steps):
    with torch.no_grad():
        y = m((torch.from_numpy(np.ascontiguousarray(audio)).cuda(), torch.from_numpy(np.ascontiguousarray(mel)).cuda(),
               torch.from_numpy(np.asarray(steps, np.float32).reshape(-1, 1)).cuda()))
    torch.cuda.synchronize()
    return y.cpu().numpy()


def read_taps(m, B, T):
    L = T * 256
    taps = {"a0": m.read_tap("a0").reshape(B, 32, L), "a1": m.read_tap("a1").reshape(B, 32, L // 4),
            "a2": m.read_tap("a2").reshape(B, 32, L // 32), "a3": m.read_tap("a3").reshape(B, 32, T)}
    for n, hop in enumerate((8, 64, 256)):
        k, b = unpack_kpack(m.read_tap(f"kpack{n}"), B, T)
        taps[f"kernels{n}"], taps[f"bias{n}"] = k, b
        taps[f"x{n}"] = m.read_tap(f"x{n}").reshape(B, 32, T * hop)
    return taps


def maxdiff(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())


def noise_from_seed(seed, B, T, N):
    z = np.zeros((N, B, 1, T * 256), np.float32)
    for n in range(1, N):
        z[n] = synth.hash_normal(seed, 2 + n, B * T * 256).reshape(B, 1, T * 256)
    return z


def table_rows(sch, N):
    """golden schedule fixture -> rows in execution order, plus the oracle-style table dict."""
    rows = []
    for n in range(N - 1, -1, -1):
        rows.append({"t": float(np.float32(sch[f"N{N}_steps"][n])), "c_eps": float(sch[f"N{N}_c_eps"][n]),
                     "c_div": float(sch[f"N{N}_c_div"][n]), "sigma": float(sch[f"N{N}_sigma_hat"][n]),
                     "c1": float(sch[f"N{N}_c1"][n]), "c2": float(sch[f"N{N}_c2"][n]), "c3": float(sch[f"N{N}_c3"][n]),
                     "add_noise": int(n > 0)})
    table = {k: sch[f"N{N}_{k}"] for k in ("steps", "c_eps", "c_div", "sigma_hat", "c1", "c2", "c3")}
    return rows, table


def exec_order_noise(z):
    """z[n] (added after reverse index n) -> noise[k] for the k-th executed step (n = N-1-k)."""
    return np.ascontiguousarray(z[::-1])


# ---- torch twin of oracle/synth.py's counter hash (integer arithmetic only: bit-identical on any device) --------------------------
_M_GOLD, _M_A, _M_B = 0x9E3779B97F4A7C15, 0xBF58476D1CE4E5B9, 0x94D049BB133111EB


def _i64(v):
    """A Python int (mod 2^64) as the int64 with the same bits."""
    v &= 0xFFFFFFFFFFFFFFFF
    return v - (1 << 64) if v >= (1 << 63) else v


def _lsr(x, k):
    """Logical right shift of int64 bit patterns (torch's >> is arithmetic)."""
    return (x >> k) & ((1 << (64 - k)) - 1)


def _splitmix64_torch(x):
    x = x + _i64(_M_GOLD)
    z = (x ^ _lsr(x, 30)) * _i64(_M_A)
    z = (z ^ _lsr(z, 27)) * _i64(_M_B)
    return z ^ _lsr(z, 31)


def _splitmix64_int(x):
    m = 0xFFFFFFFFFFFFFFFF
    x = (x + _M_GOLD) & m
    z = ((x ^ (x >> 30)) * _M_A) & m
    z = ((z ^ (z >> 27)) * _M_B) & m
    return z ^ (z >> 31)


def hash_uniform_torch(seed, stream, n, device="cuda"):
    """synth.hash_uniform(seed, stream, n) computed with torch int64 ops on `device` (two's-complement wrap-around = mod 2^64)."""
    base = _i64(_splitmix64_int((seed * 0x100000001B3 + stream) & 0xFFFFFFFFFFFFFFFF))
    idx = torch.arange(n, dtype=torch.int64, device=device)
    h = _splitmix64_torch(idx ^ base)
    u24 = _lsr(h, 40)
    return ((u24 - (1 << 23)).to(torch.float32) / float(1 << 23))


def hash_normal_torch(seed, stream, n, device="cuda"):
    """synth.hash_normal(seed, stream, n): the sum of 12 hash uniforms, accumulated in float64 in the same order."""
    acc = torch.zeros(n, dtype=torch.float64, device=device)
    for k in range(12):
        acc += hash_uniform_torch(seed, stream * 16 + k, n, device).to(torch.float64) * 0.5 + 0.5
    return (acc - 6.0).to(torch.float32)
