"""A host program in plain C over the C ABI (examples/c_host.c): the boundary carries no torch types, so a binding in any language
with a C FFI looks like this.  CPU: it compiles and links against the in-tree library.  GPU: it reproduces the Python shim's
sampler output bit for bit from a job file."""
import os
import shutil
import struct
import subprocess
import sys

import numpy as np
import pytest

from conftest import load_golden, ROOT

SRC = os.path.join(ROOT, "examples", "c_host.c")
LIBDIR = os.path.join(ROOT, "fastdiff_amd", "lib")


def build(out):
    cc = shutil.which("cc") or shutil.which("gcc")
    if cc is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("needs a C compiler and the HIP headers")
    if not os.path.exists(os.path.join(LIBDIR, "libfastdiff_hip.so")):
        pytest.skip("libfastdiff_hip.so is not built")
    cmd = [cc, "-std=c99", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"), SRC, "-o", out,
           "-L" + LIBDIR, "-lfastdiff_hip", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return out


def test_c_host_compiles_as_plain_c_against_the_header(tmp_path):
    exe = build(str(tmp_path / "c_host"))
    needed = subprocess.run(["readelf", "-d", exe], capture_output=True, text=True).stdout
    assert "libfastdiff_hip.so" in needed and "libtorch" not in needed and "libpython" not in needed


@pytest.mark.gpu
def test_c_host_reproduces_the_python_shim(tmp_path):
    import torch
    import fastdiff_amd
    import gpu_common as gc
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import synth
    exe = build(str(tmp_path / "c_host"))
    sch = load_golden("schedule")
    B, T, N = 2, 7, 4
    rows, _ = gc.table_rows(sch, N)
    sd = synth.synth_state_dict(1234)
    mel = synth.synth_mel(77, B, T)
    x_T = synth.hash_normal(78, 1, B * T * 256).reshape(B, 1, T * 256)
    z = np.stack([synth.hash_normal(78, 2 + k, B * T * 256).reshape(B, 1, T * 256) for k in range(N)])
    job = tmp_path / "job.bin"
    with open(job, "wb") as f:
        f.write(struct.pack("<i", len(sd)))
        for name, a in sd.items():
            a = np.ascontiguousarray(a, np.float32)
            f.write(struct.pack("<i", len(name)) + name.encode() + struct.pack("<i", a.ndim) + struct.pack("<%dq" % a.ndim, *a.shape))
            f.write(a.tobytes())
        f.write(struct.pack("<4i", B, T, N, 0))
        for r in rows:                                        # struct fd_step: 7 floats + int32
            f.write(struct.pack("<7fi", r["t"], r["c_eps"], r["c_div"], r["sigma"], r["c1"], r["c2"], r["c3"], r["add_noise"]))
        f.write(mel.astype(np.float32).tobytes() + x_T.astype(np.float32).tobytes() + z.astype(np.float32).tobytes())
    out = tmp_path / "out.f32"
    r = subprocess.run([exe, str(job), str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "fastdiff_hip" in r.stdout and "N=4" in r.stdout
    y_c = np.fromfile(out, np.float32).reshape(B, 1, T * 256)
    m = gc.make_model()
    with torch.no_grad():
        y_py = m.sample(torch.from_numpy(mel).cuda(), rows, x_T=torch.from_numpy(x_T).cuda(), noise=torch.from_numpy(z).cuda()).cpu().numpy()
    assert np.array_equal(y_c, y_py)
