"""Build the CPU oracle (TEST INFRASTRUCTURE -- see fastdiff_oracle.c header).

    python oracle/build.py        -> oracle/_build/libfdoracle_f64.so, libfdoracle_f32.so

The reference is pure Python, so there is no compiled `oracle/_ref` artefact: the reference
itself is executed by oracle/gen_golden.py (in the build container, where /root/reference
exists) and its outputs are frozen under tests/golden/.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_build")
SRC = os.path.join(HERE, "fastdiff_oracle.c")


def build(force: bool = False) -> dict:
    os.makedirs(OUT, exist_ok=True)
    libs = {}
    for tag, real in (("f64", "double"), ("f32", "float")):
        so = os.path.join(OUT, f"libfdoracle_{tag}.so")
        libs[tag] = so
        if not force and os.path.exists(so) and os.path.getmtime(so) >= os.path.getmtime(SRC):
            continue
        # -ffp-contract=off: no FMA contraction, so the fp32 build rounds like the reference's ATen kernels do
        cmd = ["gcc", "-O3", "-march=x86-64-v3", "-fopenmp", "-ffp-contract=off", "-fno-math-errno", "-shared", "-fPIC", "-std=c99",
               f"-DFD_REAL={real}", SRC, "-o", so, "-lm"]
        subprocess.check_call(cmd)
    return libs


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
