"""One handle, two streams, the first one destroyed in between (fd_api.cpp: follow_stream): the library remembers the stream of its
last call to order the next call behind it; a caller may destroy that stream once its work is done.  The next call (on another stream)
must neither fail nor crash, and must give the result of a fresh handle.  Run in a process of its own (a runtime that did not
validate stream handles would take the process down, not raise)."""
import ctypes as ct
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import fastdiff_amd                      # noqa: E402


def main():
    import faulthandler
    faulthandler.enable()
    hip = ct.CDLL("libamdhip64.so")
    torch.manual_seed(0)
    m = fastdiff_amd.FastDiff().cuda().eval()
    B, T = 2, 40
    mel = (torch.rand(B, 80, T) * 7.5 - 6.0).cuda()
    rows = [{"t": 190.0 - 45.0 * k, "c_eps": 0.02, "c_div": 0.99, "sigma": 0.05, "c1": 1.0, "c2": 0.0, "c3": 0.0, "add_noise": int(k < 3)}
            for k in range(4)]
    with torch.no_grad():
        want = fastdiff_amd.FastDiff().cuda().eval()
        want.load_state_dict(m.state_dict())
        ref = want.sample(mel, rows, seed=5)
        s = ct.c_void_p()
        assert hip.hipStreamCreate(ct.byref(s)) == 0
        ext = torch.cuda.ExternalStream(s.value)
        with torch.cuda.stream(ext):
            y1 = m.sample(mel, rows, seed=5)
        ext.synchronize()
        assert torch.equal(y1, ref)
        print("call on the stream: ok", flush=True)
        del ext
        rc = hip.hipStreamDestroy(s)
        print("hipStreamDestroy rc", rc, flush=True)
        y2 = m.sample(mel, rows, seed=5)          # default stream: the handle's last stream no longer exists
        torch.cuda.synchronize()
        assert torch.equal(y2, ref)
        y3 = m.sample(mel, rows, seed=5)
        assert torch.equal(y3, ref)
    print("stream switch over a destroyed stream: ok")


if __name__ == "__main__":
    main()
