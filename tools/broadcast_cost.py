"""What the sharded job's start-up message costs: `dist.broadcast_object_list` of (names, noise-stream ids, lengths) -- pickle on
the source, two broadcasts (size, payload), unpickle on every rank (fastdiff_amd/infer.py: synthesize_sharded) -- timed on a gloo
group of `world` processes on this host.  Prints one JSON line; bench.py --workload config4 adds it to the 8-rank projection as
its own term (a real RCCL node pays the same pickle work plus two small device broadcasts instead of the TCP ones).

    python tools/broadcast_cost.py [--world 2] [--items 64] [--reps 200]
"""
import argparse
import json
import os
import socket
import time

import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, n_items, reps, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        meta_src = [(["utt%03d.npy" % i for i in range(n_items)], list(range(n_items)), [200 + (37 * i) % 665 for i in range(n_items)])]
        for _ in range(5):
            meta = list(meta_src) if rank == 0 else [None]
            dist.broadcast_object_list(meta, src=0)
        dist.barrier()
        ts = []
        for _ in range(reps):
            meta = list(meta_src) if rank == 0 else [None]
            t0 = time.perf_counter()
            dist.broadcast_object_list(meta, src=0)
            ts.append(time.perf_counter() - t0)
            assert len(meta[0][0]) == n_items
        dist.barrier()
        ts.sort()
        if rank == 0:
            ret.put({"median_ms": ts[len(ts) // 2] * 1e3, "p90_ms": ts[int(len(ts) * 0.9)] * 1e3, "mean_ms": sum(ts) / len(ts) * 1e3})
    finally:
        dist.destroy_process_group()


def measure(world=2, n_items=64, reps=200):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, reps, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
    if any(p.exitcode != 0 for p in procs):
        raise RuntimeError("broadcast_cost: a rank failed")
    out = ret.get()
    out = {k: round(v, 4) for k, v in out.items()}
    out.update({"backend": "gloo", "world": world, "items": n_items, "reps": reps, "what": "dist.broadcast_object_list((names, ids, lens)) on rank 0, per call"})
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=2)
    ap.add_argument("--items", type=int, default=64)
    ap.add_argument("--reps", type=int, default=200)
    a = ap.parse_args()
    print(json.dumps(measure(a.world, a.items, a.reps)))
