"""CPU placement of one rank per GPU.

The reference starts one process per GPU (utils/trainer.py:94-107 mp.spawn) and leaves their placement to the kernel.  On an MI355X
node (8 GPUs, two sockets, 128-256 cores) eight Python ranks that each launch graphs, pack pinned buffers and wait on events share
the host: a rank whose threads sit on the socket its GPU is NOT attached to pays a cross-socket hop on every pinned copy and
doorbell, and ranks that migrate over each other's cores add jitter that a max-over-ranks timing sees in full.  `bind_rank` pins
the calling process to its own slice of the cores of the NUMA node its GPU hangs off (sysfs: the GPU's PCI function -> numa_node /
local_cpulist); ranks that share a node split its cores evenly.  Pure host logic, no HIP call; everything it cannot find out
(no sysfs entry, numa_node = -1, a container without the PCI tree) degrades to an even split of the cores this process may use.
"""
import os
from typing import Dict, List, Optional, Sequence


def parse_cpulist(text: str) -> List[int]:
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the kernel's list format, e.g. /sys/devices/system/node/node0/cpulist)."""
    cpus: List[int] = []
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            lo, hi = part.split("-", 1)
            cpus.extend(range(int(lo), int(hi) + 1))
        else:
            cpus.append(int(part))
    return sorted(set(cpus))


def pci_address(domain: int, bus: int, device: int, function: int = 0) -> str:
    return "%04x:%02x:%02x.%x" % (domain, bus, device, function)


def gpu_local_cpus(pci_addr: str, sysfs: str = "/sys") -> Optional[List[int]]:
    """The cores of the NUMA node a PCI function is attached to, or None when sysfs does not say."""
    base = os.path.join(sysfs, "bus", "pci", "devices", pci_addr)
    try:
        node = int(open(os.path.join(base, "numa_node")).read().strip())
    except (OSError, ValueError):
        node = -1
    for path in ([os.path.join(sysfs, "devices", "system", "node", "node%d" % node, "cpulist")] if node >= 0 else []) + [os.path.join(base, "local_cpulist")]:
        try:
            cpus = parse_cpulist(open(path).read())
        except (OSError, ValueError):
            continue
        if cpus:
            return cpus
    return None


def plan(local_cpus: Sequence[Optional[Sequence[int]]], allowed: Sequence[int]) -> List[List[int]]:
    """Core sets for the ranks of one host.  local_cpus[r] = the cores next to rank r's GPU (None = unknown), allowed = the cores this
    job may use.  Ranks whose GPUs share a NUMA node split that node's (allowed) cores evenly, in rank order; a rank whose node is
    unknown or has no allowed core gets an even slice of everything allowed instead.  Every rank gets at least one core."""
    allowed = sorted(set(int(c) for c in allowed))
    n = len(local_cpus)
    groups: Dict[tuple, List[int]] = {}
    for r, cpus in enumerate(local_cpus):
        key = tuple(c for c in sorted(set(cpus or [])) if c in set(allowed))
        groups.setdefault(key, []).append(r)
    out: List[List[int]] = [[] for _ in range(n)]
    for key, ranks in groups.items():
        pool = list(key)
        if not pool:                                  # unknown: this rank's share of all allowed cores, as if every rank were here
            for r in ranks:
                per = max(1, len(allowed) // n)
                out[r] = allowed[r * per % len(allowed):][:per] or allowed[:1]
            continue
        per = max(1, len(pool) // len(ranks))
        for j, r in enumerate(ranks):
            out[r] = pool[j * per % len(pool):][:per] or pool[:1]
    return out


def bind_rank(local_rank: int, local_world: int, gpu_of_rank=None, sysfs: str = "/sys", apply: bool = True) -> dict:
    """Pin this process (rank `local_rank` of `local_world` on this host) to its cores.  gpu_of_rank: rank -> PCI address of its GPU
    (default: torch.cuda.get_device_properties(rank % device_count)).  Returns what it did: {"cpus": [...], "numa_known": bool, ...}.
    FD_NO_AFFINITY=1 turns it off (the dict then says so)."""
    if os.environ.get("FD_NO_AFFINITY") == "1":
        return {"applied": False, "why": "FD_NO_AFFINITY=1"}
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return {"applied": False, "why": "sched_getaffinity unavailable"}
    addrs = []
    for r in range(local_world):
        addr = None
        try:
            if gpu_of_rank is not None:
                addr = gpu_of_rank(r)
            else:
                import torch
                p = torch.cuda.get_device_properties(r % max(1, torch.cuda.device_count()))
                addr = pci_address(p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        except Exception:      # noqa: BLE001 -- placement is best effort
            addr = None
        addrs.append(addr)
    local = [gpu_local_cpus(a, sysfs) if a else None for a in addrs]
    sets = plan(local, allowed)
    mine = sets[local_rank]
    info = {"applied": False, "cpus": len(mine), "first_cpu": mine[0] if mine else None, "numa_known": local[local_rank] is not None,
            "gpu_pci": addrs[local_rank], "allowed": len(allowed)}
    if apply and mine and len(mine) < len(allowed):
        try:
            os.sched_setaffinity(0, mine)
            info["applied"] = True
        except OSError as e:
            info["why"] = repr(e)
    return info
