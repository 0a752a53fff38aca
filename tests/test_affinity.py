"""Host logic of the per-rank CPU placement (fastdiff_amd/affinity.py): one process per GPU as the reference starts them
(utils/trainer.py:94-107), each pinned to its slice of the cores next to its GPU.  No GPU needed: sysfs is a temporary directory."""
import os

from fastdiff_amd import affinity


def fake_sysfs(root, gpus, nodes):
    """gpus: {pci address: numa node}; nodes: {node: cpulist text}."""
    for addr, node in gpus.items():
        d = root / "bus" / "pci" / "devices" / addr
        d.mkdir(parents=True)
        (d / "numa_node").write_text("%d\n" % node)
        (d / "local_cpulist").write_text(nodes.get(node, "") + "\n")
    for node, text in nodes.items():
        d = root / "devices" / "system" / "node" / ("node%d" % node)
        d.mkdir(parents=True)
        (d / "cpulist").write_text(text + "\n")
    return str(root)


def test_cpulist_format():
    assert affinity.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert affinity.parse_cpulist("") == [] and affinity.parse_cpulist("5") == [5]
    assert affinity.pci_address(0, 0xC5, 0) == "0000:c5:00.0"


def test_eight_gpus_on_two_sockets_get_disjoint_slices_of_their_own_socket(tmp_path):
    gpus = {affinity.pci_address(0, 0x10 + r, 0): (0 if r < 4 else 1) for r in range(8)}
    sysfs = fake_sysfs(tmp_path, gpus, {0: "0-63,128-191", 1: "64-127,192-255"})
    local = [affinity.gpu_local_cpus(a, sysfs) for a in gpus]
    sets = affinity.plan(local, range(256))
    assert all(len(s) == 32 for s in sets)
    for r, s in enumerate(sets):
        assert set(s) <= set(local[r])                                   # on the GPU's own socket
        for q in range(r):
            assert not set(s) & set(sets[q])                             # and nobody else's cores
    assert set().union(*sets) == set(range(256))


def test_unknown_topology_falls_back_to_an_even_split(tmp_path):
    sysfs = fake_sysfs(tmp_path, {"0000:01:00.0": -1}, {})
    assert affinity.gpu_local_cpus("0000:01:00.0", sysfs) is None
    assert affinity.gpu_local_cpus("0000:99:00.0", sysfs) is None        # no such function
    sets = affinity.plan([None] * 4, range(8))
    assert sets == [[0, 1], [2, 3], [4, 5], [6, 7]]
    # a cgroup that allows fewer cores than the node has: only allowed cores are handed out, and every rank still gets one
    sets = affinity.plan([[0, 1, 2, 3]] * 2 + [[4, 5, 6, 7]] * 2, [0, 1, 4])
    assert sets[0] == [0] and sets[1] == [1] and sets[2] == [4] and sets[3] == [4]


def test_bind_rank_reports_what_it_would_do(tmp_path, monkeypatch):
    gpus = {affinity.pci_address(0, 0x20 + r, 0): r % 2 for r in range(2)}
    allowed = sorted(os.sched_getaffinity(0))
    half = max(1, len(allowed) // 2)
    sysfs = fake_sysfs(tmp_path, gpus, {0: ",".join(map(str, allowed[:half])), 1: ",".join(map(str, allowed[half:] or allowed[:1]))})
    addr = list(gpus)
    info = affinity.bind_rank(1, 2, gpu_of_rank=lambda r: addr[r], sysfs=sysfs, apply=False)
    assert info["applied"] is False and info["numa_known"] and info["gpu_pci"] == addr[1] and info["cpus"] >= 1
    monkeypatch.setenv("FD_NO_AFFINITY", "1")
    assert affinity.bind_rank(0, 2, gpu_of_rank=lambda r: addr[r], sysfs=sysfs) == {"applied": False, "why": "FD_NO_AFFINITY=1"}


def test_bind_rank_really_pins_the_process(tmp_path):
    """In a child process (so that pytest keeps its own cores): rank 1 of 2 ends up on exactly the cores the plan hands it."""
    import subprocess
    import sys
    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) < 2:
        import pytest
        pytest.skip("needs two cores")
    half = len(allowed) // 2
    gpus = {affinity.pci_address(0, 0x30 + r, 0): r for r in range(2)}
    sysfs = fake_sysfs(tmp_path, gpus, {0: ",".join(map(str, allowed[:half])), 1: ",".join(map(str, allowed[half:]))})
    code = ("import os, sys, json; sys.path.insert(0, %r); from fastdiff_amd import affinity; addr = %r; "
            "info = affinity.bind_rank(1, 2, gpu_of_rank=lambda r: addr[r], sysfs=%r); "
            "print(json.dumps({'info': info, 'now': sorted(os.sched_getaffinity(0))}))") % (
                os.path.dirname(os.path.dirname(os.path.abspath(__file__))), list(gpus), sysfs)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    import json
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert res["info"]["applied"] is True and res["now"] == allowed[half:]
