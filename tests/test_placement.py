"""Rank placement: each rank's process is bound to the CPUs of its GPU's NUMA node, read from sysfs, and the binding is
reported -- exercised here on a fake sysfs tree, no GPU."""

import os

import pytest

from xgcm_amd import sharding as S


def _fake_sysfs(root, devices, nodes):
    """devices: {bdf: (numa_node, local_cpulist or None)}; nodes: {node: cpulist}"""
    for bdf, (node, cpulist) in devices.items():
        d = root / "bus" / "pci" / "devices" / bdf
        d.mkdir(parents=True)
        (d / "numa_node").write_text(f"{node}\n")
        if cpulist is not None:
            (d / "local_cpulist").write_text(cpulist + "\n")
    for node, cpulist in nodes.items():
        d = root / "devices" / "system" / "node" / f"node{node}"
        d.mkdir(parents=True)
        (d / "cpulist").write_text(cpulist + "\n")
    return str(root)


def test_cpulist_round_trip():
    assert S.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert S.parse_cpulist("") == [] and S.parse_cpulist("5") == [5]
    assert S.format_cpulist([0, 1, 2, 3, 8, 10, 11]) == "0-3,8,10-11"
    assert S.format_cpulist([]) == "" and S.format_cpulist([7, 5, 6]) == "5-7"
    big = "0-63,128-191"
    assert S.format_cpulist(S.parse_cpulist(big)) == big


def test_pci_placement_reads_local_cpulist_then_the_node(tmp_path):
    sysfs = _fake_sysfs(tmp_path, {"0000:c1:00.0": (1, "64-127,192-255"), "0000:05:00.0": (0, None), "0000:09:00.0": (-1, None)},
                        {0: "0-63,128-191", 1: "64-127,192-255"})
    a = S.pci_placement("0000:C1:00.0", sysfs)  # torch prints the bus id in either case
    assert a["numa_node"] == 1 and a["cpus"][:2] == [64, 65] and len(a["cpus"]) == 128
    b = S.pci_placement("0000:05:00.0", sysfs)  # no local_cpulist: the node's cpulist
    assert b["numa_node"] == 0 and len(b["cpus"]) == 128 and b["cpus"][-1] == 191
    c = S.pci_placement("0000:09:00.0", sysfs)  # numa_node -1: a single-node host, nothing to bind
    assert c["numa_node"] == -1 and c["cpus"] == []
    assert S.pci_placement("0000:ff:00.0", sysfs) == {"pci_bus_id": "0000:ff:00.0", "numa_node": None, "cpus": []}
    assert S.pci_placement(None, sysfs)["cpus"] == []


@pytest.mark.skipif(not hasattr(os, "sched_setaffinity"), reason="no sched_setaffinity")
def test_bind_applies_reports_and_never_widens(tmp_path):
    allowed = sorted(os.sched_getaffinity(0))
    try:
        keep = allowed[: max(1, len(allowed) // 2)]
        sysfs = _fake_sysfs(tmp_path, {"0000:c1:00.0": (1, S.format_cpulist(keep + [100000]))}, {})  # one CPU we do not have
        rep = S.bind_to_gpu_numa(3, sysfs=sysfs, pci_bus_id="0000:c1:00.0", apply=True)
        assert rep["bound"] and rep["local_rank"] == 3 and rep["numa_node"] == 1 and rep["pci_bus_id"] == "0000:c1:00.0"
        assert sorted(os.sched_getaffinity(0)) == keep and rep["affinity"] == S.format_cpulist(keep)
        assert rep["affinity_before"] == S.format_cpulist(allowed) and rep["n_cpus"] == len(keep) + 1
        os.sched_setaffinity(0, allowed)
        off = S.bind_to_gpu_numa(0, sysfs=sysfs, pci_bus_id="0000:c1:00.0", apply=False)  # XG_NUMA_BIND=0: reported, not applied
        assert not off["bound"] and sorted(os.sched_getaffinity(0)) == allowed and off["cpus"] == rep["cpus"]
        none = S.bind_to_gpu_numa(0, sysfs=sysfs, pci_bus_id="0000:00:00.0", apply=True)  # unknown device: nothing happens
        assert not none["bound"] and none["n_cpus"] == 0 and sorted(os.sched_getaffinity(0)) == allowed
    finally:
        os.sched_setaffinity(0, allowed)


@pytest.mark.skipif(not hasattr(os, "sched_setaffinity") or not os.path.isdir("/proc/self/task"), reason="needs /proc and sched_setaffinity")
def test_bind_moves_threads_that_already_exist(tmp_path):
    """`sched_setaffinity(0, ...)` binds the calling thread only; a pool thread started BEFORE the binding must
    end up on the GPU's CPUs too (and come back when bench.py restores the mask for its CPU-baseline leg)"""
    import threading

    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) < 2:
        pytest.skip("one CPU: nothing to narrow")
    stop, seen = threading.Event(), {}
    ready = threading.Event()

    def worker():
        seen["tid"] = threading.get_native_id()
        ready.set()
        stop.wait()

    th = threading.Thread(target=worker)
    th.start()
    ready.wait()
    try:
        keep = allowed[:1]
        sysfs = _fake_sysfs(tmp_path, {"0000:c1:00.0": (0, S.format_cpulist(keep))}, {})
        rep = S.bind_to_gpu_numa(0, sysfs=sysfs, pci_bus_id="0000:c1:00.0", apply=True)
        assert rep["bound"] and rep["threads_bound"] >= 2
        assert sorted(os.sched_getaffinity(seen["tid"])) == keep      # the thread that existed before the binding
        assert S.set_affinity_all_threads(allowed) >= 2
        assert sorted(os.sched_getaffinity(seen["tid"])) == allowed
    finally:
        stop.set()
        th.join()
        S.set_affinity_all_threads(allowed)


def test_env_switch(monkeypatch, tmp_path):
    allowed = sorted(os.sched_getaffinity(0))
    sysfs = _fake_sysfs(tmp_path, {"0000:c1:00.0": (0, S.format_cpulist(allowed[:1]))}, {})
    monkeypatch.setenv("XG_NUMA_BIND", "0")
    try:
        rep = S.bind_to_gpu_numa(0, sysfs=sysfs, pci_bus_id="0000:c1:00.0")
        assert not rep["bound"] and sorted(os.sched_getaffinity(0)) == allowed
    finally:
        os.sched_setaffinity(0, allowed)


def test_single_process_ranks_carry_a_placement_record():
    ranks = S.init_ranks(1, backend="gloo")
    assert ranks.world == 1 and ranks.placement["local_rank"] == 0 and "bound" in ranks.placement
    assert ranks.gather_objects({"a": 1}) == [{"a": 1}]
