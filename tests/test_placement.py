"""CPU placement of the per-GPU workers (deepchopper_amd/placement.py) against a fake sysfs tree: a two-socket host with
four GPUs per socket, as an 8 x MI355X node is built.  No GPU, no real sysfs."""
import os
import subprocess
import sys

import pytest

from deepchopper_amd import launcher, placement

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(autouse=True)
def _no_cgroup_quota(tmp_path, monkeypatch):
    """The plan reads the cgroup CPU quota of the box it runs on; these tests bring their own (or none)."""
    d = tmp_path / "cgroup_none"
    d.mkdir()
    monkeypatch.setenv("DC_CGROUP_ROOT", str(d))


def make_sysfs(root, gpus_per_node=4, with_kfd=True, cpulists=("0-63,128-191", "64-127,192-255")):
    """KFD topology: two CPU nodes (simd_count 0) then the GPUs in enumeration order, each pointing at renderD<minor>, whose
    PCI device names its NUMA node -- deliberately NOT in PCI-address order, so that the mapping must come from the topology."""
    n_nodes = len(cpulists)
    for n, cl in enumerate(cpulists):
        d = root / "devices/system/node" / f"node{n}"
        d.mkdir(parents=True)
        (d / "cpulist").write_text(cl + "\n")
    numa_of = [(g // gpus_per_node + 1) % n_nodes for g in range(n_nodes * gpus_per_node)]  # GPUs 0-3 on node 1, 4-7 on node 0
    for g, numa in enumerate(numa_of):
        addr = f"0000:{0x05 + 0x10 * ((g * 3) % 8):02x}:00.0"
        pci = root / "bus/pci/devices" / addr
        pci.mkdir(parents=True)
        (pci / "numa_node").write_text(f"{numa}\n")
        (pci / "vendor").write_text("0x1002\n")
        for name in (f"renderD{128 + g}", f"card{g + 1}"):
            dd = root / "class/drm" / name
            dd.mkdir(parents=True)
            os.symlink(pci, dd / "device")
    if with_kfd:
        topo = root / "class/kfd/kfd/topology/nodes"
        for n in range(n_nodes):
            (topo / str(n)).mkdir(parents=True)
            (topo / str(n) / "properties").write_text("cpu_cores_count 64\nsimd_count 0\ndrm_render_minor -1\n")
        for g in range(len(numa_of)):
            d = topo / str(n_nodes + g)
            d.mkdir(parents=True)
            (d / "properties").write_text(f"cpu_cores_count 0\nsimd_count 1024\ndrm_render_minor {128 + g}\ndomain 0\nlocation_id 1280\n")
    return numa_of


def test_cpulist_round_trip():
    assert placement.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert placement.format_cpulist([11, 0, 1, 2, 3, 8, 10]) == "0-3,8,10-11"
    assert placement.parse_cpulist("") == [] and placement.format_cpulist([]) == ""


def test_gpu_numa_nodes_from_kfd_topology_and_from_drm_cards(tmp_path, monkeypatch):
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    want = make_sysfs(tmp_path / "a")
    assert placement.gpu_numa_nodes(str(tmp_path / "a")) == want == [1, 1, 1, 1, 0, 0, 0, 0]
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "5,2")
    assert placement.gpu_numa_nodes(str(tmp_path / "a")) == [0, 1]
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    make_sysfs(tmp_path / "b", with_kfd=False)  # no KFD topology: AMD DRM cards by PCI address
    got = placement.gpu_numa_nodes(str(tmp_path / "b"))
    assert sorted(got) == [0, 0, 0, 0, 1, 1, 1, 1]
    assert placement.gpu_numa_nodes(str(tmp_path / "nothing")) == []


def test_plan_eight_ranks_two_sockets_disjoint_slices_on_the_right_node(tmp_path, monkeypatch):
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "DC_CORES_PER_RANK", "DC_PARENT_CORES"):
        monkeypatch.delenv(var, raising=False)
    numa = make_sysfs(tmp_path)
    p = placement.plan(8, sysfs=str(tmp_path), allowed=list(range(256)))
    node_cpus = {0: set(placement.parse_cpulist("0-63,128-191")), 1: set(placement.parse_cpulist("64-127,192-255"))}
    seen = set()
    for r, sl in enumerate(p["ranks"]):
        assert sl["numa_node"] == numa[r] and sl["device"] == r
        assert len(sl["cpus"]) == placement.DEFAULT_CORES_PER_RANK and set(sl["cpus"]) <= node_cpus[numa[r]]
        assert not (seen & set(sl["cpus"]))
        seen |= set(sl["cpus"])
    assert len(p["parent"]["cpus"]) == placement.DEFAULT_PARENT_CORES and not (seen & set(p["parent"]["cpus"]))
    # the measured budget: 8 x 5 + 6 = 46 of 256 hardware threads (profiles/r03_host_budget_8rank.json: 37-41 cores)
    assert len(seen) + len(p["parent"]["cpus"]) == 46
    # a job that may only use 12 CPUs of node 1 and 2 of node 0: even split, still disjoint, never empty
    p = placement.plan(8, sysfs=str(tmp_path), allowed=list(range(64, 76)) + [0, 1])
    cp = [tuple(sl["cpus"]) for sl in p["ranks"]]
    assert all(len(c) == 3 for c in cp[:4]) and len(set(sum(cp[:4], ()))) == 12
    assert all(len(c) >= 1 for c in cp[4:])  # 4 ranks on 2 CPUs: shared, said as such by the overlap
    # ranks sharing ONE GPU (the one-GPU rehearsal): disjoint slices of that GPU's node
    p = placement.plan(3, local_devices=[0, 0, 0], sysfs=str(tmp_path), allowed=list(range(256)))
    cp = [set(sl["cpus"]) for sl in p["ranks"]]
    assert all(sl["numa_node"] == 1 for sl in p["ranks"]) and not (cp[0] & cp[1]) and not (cp[1] & cp[2]) and not (cp[0] & cp[2])


def test_plan_without_any_topology_still_gives_disjoint_slices():
    p = placement.plan(2, sysfs="/nonexistent", allowed=list(range(8)), cores_per_rank=3, parent_cores=2)
    a, b = (set(sl["cpus"]) for sl in p["ranks"])
    assert len(a) == len(b) == 3 and not (a & b) and all(sl["numa_node"] == -1 for sl in p["ranks"])
    assert len(p["parent"]["cpus"]) == 2 and not (set(p["parent"]["cpus"]) & (a | b))


WORKER = """
import json, os, sys
sys.path.insert(0, %r)
from deepchopper_amd import placement
info = placement.apply_for_this_rank()
import torch
import threading
seen = []
t = threading.Thread(target=lambda: seen.append(sorted(os.sched_getaffinity(0))))  # a helper thread inherits the mask
t.start(); t.join()
print(json.dumps({"rank": int(os.environ["RANK"]), "info": info, "affinity": sorted(os.sched_getaffinity(0)),
                  "thread_affinity": seen[0], "torch_threads": torch.get_num_threads(), "omp": os.environ.get("OMP_NUM_THREADS")}))
""" % ROOT


@pytest.mark.skipif(not hasattr(os, "sched_setaffinity") or len(os.sched_getaffinity(0)) < 4, reason="needs >= 4 usable CPUs")
@pytest.mark.parametrize("how", ["own launcher", "external launcher"])
def test_two_workers_bind_to_disjoint_cpusets(tmp_path, monkeypatch, how):
    """The command's own launcher hands every worker its slice (DC_CPUSET); under an external launcher (LOCAL_RANK /
    LOCAL_WORLD_SIZE only) every rank computes the same plan and takes its own.  Either way the workers of one node end up
    on disjoint CPUs, their helper threads inherit the mask, and torch's intra-op pool is bounded by the slice."""
    monkeypatch.setenv("DC_CORES_PER_RANK", "2")
    monkeypatch.setenv("DC_PARENT_CORES", "0")
    monkeypatch.delenv("DC_CPUSET", raising=False)
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    if how == "own launcher":
        procs = launcher.spawn_ranks([sys.executable, str(script)], 2, share_gpu=True)
    else:
        procs = []
        for r in range(2):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), LOCAL_WORLD_SIZE="2", DC_SHARE_GPU="1")
            procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, text=True))
    reps = [launcher.last_json_line(o) for o in launcher.wait_ranks(procs, timeout=120)]
    a, b = (set(r["affinity"]) for r in reps)
    assert len(a) == len(b) == 2 and not (a & b)
    for r in reps:
        assert r["thread_affinity"] == r["affinity"] and r["torch_threads"] == 2 and r["omp"] == "2"
        assert placement.parse_cpulist(r["info"]["cpuset"]) == r["affinity"]


def test_single_process_and_switched_off_runs_are_left_alone(monkeypatch):
    for var in ("DC_CPUSET", "LOCAL_WORLD_SIZE", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(var, raising=False)
    before = sorted(os.sched_getaffinity(0))
    assert placement.apply_for_this_rank() is None
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("DC_CPU_PLACEMENT", "0")
    assert placement.apply_for_this_rank() is None and launcher.cpu_plan(2) is None
    assert sorted(os.sched_getaffinity(0)) == before


def test_plan_fits_the_cgroup_cpu_quota(tmp_path, monkeypatch):
    """a GPU box of the pool shows 256 CPUs in the affinity mask and has a cgroup quota of 16 CPUs' worth
    of time.  8 ranks x 5 + 6 = 46 pinned busy threads would be throttled: the plan shrinks the per-rank slices (never
    below one CPU, never overlapping) before anything overlaps, the parent keeps what is left of the budget, every slice
    stays on its GPU's NUMA node, and the cap is reported."""
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "DC_CORES_PER_RANK", "DC_PARENT_CORES"):
        monkeypatch.delenv(var, raising=False)
    numa_of = make_sysfs(tmp_path / "sys")
    allowed = list(range(256))
    node = {0: set(placement.parse_cpulist("0-63,128-191")), 1: set(placement.parse_cpulist("64-127,192-255"))}

    def quota(tag, text):
        d = tmp_path / tag
        d.mkdir()
        (d / "cpu.max").write_text(text)
        return str(d)

    def check(p, world, per_lo, per_hi, parent_n, budget):
        assert p["cpu_budget"] == budget
        seen = set()
        for r, sl in enumerate(p["ranks"]):
            assert per_lo <= len(sl["cpus"]) <= per_hi, (r, sl)
            assert not (set(sl["cpus"]) & seen), "rank slices must be disjoint"
            assert set(sl["cpus"]) <= node[numa_of[r]], "a rank stays on its GPU's NUMA node"
            seen |= set(sl["cpus"])
        par = set(p["parent"]["cpus"]) if p["parent"] else set()
        assert len(par) == parent_n and not (par & seen)
        if p["capped"]:
            assert len(seen) + len(par) <= budget

    # the pool's box: quota 16 of 256 -> 8 disjoint single-CPU slices, the parent's six inside the remaining eight
    p = placement.plan(8, sysfs=str(tmp_path / "sys"), allowed=allowed, cgroup_root=quota("q16", "1600000 100000\n"))
    assert p["capped"]
    check(p, 8, 1, 1, 6, 16)
    # quota 32: three CPUs per rank (8 x 3 = 24), parent 6, 30 <= 32
    p = placement.plan(8, sysfs=str(tmp_path / "sys"), allowed=allowed, cgroup_root=quota("q32", "3200000 100000\n"))
    assert p["capped"]
    check(p, 8, 3, 3, 6, 32)
    # quota 8 with 8 ranks: one CPU each, nothing left for a pinned parent
    p = placement.plan(8, sysfs=str(tmp_path / "sys"), allowed=allowed, cgroup_root=quota("q8", "800000 100000\n"))
    assert p["capped"] and p["parent"] is None
    check(p, 8, 1, 1, 0, 8)
    # no quota ("max") and a generous one: the full plan, not capped
    for tag, text, budget in (("qmax", "max 100000\n", 256), ("q64", "6400000 100000\n", 64)):
        p = placement.plan(8, sysfs=str(tmp_path / "sys"), allowed=allowed, cgroup_root=quota(tag, text))
        assert not p["capped"]
        check(p, 8, 5, 5, 6, budget)
    # the worker's report carries the cap (self-launched: DC_CPU_BUDGET travels with DC_CPUSET)
    p = placement.plan(8, sysfs=str(tmp_path / "sys"), allowed=allowed, cgroup_root=quota("q16b", "1600000 100000\n"))
    env = launcher.rank_env(3, 8, 29500, plan=p)
    assert env["DC_CPU_BUDGET"] == "16:capped" and env["OMP_NUM_THREADS"] == "1"
    assert placement.usable_cpu_count(quota("q16c", "1600000 100000\n"), 256) == 16
