"""Where the per-GPU workers of an N-rank run execute on the host.

The reference leaves this to Lightning / the OS (deepchopper/cli.py:126-147).  An MI355X node has 8 GPUs behind two
sockets (xGMI between the GPUs, PCIe to one socket each) and 256 hardware threads; a worker needs about five of them
(2.3 for the BGZF deflate of its output, the rest for staging, launches and Python: profiles/r03_host_budget_8rank.json)
and the parent -- the node's single inflater -- about six.  Left alone, eight workers and their helper threads wander
over both sockets, pinned staging lands on whichever node the first touch happened, and the first torch CPU op of every
worker starts one thread per hardware thread (8 x 256).  Here:

* `gpu_numa_nodes()` reads which NUMA node each GPU hangs off, from sysfs only (KFD topology -> DRM render node ->
  `numa_node`; the parent must never touch the GPU, so no HIP call);
* `plan()` gives every rank a DISJOINT slice of its GPU's node's CPUs (`cores_per_rank`, default 5) and the parent a
  slice of what is left (`parent_cores`, default 6), within the CPUs this process may use at all;
* `apply()` -- called by a worker before it imports torch or touches the GPU -- binds the process
  (`os.sched_setaffinity`: the parser / BGZF / copy threads started later inherit the mask, and pinned staging is
  allocated after binding, i.e. on the node of the CPUs that fill it) and bounds OpenMP / torch intra-op threads to
  the slice.

`DC_CPU_PLACEMENT=0` switches it off; `DC_CORES_PER_RANK`, `DC_PARENT_CORES` resize it.  Single-GPU runs are left alone.
"""
from __future__ import annotations

import glob
import os
import re

DEFAULT_CORES_PER_RANK = 5
DEFAULT_PARENT_CORES = 6


def parse_cpulist(text: str) -> list[int]:
    """"0-3,8,10-11" -> [0, 1, 2, 3, 8, 10, 11] (the kernel's cpulist format)."""
    out: list[int] = []
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-", 1)
            out.extend(range(int(a), int(b) + 1))
        else:
            out.append(int(part))
    return sorted(set(out))


def format_cpulist(cpus) -> str:
    cpus = sorted(set(int(c) for c in cpus))
    runs, i = [], 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        runs.append(str(cpus[i]) if i == j else f"{cpus[i]}-{cpus[j]}")
        i = j + 1
    return ",".join(runs)


def _read(path: str) -> str | None:
    try:
        with open(path) as fh:
            return fh.read()
    except OSError:
        return None


def cpu_quota(root: str = "/sys/fs/cgroup") -> float | None:
    """CPUs' worth of time the cgroup allows (cgroup v2 `cpu.max`, v1 `cpu.cfs_quota_us`), or None when unlimited.  A GPU
    box of the pool gives a job a share of the host this way while the affinity mask still shows every hardware thread."""
    txt = _read(os.path.join(root, "cpu.max"))
    if txt:
        q, _, p = txt.strip().partition(" ")
        if q != "max" and p:
            return int(q) / int(p)
        return None
    q, p = _read(os.path.join(root, "cpu", "cpu.cfs_quota_us")), _read(os.path.join(root, "cpu", "cpu.cfs_period_us"))
    if q and p and int(q) > 0:
        return int(q) / int(p)
    return None


def usable_cpu_count(cgroup_root: str | None = None, n_allowed: int | None = None) -> int:
    """How many CPUs a CPU-bound job here can really keep busy: the affinity mask capped by the cgroup quota."""
    n = n_allowed if n_allowed is not None else (
        len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    q = cpu_quota(cgroup_root or os.environ.get("DC_CGROUP_ROOT", "/sys/fs/cgroup"))  # (tests point this at a fake tree)
    return max(1, min(n, int(q + 0.5))) if q else n


def _visible_indices(n: int) -> list[int]:
    """HIP device index -> KFD GPU ordinal, honouring ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES when they are plain index
    lists (UUID forms are left alone: then the order is the topology's)."""
    order = list(range(n))
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var, "").strip()
        if v and re.fullmatch(r"\d+(,\d+)*", v):
            order = [order[i] for i in (int(x) for x in v.split(",")) if i < len(order)]
    return order


def gpu_numa_nodes(sysfs: str = "/sys") -> list[int]:
    """NUMA node of every GPU in HIP device order (-1: unknown or a single-node host).  sysfs only:
    /sys/class/kfd/kfd/topology/nodes/N/properties lists the compute nodes in the order the runtime enumerates them; a
    node with `simd_count` > 0 is a GPU and names its DRM render node (`drm_render_minor`) and PCI location
    (`domain`, `location_id`), whose device directory has `numa_node`.  Fallback: the AMD display-class PCI functions of
    /sys/class/drm/card*, by PCI address."""
    nodes = []
    for d in sorted(glob.glob(os.path.join(sysfs, "class/kfd/kfd/topology/nodes/*")), key=lambda p: int(os.path.basename(p))):
        props = {}
        for line in (_read(os.path.join(d, "properties")) or "").splitlines():
            k, _, v = line.partition(" ")
            if v.strip().lstrip("-").isdigit():
                props[k] = int(v)
        if props.get("simd_count", 0) <= 0:
            continue
        numa = None
        minor = props.get("drm_render_minor", -1)
        if minor >= 0:
            numa = _read(os.path.join(sysfs, f"class/drm/renderD{minor}/device/numa_node"))
        if numa is None and "location_id" in props:
            loc, dom = props["location_id"], props.get("domain", 0)
            addr = f"{dom:04x}:{(loc >> 8) & 0xFF:02x}:{(loc >> 3) & 0x1F:02x}.{loc & 7}"
            numa = _read(os.path.join(sysfs, f"bus/pci/devices/{addr}/numa_node"))
        nodes.append(int(numa) if numa is not None and numa.strip().lstrip("-").isdigit() else -1)
    if not nodes:
        cards = []
        for c in glob.glob(os.path.join(sysfs, "class/drm/card[0-9]*")):
            if not re.fullmatch(r"card\d+", os.path.basename(c)):
                continue
            dev = os.path.join(c, "device")
            if (_read(os.path.join(dev, "vendor")) or "").strip() != "0x1002":
                continue
            addr = os.path.basename(os.path.realpath(dev))
            numa = _read(os.path.join(dev, "numa_node"))
            cards.append((addr, int(numa) if numa and numa.strip().lstrip("-").isdigit() else -1))
        nodes = [n for _, n in sorted(cards)]
    return [nodes[i] for i in _visible_indices(len(nodes))]


def node_cpus(node: int, sysfs: str = "/sys") -> list[int] | None:
    txt = _read(os.path.join(sysfs, f"devices/system/node/node{node}/cpulist")) if node >= 0 else None
    return parse_cpulist(txt) if txt else None


def plan(world: int, *, local_devices: list[int] | None = None, sysfs: str | None = None, allowed: list[int] | None = None,
         cores_per_rank: int | None = None, parent_cores: int | None = None, cgroup_root: str | None = None) -> dict:
    """CPU slices for the `world` workers of THIS node and for their parent.

    local_devices[r] = HIP device index of local rank r (default r; all 0 when ranks share one GPU).  Returns
    {"ranks": [{"cpus": [...], "numa_node": n, "device": d}, ...], "parent": {"cpus": [...]} | None, "cpu_budget": b,
    "capped": bool}.  Slices are disjoint whenever the allowed CPUs suffice (world * 1 <= CPUs of the node); ranks of one node
    split that node's CPUs evenly when `cores_per_rank` each do not fit.

    The plan fits the CPU TIME the job really has: the affinity mask of a GPU box of the pool shows all 256 hardware threads
    while its cgroup quota (`cpu.max`) is 16 CPUs' worth -- 8 x 5 + 6 = 46 pinned, busy threads would be throttled, not
    helped.  `world * cores_per_rank + parent_cores` is capped at `usable_cpu_count()`: the per-rank slices shrink first
    (never below one CPU, never overlapping), the parent keeps what is left of the budget (at least two CPUs while the
    ranks can still have one each: it is the node's single inflater).  Deterministic: every rank of an externally
    launched job computes the same plan."""
    cores_per_rank = int(os.environ.get("DC_CORES_PER_RANK", cores_per_rank or DEFAULT_CORES_PER_RANK))
    parent_cores = int(os.environ.get("DC_PARENT_CORES", parent_cores if parent_cores is not None else DEFAULT_PARENT_CORES))
    sysfs = sysfs or os.environ.get("DC_SYSFS_ROOT", "/sys")  # (tests point this at a fake tree)
    if allowed is None:
        allowed = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    budget = usable_cpu_count(cgroup_root, len(allowed))
    # (a job limited by its affinity mask alone is handled below: the ranks of a node split what the mask leaves of it)
    capped = budget < len(allowed) and world * cores_per_rank + parent_cores > budget
    if capped:
        keep = min(parent_cores, 2) if budget >= world + 2 else 0
        cores_per_rank = max(1, min(cores_per_rank, (budget - keep) // world))
        parent_cores = min(parent_cores, max(budget - world * cores_per_rank, 0))
    allowed_set = set(allowed)
    local_devices = list(range(world)) if local_devices is None else list(local_devices)
    numa = gpu_numa_nodes(sysfs)
    rank_node = [numa[d] if 0 <= d < len(numa) else -1 for d in local_devices]
    pools: dict[int, list[int]] = {}
    for n in sorted(set(rank_node)):
        c = node_cpus(n, sysfs)
        c = [x for x in c if x in allowed_set] if c else []
        pools[n] = c if c else list(allowed)  # unknown node (or none of its CPUs allowed): anywhere we may run
    # two pools that fell back to "anywhere" (or a real node and the fallback) must not hand out the same CPU twice
    taken: set[int] = set()
    ranks = [None] * world
    for n in sorted(pools):
        members = [r for r in range(world) if rank_node[r] == n]
        free = [c for c in pools[n] if c not in taken]
        if not free:
            free = list(pools[n])
        per = max(1, min(cores_per_rank, len(free) // len(members)))
        for i, r in enumerate(members):
            sl = free[i * per: (i + 1) * per] or [free[i % len(free)]]  # more ranks than CPUs: overlap, never empty
            ranks[r] = {"cpus": sl, "numa_node": n, "device": local_devices[r]}
            taken.update(sl)
    left = [c for c in allowed if c not in taken]
    parent = {"cpus": left[:parent_cores]} if parent_cores > 0 and len(left) >= 1 else None
    return {"ranks": ranks, "parent": parent, "cpu_budget": budget, "capped": capped}


def enabled() -> bool:
    return os.environ.get("DC_CPU_PLACEMENT", "1") != "0" and hasattr(os, "sched_setaffinity")


def apply(cpus: list[int]) -> dict:
    """Bind this process (and every thread it starts from now on) to `cpus`; bound OpenMP / torch threads to the slice.
    Call before importing torch and before the first HIP call.  Returns what was done, for the worker's report."""
    cpus = sorted(set(cpus) & set(os.sched_getaffinity(0))) or sorted(os.sched_getaffinity(0))
    os.sched_setaffinity(0, cpus)
    n = str(len(cpus))
    for var in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
        os.environ[var] = n
    import sys

    if "torch" in sys.modules:  # imported already (a test, an embedding application): bound its pools directly
        import torch

        torch.set_num_threads(len(cpus))
    return {"cpuset": format_cpulist(cpus), "n_cpus": len(cpus)}


def apply_for_this_rank(sysfs: str | None = None) -> dict | None:
    """Worker entry: take the slice the command's own launcher put into DC_CPUSET, or -- under an external launcher
    (torch.distributed.run sets LOCAL_RANK / LOCAL_WORLD_SIZE) -- compute the node's plan and take this rank's slice.
    Nothing happens for a single-process run or with DC_CPU_PLACEMENT=0."""
    if not enabled():
        return None
    try:
        return _apply_for_this_rank(sysfs)
    except Exception:  # an unreadable topology or a refused affinity call is never a reason to fail a run
        return None


def _apply_for_this_rank(sysfs: str | None) -> dict | None:
    given = os.environ.get("DC_CPUSET")
    if given:
        info = apply(parse_cpulist(given))
        info["numa_node"] = int(os.environ.get("DC_NUMA_NODE", "-1"))
        if os.environ.get("DC_CPU_BUDGET"):  # the launcher's plan: CPUs' worth of time the job has, and whether the slices were cut to it
            info["cpu_budget"], _, cut = os.environ["DC_CPU_BUDGET"].partition(":")
            info["cpu_budget"], info["capped_to_budget"] = int(info["cpu_budget"]), cut == "capped"
        return info
    lws = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    if lws <= 1:
        return None
    lr = int(os.environ.get("LOCAL_RANK", "0"))
    share = os.environ.get("DC_SHARE_GPU") == "1" or os.environ.get("DC_BENCH_SHARE_GPU") == "1"
    p = plan(lws, local_devices=[0] * lws if share else None, sysfs=sysfs)
    info = apply(p["ranks"][lr]["cpus"])
    info["numa_node"] = p["ranks"][lr]["numa_node"]
    info["cpu_budget"], info["capped_to_budget"] = p["cpu_budget"], p["capped"]
    return info
