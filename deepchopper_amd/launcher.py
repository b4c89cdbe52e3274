"""One process per GPU, started by the command itself (`deepchopper predict --gpus N`, `bench.py --gpus N`).

The reference launches its own per-device workers through Lightning (`accelerator="gpu", devices=N`,
deepchopper/cli.py:126-147).  Here the parent process -- which never touches the GPU, so nothing that has
initialised HIP is ever forked or exec'ed -- starts N fresh interpreters with RANK / WORLD_SIZE /
LOCAL_RANK / MASTER_* in their environment (the variables `torch.distributed.run` would set, so a worker
cannot tell the two launchers apart), waits for them and merges what they report.  Reads shard with no
data-path collective, so there is nothing else to coordinate.

Plain gzip cannot be entered mid-stream, and eight ranks each inflating the whole file would spend
8 x the CPU to go no faster than one zlib stream.  `ChunkFeeder` is the node's single inflater for that
case: the parent parses the file ONCE with the native reader and deals the packed chunks round-robin to the
workers over pipes (`PipeChunks` on the other end yields `FastqChunk`s exactly like a local reader).
"""
from __future__ import annotations

import json
import os
import socket
import struct
import subprocess
import sys
import threading
from typing import Iterator

import numpy as np

from .fastq import FastqChunk, FastqReader

_MAGIC = 0x44434651  # "DCFQ"
_HDR = struct.Struct("<IIqqq")  # magic, has_name_len, n, seq bytes, header bytes


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def cpu_plan(world: int, share_gpu: bool = False) -> dict | None:
    """Where the workers and their parent run on the host (placement.plan): one disjoint slice of its GPU's NUMA node's
    CPUs per rank.  None when placement is switched off or cannot be computed (never a reason to fail a run)."""
    from . import placement

    if not placement.enabled() or world < 2:
        return None
    try:
        return placement.plan(world, local_devices=[0] * world if share_gpu else None)
    except Exception:
        return None


def rank_env(rank: int, world: int, port: int, share_gpu: bool = False, plan: dict | None = None) -> dict:
    """Environment of worker `rank`: what torch.distributed.run would export.  share_gpu: every worker uses
    device 0 (rehearsing the N > 1 path on a one-GPU box).  `plan` (cpu_plan): the worker's CPU slice travels as
    DC_CPUSET / DC_NUMA_NODE and is applied by the worker before it imports torch or touches the GPU
    (placement.apply_for_this_rank); OMP_NUM_THREADS bounds torch's intra-op pool from its first CPU op on."""
    from . import placement

    env = dict(os.environ)
    env.update({"RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": "0" if share_gpu else str(rank),
                "LOCAL_WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                "DC_SELF_LAUNCHED": "1"})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if plan is not None:
        sl = plan["ranks"][rank]
        env.update({"DC_CPUSET": placement.format_cpulist(sl["cpus"]), "DC_NUMA_NODE": str(sl["numa_node"]),
                    "OMP_NUM_THREADS": str(len(sl["cpus"])),
                    "DC_CPU_BUDGET": f"{plan.get('cpu_budget', 0)}:{'capped' if plan.get('capped') else 'fits'}"})
    return env


def bind_parent(plan: dict | None) -> dict | None:
    """The parent of a self-launched run (the node's single inflater) takes its own slice -- AFTER the workers were spawned,
    so they do not inherit it."""
    from . import placement

    if plan is None or not plan.get("parent"):
        return None
    return placement.apply(plan["parent"]["cpus"])


def spawn_ranks(argv: list[str], world: int, *, share_gpu: bool = False, extra_env: dict | None = None,
                pass_fds_per_rank: list[tuple[int, ...]] | None = None, cwd: str | None = None,
                capture_stdout: bool = True) -> list[subprocess.Popen]:
    """Start `world` workers running `argv` (rank r gets rank_env(r)).  The caller must not have initialised
    the GPU.  stdout of every worker is piped back (its last line is the worker's JSON report)."""
    port = free_port()
    procs = []
    plan = cpu_plan(world, share_gpu)
    for r in range(world):
        env = rank_env(r, world, port, share_gpu, plan)
        if extra_env:
            env.update(extra_env)
        fds = pass_fds_per_rank[r] if pass_fds_per_rank else ()
        procs.append(subprocess.Popen(argv, env=env, cwd=cwd, pass_fds=fds,
                                      stdout=subprocess.PIPE if capture_stdout else None, text=capture_stdout))
    return procs


def wait_ranks(procs: list[subprocess.Popen], timeout: float | None = None) -> list[str]:
    """Wait for every worker; returns their stdout.  A failing worker terminates the others and raises."""
    outs = [""] * len(procs)
    threads = []

    def drain(i, p):
        outs[i] = p.stdout.read() if p.stdout else ""

    for i, p in enumerate(procs):
        t = threading.Thread(target=drain, args=(i, p), daemon=True)
        t.start()
        threads.append(t)
    import time

    failed = None
    deadline = None if timeout is None else time.monotonic() + timeout
    try:
        while failed is None and any(p.poll() is None for p in procs):
            for i, p in enumerate(procs):
                rc = p.poll()
                if rc not in (None, 0):
                    failed = (i, rc)
                    break
            if deadline is not None and time.monotonic() > deadline:
                raise TimeoutError(f"workers still running after {timeout} s")
            if failed is None:
                time.sleep(0.05)
        if failed is None:
            for i, p in enumerate(procs):
                if p.returncode != 0:
                    failed = (i, p.returncode)
                    break
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
    for t in threads:
        t.join(timeout=5)
    if failed is not None:
        raise RuntimeError(f"worker rank {failed[0]} exited with status {failed[1]}")
    return outs


def last_json_line(text: str) -> dict:
    for line in reversed(text.strip().splitlines()):
        line = line.strip()
        if line.startswith("{"):
            return json.loads(line)
    raise ValueError("worker printed no JSON report")


# ----------------------------------------------------------------------------------------------------
# chunk transport: parent (single reader / inflater) -> workers
# ----------------------------------------------------------------------------------------------------
def write_chunk(fh, c: FastqChunk) -> None:
    """One frame: header, then offsets int64[n+1], header offsets int64[n+1], name_len int32[n] (optional),
    seq, qual, headers."""
    n, nb, nn = c.n, int(c.offsets[c.n]), int(c._name_offsets[c.n])
    fh.write(_HDR.pack(_MAGIC, 0 if c._name_len is None else 1, n, nb, nn))
    fh.write(np.ascontiguousarray(c.offsets[: n + 1], np.int64).tobytes())
    fh.write(np.ascontiguousarray(c._name_offsets[: n + 1], np.int64).tobytes())
    if c._name_len is not None:
        fh.write(np.ascontiguousarray(c._name_len[:n], np.int32).tobytes())
    fh.write(memoryview(np.ascontiguousarray(c.seq[:nb])))
    fh.write(memoryview(np.ascontiguousarray(c.qual[:nb])))
    fh.write(memoryview(np.ascontiguousarray(c._names[:nn])))


def _read_exact(fh, n: int) -> bytes:
    buf = bytearray(n)
    view, got = memoryview(buf), 0
    while got < n:
        k = fh.readinto(view[got:])
        if not k:
            raise EOFError("chunk stream ended inside a frame")
        got += k
    return bytes(buf)


def read_chunk(fh) -> FastqChunk | None:
    head = fh.read(_HDR.size)
    if not head:
        return None
    if len(head) < _HDR.size:
        head += _read_exact(fh, _HDR.size - len(head))
    magic, has_len, n, nb, nn = _HDR.unpack(head)
    if magic != _MAGIC:
        raise ValueError("bad chunk frame")
    offs = np.frombuffer(_read_exact(fh, 8 * (n + 1)), np.int64)
    hoffs = np.frombuffer(_read_exact(fh, 8 * (n + 1)), np.int64)
    nlen = np.frombuffer(_read_exact(fh, 4 * n), np.int32) if has_len else None
    seq = np.frombuffer(_read_exact(fh, nb), np.uint8)
    qual = np.frombuffer(_read_exact(fh, nb), np.uint8)
    names = np.frombuffer(_read_exact(fh, nn), np.uint8)
    return FastqChunk(n, seq, qual, offs, names, hoffs, nlen)


class PipeChunks:
    """Worker side: `for chunk in PipeChunks(fd)` yields the chunks the parent dealt to this rank."""

    def __init__(self, fd: int):
        self._fh = os.fdopen(fd, "rb", buffering=1 << 20)
        self.cpu_s = 0.0   # CPU seconds of the consuming thread spent reading frames (host-budget accounting)
        self.bytes = 0

    def __iter__(self) -> Iterator[FastqChunk]:
        import time

        while True:
            t0 = time.thread_time()
            c = read_chunk(self._fh)
            self.cpu_s += time.thread_time() - t0
            if c is None:
                return
            self.bytes += 2 * int(c.offsets[c.n]) + int(c._name_offsets[c.n])
            yield c

    def close(self):
        self._fh.close()


class ChunkFeeder:
    """The node's single reader / inflater: parse `path` once and deal its chunks round-robin to `world` pipes.  One writer
    thread per pipe with a short queue, so a worker that is momentarily busy does not stall the others' supply.

    Two transports.  Anonymous pipes (default): the command's own parent creates them and its workers inherit the read ends
    (`read_fds`).  Named FIFOs (`fifo_paths`): ranks started by an EXTERNAL launcher (torch.distributed.run) are siblings, not
    children -- rank 0 runs the feeder on FIFOs in a directory all ranks learn over the control plane, and every rank (rank 0
    included) opens its own with `open_fifo`.  `cpus`: CPUs for the feeder's threads (and the native inflate threads they
    start) -- under an external launcher rank 0 is a GPU worker with its own slice; the feeder takes the slice the plan keeps
    for a parent.  `cpu_s`: CPU seconds of the feeder's threads incl. the reader's native helpers, after join()."""

    def __init__(self, path: str, world: int, max_reads: int, max_bases: int, *, full_header: bool = False,
                 max_sample: int | None = None, depth: int = 3, fifo_paths: list[str] | None = None, cpus: list[int] | None = None):
        import queue

        self.path, self.world = path, world
        self.max_reads, self.max_bases, self.full_header, self.max_sample = max_reads, max_bases, full_header, max_sample
        self.fifo_paths = fifo_paths
        self.cpus = list(cpus) if cpus else None
        if fifo_paths is None:
            pipes = [os.pipe() for _ in range(world)]
            self.read_fds = [r for r, _ in pipes]
            self._write_fds = [w for _, w in pipes]
            for r in self.read_fds:
                os.set_inheritable(r, True)
        else:
            assert len(fifo_paths) == world
            for f in fifo_paths:
                os.mkfifo(f, 0o600)
            self.read_fds, self._write_fds = [], [None] * world  # (a FIFO's write end opens once its reader has: in the writer thread)
        self._queues = [queue.Queue(maxsize=depth) for _ in range(world)]
        self._threads: list[threading.Thread] = []
        self.error: BaseException | None = None
        self.n_reads = 0
        self.cpu_s = 0.0
        self._cpu_lock = threading.Lock()

    def _enter_thread(self) -> None:
        if self.cpus and hasattr(os, "sched_setaffinity"):
            try:
                os.sched_setaffinity(0, self.cpus)  # (pid 0 = the calling THREAD; threads it starts inherit the mask)
            except OSError:
                pass

    def _leave_thread(self, extra: float = 0.0) -> None:
        import time

        with self._cpu_lock:
            self.cpu_s += time.thread_time() + extra

    def start(self) -> None:
        """Call after the workers have been spawned (they inherited the read ends) / after the FIFOs' directory was announced."""
        for r in self.read_fds:
            os.close(r)
        for i in range(self.world):
            t = threading.Thread(target=self._writer, args=(i,), daemon=True)
            t.start()
            self._threads.append(t)
        t = threading.Thread(target=self._reader, daemon=True)
        t.start()
        self._threads.append(t)

    def _reader(self):
        helpers = 0.0
        try:
            self._enter_thread()
            seen = 0
            with FastqReader(self.path, full_header=self.full_header) as rd:
                for i, c in enumerate(rd.chunks(self.max_reads, self.max_bases)):
                    if self.max_sample is not None and seen + c.n > self.max_sample:
                        c = c.head(self.max_sample - seen)
                    seen += c.n
                    if c.n:
                        own = FastqChunk(c.n, c.seq.copy(), c.qual.copy(), c.offsets.copy(), c._names.copy(),
                                         c._name_offsets.copy(), None if c._name_len is None else c._name_len.copy())
                        self._queues[i % self.world].put(own)
                    if self.max_sample is not None and seen >= self.max_sample:
                        break
                helpers = rd.helper_cpu_seconds  # (complete when the stream was read to its end; a --max-sample cut-off under-counts)
            self.n_reads = seen
        except BaseException as e:
            self.error = e
        finally:
            self._leave_thread(helpers)
            for q in self._queues:
                q.put(None)

    def _writer(self, i: int):
        try:
            self._enter_thread()
            if self._write_fds[i] is None:
                self._write_fds[i] = self._open_fifo_for_writing(self.fifo_paths[i])
            with os.fdopen(self._write_fds[i], "wb", buffering=1 << 20) as fh:
                while True:
                    c = self._queues[i].get()
                    if c is None:
                        return
                    write_chunk(fh, c)
        except BaseException as e:
            # BrokenPipe: the worker died (wait_ranks / the rendezvous reports it).  Anything else (a rank that never opened its
            # FIFO, a full disk behind a pipe ...) is this feeder's error.  Either way keep taking the rank's share off the queue:
            # the reader thread deals round-robin through bounded queues and must not be left blocked on this one.
            if not isinstance(e, BrokenPipeError):
                self.error = self.error or e
            while self._queues[i].get() is not None:
                pass
        finally:
            self._leave_thread()

    @staticmethod
    def _open_fifo_for_writing(path: str, patience_s: float = 600.0) -> int:
        """A FIFO's write end opens once its reader has opened the other: wait for the rank, but not for ever (a rank that
        died before it got here would otherwise hang the feeder and, behind it, everyone's final rendezvous)."""
        import errno
        import time

        deadline = time.monotonic() + patience_s
        while True:
            try:
                fd = os.open(path, os.O_WRONLY | os.O_NONBLOCK)
                os.set_blocking(fd, True)
                return fd
            except OSError as e:
                if e.errno != errno.ENXIO or time.monotonic() > deadline:
                    raise
                time.sleep(0.01)

    def join(self) -> None:
        for t in self._threads:
            t.join()
        if self.fifo_paths:
            for f in self.fifo_paths:
                try:
                    os.unlink(f)
                except OSError:
                    pass
        if self.error is not None:
            raise self.error


def open_fifo(path: str) -> "PipeChunks":
    """Worker side of a FIFO-fed run: this rank's end of the feeder's FIFO (blocks until the feeder has opened its end)."""
    return PipeChunks(os.open(path, os.O_RDONLY))


def worker_python() -> list[str]:
    return [sys.executable]
