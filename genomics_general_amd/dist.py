"""Multi-GPU sharding of windows: one process per GPU, windows split into contiguous ranges, one RCCL all-gather
of the per-window result table (C1 in SURVEY.md section 2.2; replaces the sorter/writer re-ordering of
popgenWindows.py:108-157).  Windows are independent, so there is no other data-path communication.

Launch contract: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment (what
`python -m torch.distributed.run` sets).  The product path does not import torch: the 128-byte RCCL unique id is
handed from rank 0 to the other ranks of the node through an atomically renamed file in /tmp, everything else goes
through RCCL in libpopgen_hip.so.  `GlooComm` (torch.distributed, backend gloo) exists for the CPU tests of the
sharding / gather logic.
"""
import os
import time

import numpy as np


class World:
    def __init__(self, rank=0, size=1, local_rank=0):
        self.rank, self.size, self.local_rank = rank, size, local_rank


def world_from_env(env=None):
    env = os.environ if env is None else env
    size = int(env.get("WORLD_SIZE", "1"))
    rank = int(env.get("RANK", "0"))
    return World(rank, size, int(env.get("LOCAL_RANK", str(rank))))


def device_for(world):
    """GPU index of this rank: LOCAL_RANK, folded into the visible devices when a launcher has already restricted them
    (e.g. one device per rank through HIP_VISIBLE_DEVICES)."""
    from . import _lib
    n = _lib.device_count()
    return world.local_rank % n if n > 0 else world.local_rank


def shard_range(n_items, size, rank):
    """Contiguous balanced split: ranks [0, n%size) get one extra item."""
    base, extra = divmod(n_items, size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_counts(n_items, size):
    return [shard_range(n_items, size, r)[1] - shard_range(n_items, size, r)[0] for r in range(size)]


# ---- rendezvous of the RCCL unique id ------------------------------------------------------------------
def _rdzv_path():
    """File through which rank 0 hands the RCCL unique id to the other ranks of the node.  Under torch.distributed.run all ranks
    are children of one agent process, whose pid makes the name unique per launch; other launchers (ranks started from separate
    shells, mpirun wrappers) are keyed by MASTER_ADDR / MASTER_PORT alone, or name the file themselves with PG_RDZV_FILE."""
    if os.environ.get("PG_RDZV_FILE"):
        return os.environ["PG_RDZV_FILE"]
    key = "%s_%s" % (os.environ.get("MASTER_ADDR", "local"), os.environ.get("MASTER_PORT", "0"))
    if "TORCHELASTIC_RUN_ID" in os.environ:
        key += "_%s_%d" % (os.environ["TORCHELASTIC_RUN_ID"], os.getppid())
    return os.path.join("/tmp", "pg_rdzv_" + key.replace("/", "_"))


def exchange_unique_id(world, make_id, timeout_s=180.0):
    """rank 0 creates the id and publishes it; the others wait for the file.  Single node only."""
    path = _rdzv_path()
    if world.rank == 0:
        try:
            os.remove(path)                  # a leftover of a launch that died before its hand-over completed
        except OSError:
            pass
        try:
            uid = make_id()
        except Exception as exc:
            # the other ranks are waiting for this file: tell them, so that every rank falls back to files together instead of
            # timing out one by one
            with open(path + ".tmp%d" % os.getpid(), "wb") as f:
                f.write(("RCCL-FAILED " + str(exc)[:200]).encode())
            os.replace(path + ".tmp%d" % os.getpid(), path)
            raise
        tmp = path + ".tmp%d" % os.getpid()
        with open(tmp, "wb") as f:
            f.write(uid)
        os.replace(tmp, path)
        return uid, path
    t0 = time.time()
    while True:
        try:
            with open(path, "rb") as f:
                uid = f.read()
            if len(uid) == 128:
                return uid, path
            if uid.startswith(b"RCCL-FAILED") and os.path.getmtime(path) >= _T_START - 120.0:
                # rank 0 could not create the id: nobody waits for it (make_comm).  (An older marker is the leftover of a launch
                # that failed under the same rendezvous name: this launch's rank 0 is about to replace it.)
                raise RuntimeError("rank 0 reported: " + uid.decode("utf-8", "replace")[:200])
        except FileNotFoundError:
            pass
        if time.time() - t0 > timeout_s:
            raise TimeoutError("rank %d: no RCCL unique id at %s after %.0f s" % (world.rank, path, timeout_s))
        check_peers(world)
        time.sleep(0.01)


# ---- a rank that fails tells the others (the reference's worst habit is to hang on a worker's error, popgenWindows.py:456-460) ----
_T_START = time.time()
# The launch this process belongs to, once it is known: rank 0 makes it up (the exchange directory of the file communicator, a
# digest of the RCCL unique id), the others learn it in the rendezvous.  A failure marker quotes it, and a marker that quotes
# another launch's token is never believed (under a stable rendezvous name -- MASTER_ADDR / MASTER_PORT,
# PG_RDZV_FILE -- the marker of a launch that failed seconds ago used to end a healthy one).
_LAUNCH = {"token": None}
_UNTOKENED_SLACK_S = 2.0


class PeerFailed(RuntimeError):
    """another rank of this launch has failed: this one stops at once instead of waiting for it at the next exchange"""


def _failure_glob():
    return _rdzv_path() + ".failed_r"


def set_launch_token(token):
    _LAUNCH["token"] = None if token is None else str(token)


def forget_own_marker(world):
    """every rank, when it starts: the marker an earlier launch's rank of the same number left under this rendezvous name"""
    try:
        path = _failure_glob() + str(world.rank)
        if os.path.getmtime(path) < _T_START:
            os.remove(path)
    except OSError:
        pass


def mark_failed(world, exc):
    """leave `<rendezvous>.failed_r<rank>` with the launch's token and the reason (drivers: cli.guarded_main).  A rank that leaves
    because a peer failed leaves a marker too (the reason quotes the peer's), so that ranks which cannot see the first marker --
    it was written before they started -- do not sit out PG_COMM_TIMEOUT; the first marker of a rank stays."""
    if world.size <= 1:
        return
    try:
        path = _failure_glob() + str(world.rank)
        if isinstance(exc, PeerFailed) and os.path.exists(path) and os.path.getmtime(path) >= _T_START:
            return
        with open(path + ".tmp", "w") as f:
            f.write("pg-failure token=%s start=%.3f\n" % (_LAUNCH["token"] or "-", _T_START))
            f.write("%s: %s" % (type(exc).__name__, str(exc)[:300]))
        os.replace(path + ".tmp", path)
    except OSError:
        pass


def clear_failures(world):
    """rank 0, before it opens the launch's rendezvous: markers a dead launch left under the same name"""
    import glob
    for path in glob.glob(_failure_glob() + "*"):
        try:
            if os.path.getmtime(path) < _T_START - 1.0:
                os.remove(path)
        except OSError:
            pass


def _read_marker(path):
    """(token or None, reason) of a marker file"""
    with open(path) as f:
        text = f.read(600)
    token = None
    if text.startswith("pg-failure "):
        head, _, text = text.partition("\n")
        for field in head.split()[1:]:
            if field.startswith("token=") and field != "token=-":
                token = field[6:]
    return token, text[:300]


def peer_failure(world):
    """"rank R failed: reason" of the first marker another rank of THIS launch has left, else None.  A marker that quotes a launch
    token counts only when it is this process's token (a rank that does not know its launch yet does not believe it: it is about
    to learn the token, or rank 0 -- which has removed every older marker and therefore believes what it finds -- is about to leave
    one of its own).  A marker without a token (a rank that failed before the rendezvous) counts when it is not older than this
    process, give or take the seconds ranks of one launch start apart."""
    import glob
    mine = _LAUNCH["token"]
    for path in sorted(glob.glob(_failure_glob() + "[0-9]*")):
        try:
            r = int(path[len(_failure_glob()):])
            if r == world.rank:
                continue
            mtime = os.path.getmtime(path)
            token, reason = _read_marker(path)
            if token is not None:
                if token != mine and not (mine is None and world.rank == 0 and mtime >= _T_START - 1.0):
                    continue
            elif mtime < _T_START - (1.0 if world.rank == 0 else _UNTOKENED_SLACK_S):
                continue
            return "rank %d failed: %s" % (r, reason)
        except (OSError, ValueError):
            continue
    return None


def check_peers(world):
    msg = peer_failure(world)
    if msg:
        raise PeerFailed(msg)


# ---- communicators (same small interface) ------------------------------------------------------------------
class SoloComm:
    size, rank = 1, 0

    def allgather(self, arr):
        return np.asarray(arr, dtype=np.float64).reshape(1, -1)

    def barrier(self):
        pass

    def close(self):
        pass


class RcclComm:
    """RCCL communicator owned by an Engine's ctx (pg_comm_* of the C-ABI)."""

    def __init__(self, engine, world):
        from .engine import Engine
        self.e, self.size, self.rank, self.world = engine, world.size, world.rank, world
        self.timeout_s = float(os.environ.get("PG_COMM_TIMEOUT", "300"))
        forget_own_marker(world)
        if world.rank == 0:
            clear_failures(world)
        # RCCL prints a version banner on file descriptor 1 while it initialises; the drivers' stdout carries data (CSV, the
        # benchmark's JSON line), so fd 1 is pointed at stderr for the duration of the initialisation
        import sys
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            uid, path = exchange_unique_id(world, Engine.comm_unique_id)
            import hashlib
            set_launch_token("rccl-" + hashlib.sha1(bytes(uid)).hexdigest()[:20])
            self._guarded(engine.comm_setup, world.size, world.rank, uid)
            self._guarded(engine.comm_barrier)
        finally:
            os.dup2(saved, 1)
            os.close(saved)
        if world.rank == 0:
            try:
                os.remove(path)
            except OSError:
                pass

    def _guarded(self, fn, *args):
        """A collective whose peer has died never returns (ncclAllGather blocks in hipStreamSynchronize for good).  The call itself
        stays on this thread (the communicator and the device context are never touched from another one); a watchdog thread looks
        for a failed peer and for PG_COMM_TIMEOUT while the call is out, says why and ends the process -- the communicator cannot
        be used again, and a thread stuck in the runtime cannot be interrupted."""
        import sys
        import threading
        done = threading.Event()
        t0 = time.time()

        def watch():
            while not done.wait(0.05):
                why = peer_failure(self.world)
                if why is None and time.time() - t0 > self.timeout_s:
                    why = "no answer from the other ranks within PG_COMM_TIMEOUT = %.0f s" % self.timeout_s
                if why is not None and not done.is_set():
                    sys.stderr.write("rank %d: leaving the collective: %s\n" % (self.rank, why))
                    sys.stderr.flush()
                    mark_failed(self.world, PeerFailed(why))
                    os._exit(3)
        th = threading.Thread(target=watch, daemon=True, name="collective-watchdog")
        th.start()
        try:
            return fn(*args)
        finally:
            done.set()

    def allgather(self, arr):
        return self._guarded(self.e.comm_allgather, arr)

    def barrier(self):
        self._guarded(self.e.comm_barrier)

    def close(self):
        pass


class FileComm:
    """Host-side communicator through files next to the rendezvous file (`PG_COMM=file`).  RCCL refuses two ranks on one device,
    so this is what lets a multi-rank launch (bench.py --gpus N, the drivers) be exercised on a single-GPU box; the data path
    has no collective, only finished rows and barriers travel.  Every all-gather is one file per rank, renamed into place.

    The exchange directory is unique per launch: rank 0 makes a fresh one and publishes its name through `<rendezvous>.dir`; every
    other rank leaves a hello file with a random token in the directory it finds named there and adopts the directory only when
    rank 0's `go` file quotes that token -- so the leftovers of a launch that died under the same MASTER_ADDR / MASTER_PORT (a stale
    pointer, a stale directory with finished exchanges in it) are never read."""

    def __init__(self, world, timeout_s=None):
        import json
        import tempfile
        import uuid
        if timeout_s is None:
            timeout_s = float(os.environ.get("PG_COMM_TIMEOUT", "300"))        # how long a rank waits for the others at an exchange
        self.size, self.rank, self.timeout_s, self.world = world.size, world.rank, timeout_s, world
        self._last_check = 0.0
        forget_own_marker(world)
        if world.rank == 0:
            clear_failures(world)
        base = _rdzv_path()
        self.pointer = base + ".dir"
        self.seq = 0
        t0 = time.time()
        if self.rank == 0:
            self.dir = tempfile.mkdtemp(prefix=os.path.basename(base) + ".d.", dir=os.path.dirname(base) or ".")
            set_launch_token("dir-" + os.path.basename(self.dir))
            self._put(self.pointer, self.dir.encode())
            tokens = {}
            while len(tokens) < self.size - 1:
                for r in range(1, self.size):
                    if r not in tokens:
                        try:
                            with open(os.path.join(self.dir, "hello_r%d" % r)) as f:
                                tokens[r] = f.read()
                        except OSError:
                            pass
                if time.time() - t0 > timeout_s:
                    raise TimeoutError("rank 0: only %d of %d ranks arrived at %s" % (1 + len(tokens), self.size, self.dir))
                self._watch()
                time.sleep(0.0005)
            self._put(os.path.join(self.dir, "go"), json.dumps({str(r): t for r, t in tokens.items()}).encode())
            # every rank is here, so every rank has read the "RCCL-FAILED" note this rank may have left in the hand-over file of the
            # unique id (exchange_unique_id): it must not outlive the launch (the next launch under the same
            # MASTER_ADDR / MASTER_PORT would read it before its own rank 0 has replaced it)
            try:
                with open(base, "rb") as f:
                    stale = f.read(11) == b"RCCL-FAILED"
                if stale:
                    os.remove(base)
            except OSError:
                pass
            return
        token, said = uuid.uuid4().hex, set()
        while True:
            try:
                with open(self.pointer) as f:
                    d = f.read()
                if d and os.path.isdir(d):
                    if d not in said:
                        self._put(os.path.join(d, "hello_r%d" % self.rank), token.encode())
                        said.add(d)
                    with open(os.path.join(d, "go")) as f:
                        if json.load(f).get(str(self.rank)) == token:
                            self.dir = d
                            set_launch_token("dir-" + os.path.basename(d))
                            return
            except (OSError, ValueError):
                pass
            if time.time() - t0 > timeout_s:
                raise TimeoutError("rank %d: rank 0 never opened an exchange directory through %s" % (self.rank, self.pointer))
            self._watch()
            time.sleep(0.0005)

    def _watch(self):
        """inside every wait loop: a rank that has failed (cli.guarded_main leaves a marker) ends the wait at once; looked for 20
        times a second"""
        now = time.time()
        if now - self._last_check >= 0.05:
            self._last_check = now
            check_peers(self.world)

    @staticmethod
    def _put(path, data):
        tmp = path + ".tmp%d" % os.getpid()
        with open(tmp, "wb") as f:
            f.write(data)
        os.replace(tmp, path)

    def _name(self, seq, rank):
        return os.path.join(self.dir, "g%d_r%d.npy" % (seq, rank))

    def allgather(self, arr):
        a = np.ascontiguousarray(arr, dtype=np.float64).ravel()
        mine = self._name(self.seq, self.rank)
        with open(mine + ".tmp", "wb") as f:
            np.save(f, a)
        os.replace(mine + ".tmp", mine)
        rows, t0 = [], time.time()
        for r in range(self.size):
            path = self._name(self.seq, r)
            while not os.path.exists(path):
                if time.time() - t0 > self.timeout_s:
                    raise TimeoutError("rank %d: rank %d never arrived at exchange %d (%s)" % (self.rank, r, self.seq, path))
                self._watch()
                time.sleep(0.0005)
            rows.append(a if r == self.rank else np.load(path))
        # everybody has passed exchange seq-1 once its files of exchange seq exist: the own file of seq-1 can go
        if self.seq >= 1:
            try:
                os.remove(self._name(self.seq - 1, self.rank))
            except OSError:
                pass
        self.seq += 1
        return np.stack(rows)

    def barrier(self):
        self.allgather(np.zeros(1))

    def close(self):
        """The files of the last exchange cannot be removed by their writers (a slower rank may still have to read them): every
        rank leaves a marker, rank 0 waits for all markers and removes the directory and the pointer to it."""
        self.barrier()
        with open(os.path.join(self.dir, "done_r%d" % self.rank), "w"):
            pass
        try:
            os.remove(_failure_glob() + str(self.rank))            # (a marker of this rank from an earlier, failed launch)
        except OSError:
            pass
        if self.rank != 0:
            return
        t0 = time.time()
        while not all(os.path.exists(os.path.join(self.dir, "done_r%d" % r)) for r in range(self.size)):
            if time.time() - t0 > self.timeout_s:
                return
            time.sleep(0.001)
        for name in os.listdir(self.dir):
            try:
                os.remove(os.path.join(self.dir, name))
            except OSError:
                pass
        try:
            os.rmdir(self.dir)
            with open(self.pointer) as f:
                mine = f.read() == self.dir
            if mine:
                os.remove(self.pointer)
        except OSError:
            pass


def make_comm(engine, world):
    """The communicator of a multi-rank launch: RCCL over xGMI, or files when PG_COMM=file (ranks sharing one device)."""
    if world.size <= 1:
        return SoloComm()
    if os.environ.get("PG_COMM") == "file":
        return FileComm(world)
    try:
        return RcclComm(engine, world)
    except TimeoutError:
        raise                                 # a rank never showed up: files would wait for it just as long
    except Exception as exc:                  # RCCL itself refused (no peer access, IPC mode, ...): such failures hit every rank
        # alike; when rank 0 cannot even create the id it says so through the hand-over file and the others end up here too
        import sys
        sys.stderr.write("rank %d: RCCL communicator unavailable (%s); the finished rows travel through files instead (PG_COMM=file)\n"
                         % (world.rank, str(exc)[:200]))
        return FileComm(world)


class GlooComm:
    """CPU stand-in with the same interface (tests only): torch.distributed, backend gloo."""

    def __init__(self, world):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        if not dist.is_initialized():
            dist.init_process_group("gloo", rank=world.rank, world_size=world.size)
        self.size, self.rank = world.size, world.rank

    def allgather(self, arr):
        t = self.torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float64).ravel().copy())
        outs = [self.torch.zeros_like(t) for _ in range(self.size)]
        self.dist.all_gather(outs, t)
        return np.stack([o.numpy() for o in outs])

    def barrier(self):
        self.dist.barrier()

    def close(self):
        if self.dist.is_initialized():
            self.dist.destroy_process_group()


def gather_table(comm, local_rows, n_total):
    """All ranks hold rows of their contiguous window shard ([n_local][k] float64); returns the full
    [n_total][k] table in window order on every rank.  One all-gather of equal-size (padded) blocks."""
    local_rows = np.asarray(local_rows, dtype=np.float64)
    k = local_rows.shape[1] if local_rows.ndim == 2 else 1
    if k == 0:                       # no statistic column at all (popgenWindows.py --analysis popPairDist with one population)
        return np.zeros((n_total, 0))
    local_rows = local_rows.reshape(-1, k)
    counts = shard_counts(n_total, comm.size)
    assert local_rows.shape[0] == counts[comm.rank], "shard size mismatch"
    width = max(counts) if counts else 0
    pad = np.zeros((width, k), dtype=np.float64)
    pad[:local_rows.shape[0]] = local_rows
    allr = comm.allgather(pad.ravel()).reshape(comm.size, width, k)
    return np.concatenate([allr[r, :counts[r]] for r in range(comm.size)], axis=0) if n_total else np.zeros((0, k))


def sum_counts(comm, counts):
    """Element-wise sum over the ranks of an integer array (pair counts, site counts: additive over the sites a rank holds), the
    same on every rank.  One all-gather; the integers travel as float64 (exact below 2^53) and are added in int64."""
    a = np.ascontiguousarray(counts)
    allr = comm.allgather(a.ravel().astype(np.float64))
    return np.rint(allr).astype(np.int64).sum(axis=0).reshape(a.shape)


def gather_bytes(comm, data):
    """One byte string per rank -> the list of all of them (in rank order) on every rank: two all-gathers, the sizes and the
    payload padded to the largest (the bytes travel as the bit patterns of float64 words; an all-gather only copies)."""
    data = bytes(data)
    sizes = comm.allgather(np.array([float(len(data))])).ravel().astype(np.int64)
    width = int((int(sizes.max()) + 7) // 8) if len(sizes) else 0
    if width == 0:
        return [b"" for _ in range(comm.size)]
    buf = np.zeros(width * 8, dtype=np.uint8)
    buf[:len(data)] = np.frombuffer(data, dtype=np.uint8)
    allr = np.ascontiguousarray(comm.allgather(buf.view(np.float64))).reshape(comm.size, width)
    return [allr[r].view(np.uint8)[:int(sizes[r])].tobytes() for r in range(comm.size)]
