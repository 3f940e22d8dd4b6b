"""Ensemble sharding over the GPUs of one node: one process per GPU, RCCL over xGMI through the C ABI
(m4q_comm_* in include/m4q.h; librccl.so is loaded by libm4q_hip.so itself - no PyTorch anywhere in this package).
Ensemble members are independent closed loops (mpc4quantum/mpc.py:128-304 has no cross-instance data flow), so the
data path has NO collective: rank r runs the contiguous block [r*B/G, (r+1)*B/G) and ONE gather of the results closes
the job.

The results of a rank live in ONE contiguous byte buffer laid out by `ResultLayout`
    [ xs | us | exit_codes | steps_done | qp_solves | status ]      (each region sized for the largest shard)
On the GPU the session's output fields are bound INTO that buffer (m4q_session_bind_output): the kernel writes the
bytes RCCL sends, nothing is packed or staged through the host.  The collective sits behind a small `transport`
interface (rank, world, on_device, gather): `RcclComm` here is the product; the CPU tests plug a host transport
(tests/gloo_transport.py: torch.distributed "gloo") and a host solver into the same layout, padding and unpacking code.

Launch contract: the processes are started by any launcher that sets RANK, WORLD_SIZE, LOCAL_RANK and MASTER_PORT
(`python -m torch.distributed.run ...` does); the RCCL unique id travels from rank 0 to the others through a file
in the node's temporary directory keyed by that launch (one node: the ranks share a file system)."""
import ctypes as C
import os
import tempfile
import time

import numpy as np

from . import _lib


def shard_bounds(B, rank, world):
    """Contiguous block split; the first B % world ranks take one extra member."""
    base, extra = divmod(int(B), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _take(arr, lo, hi, B):
    """Slice the ensemble axis of an input that has one (leading axis of length B); pass shared inputs through."""
    arr = np.asarray(arr)
    return arr[lo:hi] if (arr.ndim > 0 and arr.shape[0] == B and B > 1) else arr


class ResultLayout:
    """Byte layout of one rank's results for `rows` members (time-major, as the C ABI holds them):
    xs [rows][xs_cols][n] complex128 | us [rows][ns][m] float64 | exit_codes [rows] i32 | steps_done [rows] i32 |
    qp_solves [rows][ns] i32 | status [4] i32.  xs_cols = ns + 1 (whole state history) or 1 (final state only).
    status[0] != 0: the rank's block is not valid (its launch left through the watchdog, or its solve raised): the rank still
    joins the gather - a rank that skipped it would leave every other rank waiting inside the collective."""

    FIELDS = ("xs", "us", "exit_codes", "steps_done", "qp_solves")

    def __init__(self, rows, n, m, ns, final_state_only=False):
        self.rows, self.n, self.m, self.ns = int(rows), int(n), int(m), int(ns)
        self.xs_cols = 1 if final_state_only else self.ns + 1
        self.shape = {"xs": (self.rows, self.xs_cols, self.n), "us": (self.rows, self.ns, self.m), "exit_codes": (self.rows,),
                      "steps_done": (self.rows,), "qp_solves": (self.rows, self.ns), "status": (4,)}
        self.dtype = {"xs": np.complex128, "us": np.float64, "exit_codes": np.int32, "steps_done": np.int32,
                      "qp_solves": np.int32, "status": np.int32}
        self.offset, pos = {}, 0
        for f in self.FIELDS + ("status",):
            self.offset[f] = pos
            pos += int(np.prod(self.shape[f])) * np.dtype(self.dtype[f]).itemsize
            pos = (pos + 15) // 16 * 16                       # every region starts 16-byte aligned
        self.nbytes = pos

    def field_bytes(self, f):
        return int(np.prod(self.shape[f])) * np.dtype(self.dtype[f]).itemsize

    def view(self, buf, f):
        """NumPy view of field f inside a host byte buffer (uint8 ndarray of self.nbytes)."""
        o = self.offset[f]
        return buf[o:o + self.field_bytes(f)].view(self.dtype[f]).reshape(self.shape[f])

    def pack(self, res, buf):
        """Host arrays (k <= rows members, time-major xs [k][ns+1][n] / us [k][ns][m]) -> buf."""
        k = res["us"].shape[0]
        xs = res["xs"][:, -1:, :] if self.xs_cols == 1 else res["xs"]
        self.view(buf, "xs")[:k] = xs
        for f in self.FIELDS[1:]:
            self.view(buf, f)[:k] = res[f]

    def unpack(self, buf, k):
        """First k members of a host byte buffer as a dict of arrays (copies)."""
        return {f: self.view(buf, f)[:k].copy() for f in self.FIELDS}


class DeviceBuffer:
    """Zero-filled device memory owned by libm4q_hip.so (m4q_device_alloc): what a gather sends from / receives into."""

    def __init__(self, nbytes, device=-1):
        self._L = _lib.lib()
        self.nbytes = int(nbytes)
        h = C.c_void_p()
        _lib.check(self._L.m4q_device_alloc(self.nbytes, int(device), C.byref(h)))
        self.ptr = h.value

    def read(self, offset=0, nbytes=None):
        n = self.nbytes - offset if nbytes is None else int(nbytes)
        out = np.empty(n, dtype=np.uint8)
        _lib.check(self._L.m4q_device_read(out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr + offset), n))
        return out

    def write(self, host, offset=0):
        a = np.ascontiguousarray(host).view(np.uint8).reshape(-1)
        _lib.check(self._L.m4q_device_write(C.c_void_p(self.ptr + offset), a.ctypes.data_as(C.c_void_p), a.nbytes))

    def free(self):
        if getattr(self, "ptr", None):
            self._L.m4q_device_free(C.c_void_p(self.ptr))
            self.ptr = None

    __del__ = free


def _launch_key():
    """Names THIS launch on this node: the launcher's port plus the identity (pid, start time) of the process that started
    the ranks (the launcher agent is the parent of every rank), so that a file left by an earlier launch is never read."""
    port = os.environ.get("MASTER_PORT", "0")
    ppid = os.getppid()
    try:
        start = open("/proc/%d/stat" % ppid).read().rsplit(")", 1)[1].split()[19]
    except (OSError, IndexError):
        start = "0"
    return "%s_%d_%s" % (port, ppid, start)


def exchange_unique_id(rank, world, path=None, timeout=300.0, make_id=None):
    """Rank 0 creates the RCCL unique id (m4q_comm_unique_id) and publishes it in a file (written aside and renamed: readers
    see all of it or nothing); the others wait for the file.  M4Q_UID_FILE overrides the path.
    make_id: a callable returning the UNIQUE_ID_BYTES to publish instead of RCCL's id (the device-free launch rehearsal,
    `bench.py --launch-check`, runs this same file protocol with a random token)."""
    path = path or os.environ.get("M4Q_UID_FILE") or os.path.join(tempfile.gettempdir(), "m4q_uid_" + _launch_key())
    n = _lib.UNIQUE_ID_BYTES
    if rank == 0:
        if make_id is None:
            buf = (C.c_char * n)()
            _lib.check(_lib.lib().m4q_comm_unique_id(C.cast(buf, C.c_void_p)))
            uid = bytes(buf)
        else:
            uid = bytes(make_id())
            assert len(uid) == n
        if world > 1:
            tmp = "%s.%d.tmp" % (path, os.getpid())
            with open(tmp, "wb") as f:
                f.write(uid)
            os.replace(tmp, path)
        return uid, path
    t0 = time.time()
    while True:
        try:
            data = open(path, "rb").read()
            if len(data) == n:
                return data, path
        except OSError:
            pass
        if time.time() - t0 > timeout:
            raise TimeoutError("rank %d: no RCCL unique id at %s after %.0f s (did rank 0 start?)" % (rank, path, timeout))
        time.sleep(0.01)


def free_port():
    """A TCP port nobody listens on right now (names the launch: MASTER_PORT is part of the unique-id file's key)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_local_ranks(argv, world, timeout=3000.0, env=None, out=None, err=None):
    """Start `world` child processes of `argv` on this node, one per GPU, and wait for them: the launcher `bench.py --gpus N` and
    any script built on `mpc_batch_sharded` use when no external launcher (`torch.distributed.run`) has set WORLD_SIZE.
    The caller must not have touched the GPU and this function does not: children are started with subprocess (never exec), each
    with RANK = LOCAL_RANK = r, WORLD_SIZE, MASTER_ADDR = 127.0.0.1, a free MASTER_PORT and one shared M4Q_UID_FILE (the file
    rank 0 publishes the RCCL unique id in).  Rank 0's stdout is relayed to `out` line by line as it comes, the other ranks'
    stdout is dropped, every rank's stderr goes to `err` (default: inherited).  Returns 0 when every rank exits 0.  When a rank fails, or the deadline
    passes, the others are terminated (then killed) and the return code is that rank's (124 for the deadline): no rank is left
    waiting inside a collective for a sibling that is gone."""
    import subprocess
    import sys
    import threading
    out = out or sys.stdout
    log = sys.stderr                        # the launcher's own messages; `err` (None: inherited) is the ranks' stderr
    port = free_port()
    uid = os.path.join(tempfile.gettempdir(), "m4q_uid_%d_%d_%d" % (port, os.getpid(), int(time.time() * 1e3)))
    base = dict(os.environ if env is None else env)
    base.update(WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), M4Q_UID_FILE=uid)
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: what RCCL needs between processes on these hosts
    procs = []
    for r in range(world):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(list(argv), env=e, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=err,
                                      start_new_session=True))

    def relay(stream):
        for line in iter(stream.readline, b""):
            out.write(line.decode(errors="replace"))
            out.flush()

    th = threading.Thread(target=relay, args=(procs[0].stdout,), daemon=True)
    th.start()
    deadline = time.time() + timeout
    rc = 0
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            rc = bad[0][1] if bad[0][1] > 0 else 128 - bad[0][1]
            log.write("launch: rank %d exited with code %d; stopping the other ranks\n" % bad[0])
            break
        if all(c == 0 for c in codes):
            break
        if time.time() > deadline:
            rc = 124
            log.write("launch: ranks %s still running after %.0f s; stopping them\n" % ([r for r, c in enumerate(codes) if c is None], timeout))
            break
        time.sleep(0.05)
    if rc:
        import signal
        for p in procs:                     # each child leads its own process group (its CPU-baseline workers go with it)
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGTERM)
                except OSError:
                    pass
        t_end = time.time() + 10.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except OSError:
                    pass
                p.wait()
    th.join(timeout=5.0)
    for path in (uid,):
        try:
            os.unlink(path)
        except OSError:
            pass
    return rc


class RcclComm:
    """One RCCL communicator per process (m4q_comm_create), device-resident gathers on its own stream."""

    on_device = True

    def __init__(self, rank, world, unique_id, device=-1):
        self._L = _lib.lib()
        self.rank, self.world = int(rank), int(world)
        self._h = C.c_void_p()
        idbuf = C.create_string_buffer(unique_id, _lib.UNIQUE_ID_BYTES)
        _lib.check(self._L.m4q_comm_create(self.rank, self.world, C.cast(idbuf, C.c_void_p), int(device), C.byref(self._h)))
        self._uid_path = None

    @classmethod
    def from_env(cls, device=None):
        """RANK / WORLD_SIZE / LOCAL_RANK as a launcher sets them (defaults: one rank).  Call before anything else touches the GPU
        only in the sense every multi-process GPU program must: the processes exist before their first HIP call."""
        rank = int(os.environ.get("RANK", "0"))
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
        uid, path = exchange_unique_id(rank, world)
        comm = cls(rank, world, uid, device)           # collective: returns once every rank has joined (all have read the file)
        comm.device = device
        if rank == 0 and world > 1:
            try:
                os.unlink(path)
            except OSError:
                pass
        return comm

    def gather(self, send_ptr, recv_ptr, nbytes, dst=0, slot=0, after=None):
        """ncclGather of nbytes device bytes per rank, enqueued behind everything queued on session `after`'s stream."""
        _lib.check(self._L.m4q_comm_gather(self._h, after._h if after is not None else None, C.c_void_p(send_ptr),
                                           C.c_void_p(recv_ptr) if recv_ptr else None, int(nbytes), int(dst), int(slot)))

    def wait(self, slot=-1):
        _lib.check(self._L.m4q_comm_wait(self._h, int(slot)))

    def allreduce(self, values, op="sum"):
        a = np.ascontiguousarray(values, dtype=np.float64).copy()
        _lib.check(self._L.m4q_comm_allreduce_f64(self._h, a.ctypes.data_as(_lib._dp), a.size, 0 if op == "sum" else 1))
        return a

    def barrier(self):
        _lib.check(self._L.m4q_comm_allreduce_f64(self._h, None, 0, 0))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.m4q_comm_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close


class ShardedResults:
    """A rank's result buffer and the one collective that moves it.
    transport.on_device (RcclComm): the session's outputs are bound into a library-owned device buffer; the gather is
    enqueued behind the kernel on the communicator's stream (no host synchronisation in between).
    Host transports (tests): the same layout filled from host arrays.
    final_state_only: the buffer carries xs[:, -1] instead of the whole state history (the history stays in the session)."""

    _FIELD_ID = {"xs": _lib.F_XS, "us": _lib.F_US, "exit_codes": _lib.F_CODES, "steps_done": _lib.F_STEPS_DONE,
                 "qp_solves": _lib.F_QP_SOLVES}

    def __init__(self, sess, rows, transport, dst=0, final_state_only=False, slot=0):
        self.sess, self.tr, self.dst, self.slot = sess, transport, dst, int(slot)
        p = sess.problem
        self.layout = ResultLayout(rows, p.dim_x, p.dim_u, p.n_steps, final_state_only)
        if sess.B > rows:
            raise ValueError("the buffer holds %d members, the session has %d" % (rows, sess.B))
        self.on_device = bool(transport.on_device)
        self.final_state_only = final_state_only
        self.world, self.rank = transport.world, transport.rank
        self.in_flight = False
        self.failed = None            # the exception that invalidated this rank's block, if any
        if self.on_device:
            if not hasattr(sess, "bind_output"):
                raise NotImplementedError("a device transport gathers a session's device buffer")
            dev = getattr(transport, "device", -1)
            self.buf = DeviceBuffer(self.layout.nbytes, dev)
            self.recv = DeviceBuffer(self.layout.nbytes * self.world, dev) if self.rank == dst else None
            self.bind()
        else:
            self.buf = np.zeros(self.layout.nbytes, dtype=np.uint8)
            self.recv = None

    def bind(self):
        """Point the session's output fields at THIS buffer (two ShardedResults can alternate on one session, so that the
        gather of one run overlaps the next run's kernel)."""
        if not self.on_device:
            return
        for f in ResultLayout.FIELDS:
            if f == "xs" and self.final_state_only:
                continue                      # the history stays in the session's own XS field; its last column is copied in gather()
            # the session's field covers sess.B members: a prefix of the region sized for `rows`
            self.sess.bind_output(self._FIELD_ID[f], self.buf.ptr + self.layout.offset[f], self.sess.field_bytes(self._FIELD_ID[f]))

    def wait(self):
        """Block until the gather last started from this buffer has completed (no-op otherwise)."""
        if self.in_flight:
            self.tr.wait(self.slot)
            self.in_flight = False

    def mark_failed(self, exc):
        """This rank's block is not valid; it still joins the gather, carrying the status word."""
        self.failed = exc

    def gather(self, wait=True):
        """The one collective of the job.  Call after sess.run(); on dst the per-rank bytes are then in self.recv / returned
        (host transports).  wait=False (device): return as soon as the collective is enqueued - call wait() before this buffer
        is bound and written again."""
        self.wait()
        lay = self.layout
        if self.on_device:
            if self.final_state_only:
                self.sess.copy_final_state(self.buf.ptr + lay.offset["xs"])
            self.sess.copy_status(self.buf.ptr + lay.offset["status"])
            self.tr.gather(self.buf.ptr, self.recv.ptr if self.recv is not None else None, lay.nbytes, self.dst, self.slot,
                           after=self.sess)
            self.in_flight = True
            if wait:
                self.wait()
                try:
                    self.sess.sync()                   # M4Q_E_TIMEOUT of THIS rank surfaces here, after the collective
                except _lib.M4qError as e:
                    self.failed = e
            return None
        res = None
        if self.failed is None:
            try:
                self.sess.sync()
                res = self.sess.results()
            except Exception as e:                     # noqa: BLE001 - whatever invalidated the block, the rank joins the gather
                self.failed = e
        if res is not None:                            # (a rank without members sends its zeroed buffer)
            lay.pack(res, self.buf)
        lay.view(self.buf, "status")[0] = 0 if self.failed is None else 1
        self.outs = self.tr.gather_host(self.buf, self.dst)
        return self.outs

    def unpack(self, counts):
        """dst only: host dict of the whole ensemble from the gathered buffers; counts[r] = members of rank r.
        Raises if any rank flagged its block invalid."""
        lay = self.layout
        if self.on_device:
            whole = self.recv.read()
            outs = [whole[r * lay.nbytes:(r + 1) * lay.nbytes] for r in range(self.world)]
        else:
            outs = self.outs
        bad = [r for r, o in enumerate(outs) if lay.view(o, "status")[0] != 0]
        if bad:
            raise _lib.M4qError(_lib.E_TIMEOUT, "rank(s) %s reported an invalid block (watchdog expiry or a failed solve): "
                                                "the gathered ensemble is not valid" % bad)
        parts = [lay.unpack(o, k) for o, k in zip(outs, counts) if k > 0]
        return {f: np.concatenate([q[f] for q in parts], axis=0) for f in ResultLayout.FIELDS}

    def close(self):
        if self.on_device:
            self.wait()
            self.buf.free()
            if self.recv is not None:
                self.recv.free()


class _HostBlock:
    """Stand-in for a session when the block was solved by an injected host callable (CPU tests): same attributes
    ShardedResults reads."""

    class _P:
        pass

    def __init__(self, res, n, m, ns):
        self.problem = self._P()
        self.problem.dim_x, self.problem.dim_u, self.problem.n_steps = n, m, ns
        self.B = 0 if res is None else res["us"].shape[0]
        self._res = res

    def sync(self):
        pass

    def results(self):
        return self._res


def mpc_batch_sharded(x0, models, dim_u, order, X_targ, U_targ, clock, plant_op0, plant_ops, Q, R, Qf, sat, du=None,
                      transport=None, dst=0, solver=None, final_state_only=False, **kw):
    """Same contract as mpc.mpc_batch, evaluated by every rank on its block; rank `dst` returns the full-ensemble dict
    (xs [B, n, cols], us [B, m, n_steps], exit_codes, steps_done, qp_solves), the others None.
    transport: an RcclComm (default: RcclComm.from_env()) - the block runs on the rank's GPU with its outputs bound into the
    gather buffer.  Tests pass a host transport and `solver`, a host callable with mpc_batch's signature, whose results take
    the same packed path."""
    from .mpc import open_session
    own = transport is None
    if own:
        transport = RcclComm.from_env()
    rank, world = transport.rank, transport.world
    x0 = np.asarray(x0)
    B, n = x0.shape
    ns = clock.n_steps
    lo, hi = shard_bounds(B, rank, world)
    counts = [shard_bounds(B, r, world)[1] - shard_bounds(B, r, world)[0] for r in range(world)]
    rows = max(counts)
    if transport.on_device and min(counts) == 0:
        # decided on data every rank has, so every rank raises - none is left waiting inside the collective
        raise ValueError("more ranks (%d) than ensemble members (%d)" % (world, B))
    if models is None:
        # models built on the device from generators (and per-member scales): every rank builds its own block's
        # (mpc.open_session: shared generators at order 1 then run the shared-generator kernel where it exists)
        if kw.get("generators") is None:
            raise TypeError("models is None: pass generators (and scales)")
        kw = dict(kw)
        g = np.asarray(kw["generators"])
        kw["generators"] = g[lo:hi] if g.ndim == 4 and g.shape[0] == B else g
        if kw.get("scales") is not None:
            kw["scales"] = np.asarray(kw["scales"])[lo:hi]
        models_blk = None
    else:
        models = np.asarray(models)
        if models.ndim == 2:
            models = models[None]
        models_blk = _take(models, lo, hi, B)
    op0 = np.asarray(plant_op0)
    ops = np.asarray(plant_ops)
    args = (x0[lo:hi], models_blk, dim_u, order,
            X_targ if np.ndim(X_targ) == 2 else _take(X_targ, lo, hi, B),
            U_targ if np.ndim(U_targ) == 2 else _take(U_targ, lo, hi, B), clock,
            _take(op0, lo, hi, B) if op0.ndim == 3 else op0, _take(ops, lo, hi, B) if ops.ndim == 4 else ops,
            Q, R, Qf, sat, du)
    sess = sr = None
    try:
        failed = None
        if not transport.on_device:
            if solver is None:
                from .mpc import mpc_batch as solver
            res = None
            if hi > lo:
                try:
                    res = dict(solver(*args, **kw))
                    res["xs"], res["us"] = np.swapaxes(res["xs"], 1, 2), np.swapaxes(res["us"], 1, 2)     # time-major
                except Exception as e:                 # noqa: BLE001 - the rank still joins the gather (status word)
                    failed, res = e, None
            block = _HostBlock(res, n, dim_u, ns)
        else:
            kw.setdefault("device", getattr(transport, "device", -1))
            sess = block = open_session(*args, **kw)
        sr = ShardedResults(block, rows, transport, dst, final_state_only)
        if failed is not None:
            sr.mark_failed(failed)
        if sess is not None:
            sess.run(0, ns)
        sr.gather()
        if sr.failed is not None and rank != dst:
            raise sr.failed
        if rank != dst:
            return None
        out = sr.unpack(counts)                        # raises if any rank (this one included) flagged its block
    finally:
        if sr is not None:
            sr.close()
        if sess is not None:
            sess.close()
        if own:
            transport.close()
    out["xs"] = np.swapaxes(out["xs"], 1, 2)
    out["us"] = np.swapaxes(out["us"], 1, 2)
    return out
