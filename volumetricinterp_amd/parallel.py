"""Multi-GPU plumbing: one process per GPU, records sharded in contiguous blocks.

The reference's record loop (volumetricinterp/interpolate.py:511) carries no state between records, so
the fit shards embarrassingly over timesteps and the evaluation over (timestep, point-tile) pairs
(SURVEY 8e).  There is NO collective on the data path; the only communication is one broadcast of the
shared parameters (beam geometry, regularisation matrices, hull facets) from rank 0 before the work
starts, a barrier / max-reduction around timed regions, and an optional gather of the coefficient rows to
rank 0 for a single HDF5 writer.

Two transports:

* ``backend='rccl'`` (GPU runs): a tiny control plane over a Unix-domain socket (rank 0 serves; ranks are
  launched by ``torch.distributed.run`` on ONE node and read RANK / WORLD_SIZE / LOCAL_RANK / MASTER_PORT from
  the environment) carries the RCCL unique id, barriers and scalar reductions; the parameter broadcast itself
  is ``ncclBroadcast`` on device buffers through libvinterp.so (``vi_rccl_*``), i.e. RCCL over xGMI.  PyTorch
  is deliberately NOT imported in GPU processes: its wheel bundles its own HIP runtime, and two HIP runtimes
  in one process crash (observed: SIGSEGV when torch was imported after libvinterp.so).
* ``backend='gloo'`` (CPU tests): ``torch.distributed`` with the gloo backend.
"""
import json
import os
import socket
import stat
import struct
import time

import numpy as np


def env_rank():
    """(rank, world_size, local_rank) from the launcher's environment (torch.distributed.run)."""
    return (int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1')),
            int(os.environ.get('LOCAL_RANK', '0')))


def shard_bounds(T, rank, world):
    """Contiguous block [lo, hi) of T records owned by `rank`: ceil(T / world) records per rank."""
    per = -(-T // world)
    lo = min(T, rank * per)
    return lo, min(T, lo + per)


def _send_msg(sock, payload):
    sock.sendall(struct.pack('<Q', len(payload)) + payload)


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(min(1 << 20, n - len(buf)))
        if not chunk:
            raise ConnectionError('control channel closed')
        buf += chunk
    return bytes(buf)


def _recv_msg(sock):
    (n,) = struct.unpack('<Q', _recv_exact(sock, 8))
    return _recv_exact(sock, n)


# Wire format of the control plane: length-prefixed frames of raw bytes; a list of frames is a count followed by the
# frames; arrays travel as a JSON header (shape) + their float64 bytes.  Nothing received is ever unpickled.
def _pack_parts(parts):
    return struct.pack('<I', len(parts)) + b''.join(struct.pack('<Q', len(p)) + p for p in parts)


def _unpack_parts(blob):
    (n,) = struct.unpack_from('<I', blob, 0)
    o, out = 4, []
    for _ in range(n):
        (k,) = struct.unpack_from('<Q', blob, o)
        out.append(bytes(blob[o + 8:o + 8 + k]))
        o += 8 + k
    return out


def _pack_array(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    head = json.dumps(list(a.shape)).encode('ascii')
    return struct.pack('<I', len(head)) + head + a.tobytes()


def _unpack_array(blob):
    (k,) = struct.unpack_from('<I', blob, 0)
    shape = tuple(int(x) for x in json.loads(blob[4:4 + k].decode('ascii')))
    return np.frombuffer(blob, dtype=np.float64, offset=4 + k).reshape(shape).copy()


def _private_socket_dir():
    """A directory only this user can enter, for the rendezvous socket: $XDG_RUNTIME_DIR when set, else
    /tmp/vinterp-<uid> created 0700.  A directory of that name that belongs to someone else, or that others may write
    to, is refused (another local user could otherwise pre-create the socket path and impersonate rank 0)."""
    base = os.environ.get('XDG_RUNTIME_DIR')
    if base and os.path.isdir(base) and os.stat(base).st_uid == os.getuid():
        return base
    d = os.path.join('/tmp', 'vinterp-%d' % os.getuid())
    try:
        os.mkdir(d, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(d)
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise PermissionError('%s is not a private directory of uid %d' % (d, os.getuid()))
    return d


def _peer_uid(conn):
    cred = conn.getsockopt(socket.SOL_SOCKET, socket.SO_PEERCRED, struct.calcsize('3i'))
    return struct.unpack('3i', cred)[1]


class SocketGroup(object):
    """Star-shaped control plane of the ranks of one node: rank 0 serves a Unix-domain socket."""

    def __init__(self, rank, world, timeout=120., data_timeout=None):
        """timeout: rendezvous (accept / connect).  data_timeout: the longest a rank waits inside allgather / bcast for a
        slower peer - e.g. for rank 0 to finish its own share of a fit before it serves the gather; default
        VINTERP_COMM_TIMEOUT seconds if set, else 10 x timeout; 0 or 'none' = wait for ever."""
        self.rank, self.world = rank, world
        if data_timeout is None:
            env = os.environ.get('VINTERP_COMM_TIMEOUT', '')
            data_timeout = (None if env.strip().lower() in ('0', 'none', 'inf') else float(env)) if env else 10 * timeout
        elif not data_timeout:
            data_timeout = None
        self.data_timeout = data_timeout
        run = os.environ.get('TORCHELASTIC_RUN_ID', 'none')
        port = os.environ.get('MASTER_PORT', '0')
        self.path = os.environ.get('VINTERP_RDV_PATH') or os.path.join(_private_socket_dir(),
                                                                       'rdv_%s_%s.sock' % (port, run))
        self.peers = {}
        self.timeout = timeout
        if rank == 0:
            try:
                os.unlink(self.path)
            except OSError:
                pass
            srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
            srv.bind(self.path)
            os.chmod(self.path, 0o600)
            srv.listen(world)
            srv.settimeout(timeout)
            while len(self.peers) < world - 1:
                conn, _ = srv.accept()
                if _peer_uid(conn) != os.getuid():          # not one of our ranks
                    conn.close()
                    continue
                conn.settimeout(timeout)
                (r,) = struct.unpack('<I', _recv_exact(conn, 4))
                if not (1 <= r < world) or r in self.peers:
                    conn.close()
                    for c in self.peers.values():
                        c.close()
                    srv.close()
                    raise RuntimeError('rendezvous at %s: a peer announced rank %d (%s) - two launches sharing one '
                                       'rendezvous path, or a wrong RANK / WORLD_SIZE in the environment'
                                       % (self.path, r, 'already taken' if r in self.peers else 'world size %d' % world))
                conn.settimeout(self.data_timeout)          # finite by default: a dead rank must not hang the others for ever
                self.peers[r] = conn
            srv.close()
            try:
                os.unlink(self.path)
            except OSError:
                pass
        else:
            t0 = time.time()
            while True:
                s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
                try:
                    s.connect(self.path)
                    break
                except OSError:
                    s.close()
                    if time.time() - t0 > timeout:
                        raise TimeoutError('rank %d could not reach rank 0 at %s' % (rank, self.path))
                    time.sleep(0.05)
            if _peer_uid(s) != os.getuid():
                s.close()
                raise PermissionError('the process serving %s does not belong to uid %d' % (self.path, os.getuid()))
            s.settimeout(self.data_timeout)
            s.sendall(struct.pack('<I', rank))
            self.sock = s

    def allgather(self, payload):
        """Every rank contributes bytes; every rank gets the list (rank order)."""
        if self.rank == 0:
            parts = [payload] + [None] * (self.world - 1)
            for r, c in self.peers.items():
                parts[r] = _recv_msg(c)
            blob = _pack_parts(parts)
            for c in self.peers.values():
                _send_msg(c, blob)
            return parts
        _send_msg(self.sock, payload)
        return _unpack_parts(_recv_msg(self.sock))

    def bcast(self, payload):
        """Rank 0's bytes to everyone."""
        if self.rank == 0:
            for c in self.peers.values():
                _send_msg(c, payload)
            return payload
        return _recv_msg(self.sock)

    def close(self):
        if self.rank == 0:
            for c in self.peers.values():
                c.close()
        else:
            self.sock.close()


class Comm(object):
    """Process group of the run (a no-op for a single process)."""

    def __init__(self, backend=None, ctx=None):
        self.rank, self.world, self.local_rank = env_rank()
        self.backend = None
        self.ctx = ctx                      # _lib.Context for RCCL broadcasts (backend 'rccl')
        self.grp = None
        self.dist = None
        self.rccl_ready = False
        self.notes = []
        if self.world <= 1:
            return
        if backend is None:
            backend = 'rccl' if ctx is not None else 'gloo'
        self.backend = backend
        if backend == 'gloo':
            import torch
            import torch.distributed as dist
            self.torch = torch
            dist.init_process_group(backend='gloo', rank=self.rank, world_size=self.world)
            self.dist = dist
        elif backend in ('rccl', 'socket'):
            self.grp = SocketGroup(self.rank, self.world)
            if backend == 'rccl' and ctx is not None:
                self._init_rccl()
        else:
            raise ValueError('unknown backend %r' % backend)

    def _init_rccl(self, timeout=90.):
        """Create the RCCL communicator.  The call is guarded by a timeout: a bootstrap that cannot complete (no
        usable network interface for the out-of-band exchange, for instance) must degrade to the control-socket
        broadcast, not hang the run."""
        import ctypes as C
        import threading
        from . import _lib
        ident = b''
        if self.rank == 0:
            buf = C.create_string_buffer(128)
            if _lib.lib.vi_rccl_unique_id(buf) == 0:
                ident = buf.raw
        ident = self.grp.bcast(ident)
        result = {}

        def work():
            result['rc'] = _lib.lib.vi_rccl_init(self.ctx.handle, self.world, self.rank, ident)
            err = _lib.lib.vi_last_error()           # thread-local in the library: read it on this thread
            result['err'] = err.decode('utf-8', 'replace') if err else ''

        ok = False
        if len(ident) == 128:
            th = threading.Thread(target=work, daemon=True)
            th.start()
            th.join(timeout)
            ok = (not th.is_alive()) and result.get('rc') == 0
            if th.is_alive():
                self.notes.append('ncclCommInitRank did not return within %.0f s' % timeout)
        oks = self.grp.allgather(b'1' if ok else b'0')
        self.rccl_ready = all(o == b'1' for o in oks)
        if not self.rccl_ready:
            self.notes.append('RCCL unavailable (%s); shared parameters go over the control socket'
                              % (result.get('err') or 'failed on another rank'))

    # ------------------------------------------------------------------------------------------
    def broadcast_arrays(self, arrays, src=0):
        """Broadcast a dict of float64 ndarrays from rank 0 (shapes known only there).

        One flat buffer, one collective: ~0.2 MB at the default order - latency-bound on xGMI."""
        if self.world <= 1:
            return arrays
        if src != 0:
            raise ValueError('only rank 0 can be the source')
        if self.backend == 'gloo':
            torch, dist = self.torch, self.dist
            meta = [(n, tuple(np.asarray(arrays[n]).shape)) for n in sorted(arrays)] if self.rank == 0 else None
            box = [meta]
            dist.broadcast_object_list(box, src=0)
            meta = box[0]
            total = int(sum(int(np.prod(s, dtype=np.int64)) for _, s in meta))
            flat = (np.concatenate([np.asarray(arrays[n], dtype=np.float64).ravel() for n, _ in meta])
                    if self.rank == 0 and total else np.empty(total))
            t = torch.from_numpy(flat)
            dist.broadcast(t, src=0)
            flat = t.numpy()
        else:
            meta = [(n, tuple(np.asarray(arrays[n]).shape)) for n in sorted(arrays)] if self.rank == 0 else None
            meta = [(n, tuple(sh)) for n, sh in json.loads(self.grp.bcast(json.dumps(meta).encode('ascii') if self.rank == 0
                                                                    else b'').decode('ascii'))]
            total = int(sum(int(np.prod(s, dtype=np.int64)) for _, s in meta))
            flat = (np.concatenate([np.asarray(arrays[n], dtype=np.float64).ravel() for n, _ in meta])
                    if self.rank == 0 and total else np.empty(total))
            done = False
            if self.rccl_ready and total:
                from . import _lib
                d = self.ctx.to_device(flat) if self.rank == 0 else self.ctx.empty((total,))
                rc = _lib.lib.vi_rccl_bcast_f64(self.ctx.handle, d.ptr, total, 0)
                oks = self.grp.allgather(b'1' if rc == 0 else b'0')
                if all(o == b'1' for o in oks):
                    flat = d.download()
                    done = True
                else:
                    self.notes.append('ncclBroadcast failed; fell back to the control socket')
            if not done:
                flat = np.frombuffer(self.grp.bcast(flat.tobytes() if self.rank == 0 else b''), dtype=np.float64)
        out, o = {}, 0
        for n, s in meta:
            k = int(np.prod(s, dtype=np.int64))
            out[n] = np.array(flat[o:o + k]).reshape(s)
            o += k
        return out

    def gather_rows(self, local, T):
        """Concatenate the per-rank row blocks (in rank order) on every rank; (T, ...) result."""
        local = np.ascontiguousarray(local, dtype=np.float64)
        if self.world <= 1:
            return local
        if self.backend == 'gloo':
            torch, dist = self.torch, self.dist
            per = -(-T // self.world)
            pad = np.zeros((per,) + local.shape[1:])
            pad[:local.shape[0]] = local
            t = torch.from_numpy(pad)
            outs = [torch.empty_like(t) for _ in range(self.world)]
            dist.all_gather(outs, t)
            return np.concatenate([o.numpy() for o in outs], axis=0)[:T]
        parts = self.grp.allgather(_pack_array(local))
        return np.concatenate([_unpack_array(p) for p in parts], axis=0)[:T]

    def barrier(self):
        if self.world <= 1:
            return
        if self.backend == 'gloo':
            self.dist.barrier()
        else:
            self.grp.allgather(b'')

    def max_over_ranks(self, x):
        if self.world <= 1:
            return float(x)
        if self.backend == 'gloo':
            t = self.torch.tensor([float(x)], dtype=self.torch.float64)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            return float(t.item())
        return max(struct.unpack('<d', p)[0] for p in self.grp.allgather(struct.pack('<d', float(x))))

    def allgather_bytes(self, payload):
        """Every rank contributes a byte string; every rank gets the list in rank order (control plane, small data)."""
        if self.world <= 1:
            return [payload]
        if self.backend == 'gloo':
            out = [None] * self.world
            self.dist.all_gather_object(out, bytes(payload))
            return out
        return self.grp.allgather(bytes(payload))

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()
            self.dist = None
        if self.grp is not None:
            if self.rccl_ready and self.ctx is not None:
                from . import _lib
                _lib.lib.vi_rccl_destroy(self.ctx.handle)
            self.grp.close()
            self.grp = None
