"""Multi-GPU plumbing: one process per GPU, records sharded in contiguous blocks.

The reference's record loop (volumetricinterp/interpolate.py:511) carries no state between records, so
the fit shards embarrassingly over timesteps and the evaluation over (timestep, point-tile) pairs
(SURVEY 8e).  There is NO collective on the data path; the only communication is one broadcast of the
shared parameters (beam geometry, regularisation matrices, hull facets) from rank 0 before the work
starts, and an optional gather of the coefficient rows to rank 0 for a single HDF5 writer.

``torch.distributed`` is used purely as the transport (backend "nccl" is RCCL over xGMI on the GPU box,
"gloo" in the CPU tests); nothing here touches the compute path, which stays in libvinterp.so.
"""
import os

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


class Comm(object):
    """Thin wrapper over a torch.distributed process group (or a no-op for a single process)."""

    def __init__(self, backend=None, device=None):
        self.rank, self.world, self.local_rank = env_rank()
        self.dist = None
        self.device = device
        if self.world > 1:
            import torch
            import torch.distributed as dist
            self.torch = torch
            if backend is None:
                backend = 'nccl' if torch.cuda.is_available() else 'gloo'
            self.backend = backend
            if backend == 'nccl':
                torch.cuda.set_device(self.local_rank)
                self.device = torch.device('cuda', self.local_rank)
                dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world,
                                        device_id=self.device)
            else:
                self.device = torch.device('cpu')
                dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world)
            self.dist = dist

    def broadcast_arrays(self, arrays, src=0):
        """Broadcast a dict of float64 ndarrays (shapes known on every rank only through rank `src`).

        One flat buffer, one collective: ~0.3 MB at the default order - latency-bound on xGMI."""
        if self.dist is None:
            return arrays
        torch, dist = self.torch, self.dist
        if self.rank == src:
            names = sorted(arrays)
            meta = [(n, tuple(np.asarray(arrays[n]).shape)) for n in names]
        else:
            meta = None
        box = [meta]
        dist.broadcast_object_list(box, src=src)
        meta = box[0]
        total = int(sum(int(np.prod(s, dtype=np.int64)) for _, s in meta))
        if self.rank == src:
            flat = np.concatenate([np.asarray(arrays[n], dtype=np.float64).ravel() for n, _ in meta]) \
                if total else np.zeros(0)
        else:
            flat = np.empty(total)
        t = torch.from_numpy(flat).to(self.device)
        dist.broadcast(t, src=src)
        flat = t.cpu().numpy()
        out, o = {}, 0
        for n, s in meta:
            k = int(np.prod(s, dtype=np.int64))
            out[n] = flat[o:o + k].reshape(s).copy()
            o += k
        return out

    def gather_rows(self, local, T):
        """Concatenate the per-rank row blocks (in rank order) on every rank; (T, ...) result."""
        local = np.ascontiguousarray(local, dtype=np.float64)
        if self.dist is None:
            return local
        torch, dist = self.torch, self.dist
        per = -(-T // self.world)
        pad = np.zeros((per,) + local.shape[1:])
        pad[:local.shape[0]] = local
        t = torch.from_numpy(pad).to(self.device)
        outs = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(outs, t)
        full = np.concatenate([o.cpu().numpy() for o in outs], axis=0)
        return full[:T]

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
            if self.backend == 'nccl':
                self.torch.cuda.synchronize()

    def max_over_ranks(self, x):
        if self.dist is None:
            return float(x)
        t = self.torch.tensor([float(x)], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()
            self.dist = None
