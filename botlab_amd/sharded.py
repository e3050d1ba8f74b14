"""Particle sharding across the GPUs of one node: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI; "gloo" in the CPU tests).

Exchange per moved update:
  1. every rank runs resample-gather + action + sensor model for its block of output particles [lo, hi) and writes its
     slice of the 16-byte exchange record (x, y, theta, weight-units);
  2. ONE all-gather of the record (in place: each rank's slice already sits at its offset) -- the only collective;
  3. every rank scans the integer weight units of all N particles (exact, so every rank derives the same cumulative
     and the same total) and forms the pose estimate from the gathered record -- the reference's own serially rounded
     float sums over all N particles in order -- so the estimate is bit-identical on every rank, for every shard count,
     and equal to one GPU's.  That end of the update is either `updateFilter` (own launches) or, after `updateBegin`,
     rides in the map kernel as on one GPU.
The map update, distance grid and A* are replicated (every rank applies the identical integer update; no traffic).

The engine behind a shard is pluggable so the orchestration is testable without a GPU: the product engine is
HipShardEngine (libbotlab_hip.so); tests/ supply a CPU stand-in to exercise the collectives under gloo.
"""
import ctypes as C
import os

import numpy as np
import torch
import torch.distributed as dist

from . import host
from ._capi import check


def shard_bounds(num_particles, rank, world):
    """Block partition with equal padded block size S = ceil(N / world); shard = [rank*S, min(N, (rank+1)*S))."""
    S = (num_particles + world - 1) // world
    lo = rank * S
    hi = min(num_particles, lo + S)
    if lo >= hi:
        raise ValueError(f"rank {rank} of {world} would own no particle of {num_particles}")
    return lo, hi, S


class HipShardEngine:
    """One shard on one MI355X.  The exchange buffers are torch tensors (device memory + stream plumbing) handed to
    the library, so collectives run on them in place."""

    def __init__(self, num_particles, rank, world, device):
        self.N, self.rank, self.world = num_particles, rank, world
        self.lo, self.hi, self.S = shard_bounds(num_particles, rank, world)
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        # ONE explicit stream carries both the library's kernels and the collectives (issued under
        # torch.cuda.stream(self.stream)), so they are stream-ordered without events.  torch's default stream has the
        # raw handle 0, which the C ABI reads as "create your own stream" -- never hand that one over.
        self.stream = torch.cuda.Stream(self.device)
        assert self.stream.cuda_stream != 0
        self.ctx = host.Context(device, stream=self.stream.cuda_stream)
        padded = self.S * world
        with torch.cuda.stream(self.stream):          # zero-fills on OUR stream: the null stream (and its hardware queue) stays unused
            self.rec = [torch.zeros(padded, 4, dtype=torch.float32, device=self.device) for _ in range(2)]
        self.stream.synchronize()
        self.pf = host.ParticleFilter(num_particles, ctx=self.ctx, shard=(self.lo, self.hi))
        check(self.ctx.lib.bl_pf_set_exchange_buffers(self.pf.h, self.rec[0].data_ptr(), self.rec[1].data_ptr()))

    def init_at_pose(self, pose, seed):
        self.pf.initializeFilterAtPose(pose, seed=seed)

    def set_particles(self, particles, units=None):
        self.pf.setParticles(particles, units)

    def begin(self, odometry, scan, grid, rand_value, noise=None):
        return self.pf.updateBegin(odometry, scan, grid, rand_value, noise)

    def exchange_record(self):
        ptr = self.ctx.lib.bl_pf_exchange_rec_ptr(self.pf.h)
        for t in self.rec:
            if t.data_ptr() == ptr:
                return t
        raise RuntimeError("exchange record pointer does not match a bound buffer")

    def end(self, want_pose=True):
        return self.pf.updateEnd(want_pose)

    def particles(self):
        return self.pf.particles()


class ShardedParticleFilter:
    """ParticleFilter whose particles are block-partitioned over the ranks of a process group."""

    def __init__(self, engine, group=None):
        self.engine = engine
        self.group = group
        self.world = engine.world
        self.rank = engine.rank
        # exercise the collectives even with one rank (plumbing check on a 1-GPU box)
        self.force_collectives = bool(os.environ.get("BOTLAB_FORCE_COLLECTIVES")) and dist.is_initialized()
        self._views = None
        self._stream_current = False
        self.comm = None
        if (self.world > 1 or self.force_collectives) and dist.is_initialized() and hasattr(engine, "ctx") \
                and dist.get_backend(self.group) == "nccl" and not os.environ.get("BOTLAB_TORCH_COLLECTIVES"):
            self.comm = self._direct_comm()

    def _direct_comm(self):
        """The library's own RCCL communicator (csrc/bl_comm.hip): the all-gather is then ONE call that enqueues the collective
        on the filter's stream.  torch.distributed is the rendezvous only (rank 0's unique id is broadcast).  Every rank ends
        up with the same answer: if the communicator did not come up anywhere, all ranks keep torch's all_gather."""
        lib = self.engine.ctx.lib
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so").encode()

        def everyone(ok):
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.engine.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
            return int(flag.item()) == 1

        # 1. every rank can load the library (no RCCL call yet: a rank that could not would leave the others waiting inside
        #    the collective initialisation below)
        if not everyone(os.path.exists(path.decode()) and lib.bl_comm_load(path) == 0):
            return None
        # 2. rank 0's unique id reaches every rank
        ident = [None]
        if self.rank == 0:
            buf = C.create_string_buffer(128)
            if lib.bl_comm_unique_id(path, buf) == 0:
                ident[0] = bytes(buf.raw)
        dist.broadcast_object_list(ident, src=0, group=self.group)
        if ident[0] is None:
            return None
        # 3. the communicator (collective), then agreement that it came up everywhere
        h = C.c_void_p()
        ok = lib.bl_comm_create(self.engine.ctx.h, path, ident[0], self.rank, self.world, C.byref(h)) == 0
        if not everyone(ok):
            if ok:
                lib.bl_comm_destroy(h)
            return None
        return h

    def close(self):
        if self.comm is not None:
            self.engine.ctx.lib.bl_comm_destroy(self.comm)
            self.comm = None

    def initializeFilterAtPose(self, pose, seed=1):
        self.engine.init_at_pose(pose, seed)          # counter-based: every rank generates the identical full record

    def setParticles(self, particles, units=None):
        self.engine.set_particles(particles, units)

    def _exchange_views(self):
        """(whole record, this rank's slice) per exchange buffer, made once: slicing a tensor costs microseconds per call."""
        if self._views is None:
            S = self.engine.S
            self._views = {}
            for t in getattr(self.engine, "rec", []):
                self._views[t.data_ptr()] = (t, t[self.rank * S:(self.rank + 1) * S])
        return self._views

    def updateFilter(self, odometry, scan, grid, rand_value, noise=None, want_pose=True):
        self.updateBegin(odometry, scan, grid, rand_value, noise)
        return self.engine.end(want_pose)

    def updateBegin(self, odometry, scan, grid, rand_value, noise=None):
        """The update up to and including the exchange.  Its end -- weight prefix and pose estimate, identical on every rank -- is
        either `engine.end()` or rides in the map kernel (Mapping.updateMapFinishingFilter / AsyncPlanner.
        submit_with_map_update_finishing with `engine.pf`), as on a single GPU."""
        moved = self.engine.begin(odometry, scan, grid, rand_value, noise)
        if moved and self.comm is not None:
            lib = self.engine.ctx.lib
            check(lib.bl_comm_all_gather_inplace(self.comm, lib.bl_pf_exchange_rec_ptr(self.engine.pf.h), self.engine.S * 4))
        elif moved and (self.world > 1 or self.force_collectives):
            stream = getattr(self.engine, "stream", None)
            if stream is not None:
                # the collective is ordered on the engine's stream: that stream is made torch's current stream of this thread
                # once (entering a stream context per step costs ~10 us of host time)
                if not self._stream_current:
                    torch.cuda.set_stream(stream)
                    self._stream_current = True
                ptr = self.engine.ctx.lib.bl_pf_exchange_rec_ptr(self.engine.pf.h)
                rec, mine = self._exchange_views()[ptr]
                dist.all_gather_into_tensor(rec, mine, group=self.group)
            else:
                rec = self.engine.exchange_record()
                S = self.engine.S
                dist.all_gather_into_tensor(rec, rec[self.rank * S:(self.rank + 1) * S], group=self.group)
        return moved

    def particles(self):
        return self.engine.particles()
