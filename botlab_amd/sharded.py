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


COMPOSED_ALIGN = 2048       # a rank's block is whole finish groups (2 x 1024 particles) and whole scan tiles (512) ...


def composed_align(num_particles):
    """... for the particle counts whose finish runs large groups; below that (bl_mcl_finish.h: MCLF_GT_SWITCH) a group is
    2 x 256 particles = one scan tile, and blocks of 512 let eight ranks share 100 000 particles."""
    return COMPOSED_ALIGN if num_particles >= 160_000 else 512



def shard_bounds(num_particles, rank, world, align=1):
    """Block partition with equal padded block size S = ceil(N / world), rounded up to a multiple of `align`;
    shard = [rank*S, min(N, (rank+1)*S))."""
    S = (num_particles + world - 1) // world
    S = (S + align - 1) // align * align
    lo = rank * S
    hi = min(num_particles, lo + S)
    if lo >= hi:
        raise ValueError(f"rank {rank} of {world} would own no particle of {num_particles}")
    return lo, hi, S


class _DeviceArray:
    """A library-owned device buffer as something torch.as_tensor understands (no copy, no ownership)."""

    def __init__(self, ptr, nfloats):
        self.__cuda_array_interface__ = {"shape": (int(nfloats),), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


def composed_possible(num_particles, world):
    """The composed finish needs every rank to own particles once the blocks are whole finish groups."""
    if world < 2 or world > 8 or os.environ.get("BOTLAB_SHARD_REPLICATED"):
        return False
    S = (num_particles + world - 1) // world
    a = composed_align(num_particles)
    S = (S + a - 1) // a * a
    return (world - 1) * S < num_particles


class HipShardEngine:
    """One shard on one MI355X.  Replicated form: the exchange buffers are torch tensors (device memory + stream plumbing)
    handed to the library, so the all-gather of the whole record runs on them in place.  Composed form (`composed=True`, see
    ShardedParticleFilter): the library owns the records; every rank maps the other ranks' arrays (IPC) and the exchange is two
    small all-gathers of library buffers."""

    def __init__(self, num_particles, rank, world, device, composed=False):
        self.N, self.rank, self.world = num_particles, rank, world
        self.composed = bool(composed)
        self.lo, self.hi, self.S = shard_bounds(num_particles, rank, world, composed_align(num_particles) if self.composed else 1)
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        # ONE explicit stream carries both the library's kernels and the collectives (issued under
        # torch.cuda.stream(self.stream)), so they are stream-ordered without events.  torch's default stream has the
        # raw handle 0, which the C ABI reads as "create your own stream" -- never hand that one over.
        self.stream = torch.cuda.Stream(self.device)
        assert self.stream.cuda_stream != 0
        self.ctx = host.Context(device, stream=self.stream.cuda_stream)
        self.pf = host.ParticleFilter(num_particles, ctx=self.ctx, shard=(self.lo, self.hi))
        self.rec = []
        self._opened = []
        if not self.composed:
            padded = self.S * world
            with torch.cuda.stream(self.stream):          # zero-fills on OUR stream: the null stream (and its hardware queue) stays unused
                self.rec = [torch.zeros(padded, 4, dtype=torch.float32, device=self.device) for _ in range(2)]
            self.stream.synchronize()
            check(self.ctx.lib.bl_pf_set_exchange_buffers(self.pf.h, self.rec[0].data_ptr(), self.rec[1].data_ptr()))

    # ---- composed form: mapping the ranks' arrays into each other
    def shard_setup(self):
        check(self.ctx.lib.bl_pf_shard_setup(self.pf.h, self.rank, self.world, self.S))

    def local_handles(self):
        """IPC handles of this rank's two records and its weight prefix (3 x 64 bytes)."""
        lib = self.ctx.lib
        ptrs = [C.c_void_p() for _ in range(3)]
        check(lib.bl_pf_shard_local_ptrs(self.pf.h, *[C.byref(p) for p in ptrs]))
        out = []
        for p in ptrs:
            buf = C.create_string_buffer(64)
            check(lib.bl_ipc_export(p, buf))
            out.append(bytes(buf.raw))
        return out, [p.value for p in ptrs]

    def set_peers(self, handles_by_rank, own_ptrs):
        lib = self.ctx.lib
        for r, hs in enumerate(handles_by_rank):
            if r == self.rank:
                ptrs = own_ptrs
            else:
                ptrs = []
                for h in hs:
                    q = C.c_void_p()
                    check(lib.bl_ipc_open(h, C.byref(q)))
                    self._opened.append(q)
                    ptrs.append(q.value)
            check(lib.bl_pf_shard_set_peer(self.pf.h, r, ptrs[0], ptrs[1], ptrs[2]))
        check(lib.bl_pf_shard_commit(self.pf.h))

    # ---- peer-store form of the exchange: the tile-sum buffer, the exchange blocks and the counter table of every rank
    def local_handles_peer(self):
        lib = self.ctx.lib
        ptrs = [C.c_void_p() for _ in range(3)]
        check(lib.bl_pf_shard_local_ptrs_peer(self.pf.h, *[C.byref(p) for p in ptrs]))
        out = []
        for p in ptrs:
            buf = C.create_string_buffer(64)
            check(lib.bl_ipc_export(p, buf))
            out.append(bytes(buf.raw))
        return out, [p.value for p in ptrs]

    def set_peers_peer(self, handles_by_rank, own_ptrs):
        lib = self.ctx.lib
        for r, hs in enumerate(handles_by_rank):
            if r == self.rank:
                ptrs = own_ptrs
            else:
                ptrs = []
                for h in hs:
                    q = C.c_void_p()
                    check(lib.bl_ipc_open(h, C.byref(q)))
                    self._opened.append(q)
                    ptrs.append(q.value)
            check(lib.bl_pf_shard_set_peer_buffers(self.pf.h, r, ptrs[0], ptrs[1], ptrs[2]))
        check(lib.bl_pf_shard_peer_commit(self.pf.h))

    def peer_selftest(self):
        ok = C.c_int(0)
        check(self.ctx.lib.bl_pf_shard_peer_selftest(self.pf.h, C.byref(ok)))
        return bool(ok.value)

    def shard_buffers(self):
        """(tile-sums buffer, exchange buffer) of the composed finish as flat float32 tensors over the library's memory."""
        lib = self.ctx.lib
        a, b, na, nb = C.c_void_p(), C.c_void_p(), C.c_size_t(), C.c_size_t()
        check(lib.bl_pf_shard_buffers(self.pf.h, C.byref(a), C.byref(na), C.byref(b), C.byref(nb)))
        self.sums_floats, self.xchg_floats = na.value // 4, nb.value // 4
        sums = torch.as_tensor(_DeviceArray(a.value, self.sums_floats * self.world), device=self.device)
        xchg = torch.as_tensor(_DeviceArray(b.value, self.xchg_floats * self.world), device=self.device)
        return (sums, a.value), (xchg, b.value)

    def traffic(self):
        """bytes per rank and update of the composed exchange: (sent into the all-gathers, received, own block of records)"""
        v = (C.c_int64 * 3)()
        check(self.ctx.lib.bl_pf_shard_traffic(self.pf.h, v))
        return tuple(int(x) for x in v)

    def close_peers(self):
        for q in self._opened:
            self.ctx.lib.bl_ipc_close(q)
        self._opened = []

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


def ipc_probe(ctx, rank, world, group=None):
    """Can every rank map every other rank's device memory (hipIpc)?  Collective; the same answer on every rank.  bench.py asks
    before it makes the engines, because the composed form fixes other shard bounds than the replicated one."""
    lib = ctx.lib
    ok, handle, p = True, None, C.c_void_p()
    def word_of(r):                                  # what rank r leaves in its probe buffer
        return (0x9E3779B1 * (r + 1)) & 0xFFFFFFFF

    try:
        check(lib.bl_dev_alloc(ctx.h, 4096, C.byref(p)))
        check(lib.bl_dev_word(ctx.h, p, 1, C.byref(C.c_uint32(word_of(rank)))))
        buf = C.create_string_buffer(64)
        check(lib.bl_ipc_export(p, buf))
        handle = bytes(buf.raw)
    except Exception:                                # noqa: BLE001
        ok = False
    gathered = [None] * world
    dist.all_gather_object(gathered, (handle, int(ctx.device)), group=group)      # (behind every rank's write: bl_dev_word synchronises)
    devices = [g[1] for g in gathered]
    gathered = [g[0] for g in gathered]
    opened = []
    if ok and all(h is not None for h in gathered):
        for r, h in enumerate(gathered):
            if r == rank:
                continue
            # a kernel that reads a mapping of a device this one has no peer access to faults instead of failing: ask first
            if devices[r] != int(ctx.device) and not torch.cuda.can_device_access_peer(int(ctx.device), devices[r]):
                ok = False
                break
            q = C.c_void_p()
            if lib.bl_ipc_open(h, C.byref(q)) != 0:
                ok = False
                break
            opened.append(q)
            got = C.c_uint32(0)                      # ... and a kernel of THIS device reads what the owner wrote
            if lib.bl_dev_word(ctx.h, q, 0, C.byref(got)) != 0 or got.value != word_of(r):
                ok = False
                break
    else:
        ok = False
    dev = "cpu" if dist.get_backend(group) != "nccl" else torch.device("cuda", ctx.device)
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    for q in opened:
        lib.bl_ipc_close(q)
    dist.barrier(group=group)                         # every mapping is gone before anybody frees
    if p.value:
        lib.bl_dev_free(p)
    return int(flag.item()) == 1


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
        self.composed = False
        self.peer = False                    # composed finish whose exchange is peer stores (no collective)
        self.peer_why = ""
        self._sums = self._xchg = None
        if (self.world > 1 or self.force_collectives) and dist.is_initialized() and hasattr(engine, "ctx") \
                and dist.get_backend(self.group) == "nccl" and not os.environ.get("BOTLAB_TORCH_COLLECTIVES"):
            self.comm = self._direct_comm()

    def _everyone(self, ok):
        """Do all ranks say yes?  (a rank that cannot go on must not leave the others inside a collective)"""
        dev = getattr(self.engine, "device", None)
        if dist.get_backend(self.group) != "nccl":
            dev = "cpu"
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        return int(flag.item()) == 1

    def _setup_composed(self):
        """Once the filter holds particles: every rank maps the other ranks' records and weight prefixes (IPC handles travel over
        torch.distributed, the rendezvous) and the engine switches to the composed finish.  All ranks agree on every step; a rank
        that cannot map another's memory makes the whole job stop with the reason -- the shard bounds of the composed form are
        fixed when the engines are made (probe with `ipc_probe` first, as bench.py does, to choose the replicated form instead)."""
        eng = self.engine
        ok, why = True, ""
        try:
            eng.shard_setup()
            handles, own = eng.local_handles()
        except Exception as e:                       # noqa: BLE001 -- reported below, on every rank
            ok, why, handles, own = False, repr(e), None, None
        if not self._everyone(ok):
            raise RuntimeError(f"composed finish: set-up failed on some rank ({why or 'another rank'})")
        gathered = [None] * self.world
        dist.all_gather_object(gathered, handles, group=self.group)
        try:
            eng.set_peers(gathered, own)
        except Exception as e:                       # noqa: BLE001
            ok, why = False, repr(e)
        if not self._everyone(ok):
            raise RuntimeError(f"composed finish: a rank could not map another rank's memory ({why or 'another rank'})")
        self._sums, self._xchg = eng.shard_buffers()
        self.composed = True
        self.peer = False
        if not os.environ.get("BOTLAB_SHARD_NO_PEER_STORES") and hasattr(eng, "local_handles_peer"):
            self._setup_peer_stores()

    def _setup_peer_stores(self):
        """The exchange without collectives: every rank maps the others' tile-sum buffers, exchange blocks and counter tables, then
        a self-test pushes a pattern to every rank and checks every rank's pattern (device-side spin limit: a dead link gives 'no',
        not a hang).  ALL ranks keep the form or all leave it for the collective one -- agreed over torch.distributed."""
        eng = self.engine
        ok, why = True, ""
        try:
            handles, own = eng.local_handles_peer()
        except Exception as e:                       # noqa: BLE001
            ok, why, handles, own = False, repr(e), None, None
        if not self._everyone(ok):
            self.peer_why = f"buffers could not be exported ({why or 'another rank'})"
            return
        gathered = [None] * self.world
        dist.all_gather_object(gathered, handles, group=self.group)
        try:
            eng.set_peers_peer(gathered, own)
        except Exception as e:                       # noqa: BLE001
            ok, why = False, repr(e)
        if not self._everyone(ok):
            # (a rank that did commit leaves the form again: nobody may push into buffers some rank has not mapped)
            if ok:
                check(eng.ctx.lib.bl_pf_shard_peer_reset(eng.pf.h, 0))
            self.peer_why = f"a rank could not map another rank's buffers ({why or 'another rank'})"
            return
        dist.barrier(group=self.group)               # every rank has committed before anybody pushes
        try:
            ok = eng.peer_selftest()
        except Exception as e:                       # noqa: BLE001
            ok, why = False, repr(e)
        keep = self._everyone(ok)
        check(eng.ctx.lib.bl_pf_shard_peer_reset(eng.pf.h, 1 if keep else 0))
        self.peer = keep
        self.peer_why = "" if keep else f"the self-test failed on some rank ({why or ('this rank' if not ok else 'another rank')})"

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
        if self.composed:
            self.engine.ctx.sync()
            if dist.is_initialized():
                dist.barrier(group=self.group)        # nobody unmaps memory another rank's kernels may still read
            self.engine.close_peers()
            self.composed = False
        if self.comm is not None:
            self.engine.ctx.lib.bl_comm_destroy(self.comm)
            self.comm = None

    def initializeFilterAtPose(self, pose, seed=1):
        self.engine.init_at_pose(pose, seed)          # counter-based: every rank generates the identical full record
        if getattr(self.engine, "composed", False) and not self.composed:
            self._setup_composed()

    def setParticles(self, particles, units=None):
        self.engine.set_particles(particles, units)
        if getattr(self.engine, "composed", False) and not self.composed:
            self._setup_composed()

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
        if moved and self.composed:
            self._exchange_composed()
        elif moved and self.comm is not None:
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

    def _exchange_composed(self):
        """tile sums -> all-gather #1 -> groups -> all-gather #2, everything on the filter's stream."""
        lib, pf = self.engine.ctx.lib, self.engine.pf.h
        if getattr(self, "peer", False):
            check(lib.bl_pf_shard_exchange_peer(pf))     # no collective: pushes into the other ranks' buffers + counter waits
            return
        if self.comm is not None:
            check(lib.bl_pf_shard_exchange(pf, self.comm))
            return
        if not self._stream_current:
            torch.cuda.set_stream(self.engine.stream)
            self._stream_current = True
        for stage, (buf, _), per_rank in ((1, self._sums, self.engine.sums_floats), (2, self._xchg, self.engine.xchg_floats)):
            check(lib.bl_pf_shard_stage(pf, stage))
            dist.all_gather_into_tensor(buf, buf[self.rank * per_rank:(self.rank + 1) * per_rank], group=self.group)

    def exchange_bytes_per_update(self):
        """(bytes this rank sends into collectives, bytes it receives from them, bytes of source records its k_mcl_main reads --
        from wherever they lie) per moved update"""
        if self.composed:
            return self.engine.traffic()
        per = self.engine.S * 16
        return (per, (self.world - 1) * per, (self.engine.hi - self.engine.lo) * 16) if self.world > 1 or self.force_collectives else (0, 0, (self.engine.hi - self.engine.lo) * 16)

    def particles(self):
        return self.engine.particles()
