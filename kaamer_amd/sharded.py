"""Sharded index over several devices: one process per GPU.

The table is partitioned by hash prefix (`kaamer_shard_of`); rank r holds shard r.  Per
batch, every rank searches the WHOLE query batch against its shard (the kernels skip the
keys they do not own), which yields partial hit lists `(protein id, partial Kmatch, first
position)` per query.  Query q is owned by rank `q % world`: one all-to-all (RCCL over
xGMI: every rank talks to every other rank at once, all seven links busy — not a ring)
moves each partial list to its owner, and the owner merges them on the device
(`kaamer_exchange_merge`: integer sums, so the result is bit-identical to the one-GPU path).

`ShardedSearcher` is plumbing only: C-ABI calls (kaamer_search_device, kaamer_exchange_pack,
kaamer_exchange_merge, kaamer_topn_device) around ONE equal-split collective.  Transports:

  "rccl"   kaamer_rccl_alltoall on an ncclComm_t this module creates with ncclCommInitRank
           (grouped ncclSend / ncclRecv; what a Go host would call) -- no host synchronisation
  "torch"  torch.distributed.all_to_all_single on the device blocks (backend nccl = RCCL)
  "host"   blocks staged through pinned host memory and exchanged with a CPU backend (gloo):
           for ranks that share one device (RCCL cannot put two ranks on one GPU) and for
           hosts without a peer path.  The blocks are the same bytes.

(A single-process, multi-device index needs none of this: kaamer_index_open_sharded drives
all shards from one process, see include/kaamer_hip.h.)
"""
import ctypes as C

import torch
import torch.distributed as dist

from . import abi, api


class _DevView:
    """zero-copy torch view of a raw device pointer owned by the C library"""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (int(ptr), False), "version": 3}


def dev_tensor(ptr, n, dtype):
    typestr = {torch.int32: "<i4", torch.int64: "<i8", torch.uint8: "|u1"}[dtype]
    if n == 0:
        return torch.zeros(0, dtype=dtype, device="cuda")
    return torch.as_tensor(_DevView(ptr, n, typestr), device="cuda")


class _NcclUniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]  # NCCL_UNIQUE_ID_BYTES


def _rccl():
    """librccl as the process already has it (torch's copy has the soname librccl.so.1), else the system's"""
    for name in ("librccl.so.1", "librccl.so"):
        try:
            L = C.CDLL(name, mode=C.RTLD_GLOBAL)
            break
        except OSError:
            L = None
    if L is None:
        raise ImportError("librccl is not loadable")
    L.ncclGetUniqueId.argtypes = [C.POINTER(_NcclUniqueId)]
    L.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _NcclUniqueId, C.c_int]
    L.ncclCommDestroy.argtypes = [C.c_void_p]
    L.ncclCommCount.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    L.ncclGetErrorString.restype = C.c_char_p
    L.ncclGetErrorString.argtypes = [C.c_int]
    return L


class RcclComm:
    """An ncclComm_t of `world` ranks made with ncclGetUniqueId / ncclCommInitRank, the calls a Go host makes through
    cgo.  The unique id travels through `group` (any torch.distributed group of the same ranks;
    world 1 needs none)."""

    def __init__(self, rank, world, group=None):
        self.L = _rccl()
        uid = _NcclUniqueId()
        if rank == 0:
            self._chk(self.L.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
        if world > 1:
            box = [bytes(uid.internal) if rank == 0 else None]
            dist.broadcast_object_list(box, src=0, group=group)
            C.memmove(C.byref(uid), box[0], 128)
        self.handle = C.c_void_p()
        self._chk(self.L.ncclCommInitRank(C.byref(self.handle), world, uid, rank), "ncclCommInitRank")
        n = C.c_int()
        self._chk(self.L.ncclCommCount(self.handle, C.byref(n)), "ncclCommCount")
        self.world = n.value  # the rank count RCCL saw

    def _chk(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s: %s" % (what, self.L.ncclGetErrorString(rc).decode()))

    def close(self):
        if self.handle:
            self.L.ncclCommDestroy(self.handle)
            self.handle = C.c_void_p()


PHASES = ("search", "pack", "alltoall", "merge", "topn")


class ShardedSearcher:
    """Rank-local driver of the sharded index: search my shard, pack, all-to-all, merge my queries, post-steps.
    Every compute step is a C-ABI call enqueued on one stream.  Works for protein and for nucleotide / reads input
    (every rank translates; the owner's post-steps use its own ORFs: orf_source)."""

    def __init__(self, index, rank, world, max_seq_bytes, max_seqs, seq_type=abi.PROTEIN, max_entries_per_peer=1 << 20,
                 group=None, max_hits=0, g_tier_slots=0, first_pos=None, transport="torch", comm=None, adaptive=True,
                 margin=0.25, direct_at_world1=False, concurrent_batches=0):
        """first_pos: carry the lowest matching position of every hit through the exchange.  Default: as the reference
        fills PositionHits (search.go:416) -- nucleotide / reads input yes (SetBestStartCodon reads it), protein input no
        (a third less to pack, send, unpack and merge).
        transport: "rccl" (comm: an RcclComm; made here when None), "torch", or "host" (see the module docstring).
        adaptive: size the blocks of a batch from what the batch before last needed (+ margin) instead of the buffers'
        capacity, so that the all-to-all moves payload; the figures come out of the received block headers and are the
        same on every rank (kaamer_exchange_stats).  A batch that does not fit raises KaamerError(E_CAPACITY) from
        finish() on every rank; `run()` repeats it at full capacity.
        direct_at_world1: with ONE shard the partial lists ARE the results -- skip pack, exchange and merge (the second
        counting pass) and run the post-steps on the search workspace.
        concurrent_batches: how many searchers of this process have batches in flight next to each other (ShardedPipeline);
        handed to the workspaces (kaamer_workspace_opts.concurrent_batches)."""
        assert transport in ("rccl", "torch", "host")
        self.index, self.rank, self.world, self.group = index, rank, world, group
        self.transport = transport
        self.nucl = seq_type in (abi.READS, abi.NUCLEOTIDE)
        if first_pos is None:
            first_pos = self.nucl
        fp = 1 if first_pos else 2
        self.ws_first_pos = bool(first_pos)
        self.ws = api.Workspace(index, max_seq_bytes, max_seqs, seq_type=seq_type, first_pos=fp, max_hits=max_hits,
                                g_tier_slots=g_tier_slots, concurrent_batches=concurrent_batches)
        self.layout = abi.ExchangeLayout()
        abi.check(abi.lib().kaamer_exchange_layout_init(world, rank, self.ws.query_capacity, max_entries_per_peer,
                                                        C.byref(self.layout)))
        L = self.layout            # the capacity layout: what the buffers hold
        self.wire = L              # the layout of the batch being enqueued
        self.adaptive, self.margin = bool(adaptive), float(margin)
        self._full_next = False    # a batch overflowed its blocks: full capacity until the statistics have caught up
        self.n_steps = 0
        self.direct = bool(direct_at_world1) and world == 1
        self.mws = api.Workspace(index, 64, L.q_cap, max_queries=L.q_cap, first_pos=fp, max_hits=world * L.e_cap,
                                 g_tier_slots=g_tier_slots)
        n = world * int(L.block_words)
        self.block_bytes = 4 * int(L.block_words)
        self.send = torch.empty(n, dtype=torch.int32, device="cuda")
        # (through RCCL the blocks really travel, also from a rank to itself: a receive buffer of its own)
        self.recv = torch.empty(n, dtype=torch.int32, device="cuda") if (world > 1 or transport == "rccl") else self.send
        self.comm, self._own_comm = comm, False
        if transport == "rccl" and comm is None:
            self.comm, self._own_comm = RcclComm(rank, world, group), True
        if transport == "host" and world > 1:
            self.h_send = torch.empty(n, dtype=torch.int32).pin_memory()
            self.h_recv = torch.empty(n, dtype=torch.int32).pin_memory()
        self.phase_ms = None

    def _choose_layout(self):
        """the block layout of the next batch: the capacity layout for the first two batches and after an overflow, then
        what the batch before last needed + margin (that merge's header scan is long done: no stall, and every rank reads
        the same figures, so all ranks pick the same layout without talking to each other)"""
        if not self.adaptive or self.n_steps < 2:
            return self.layout
        _, nq, need, ovf = self.mws.exchange_stats(back=1)
        if self._full_next or ovf:
            self._full_next = False
            return self.layout
        fit = abi.ExchangeLayout()
        abi.check(abi.lib().kaamer_exchange_layout_fit(C.byref(self.layout), int(nq * (1.0 + self.margin)) + 64,
                                                       int(need * (1.0 + self.margin)) + 1024, int(self.ws_first_pos), C.byref(fit)))
        return fit

    def _alltoall(self, stream):
        raw = stream.cuda_stream
        n = self.world * int(self.wire.block_words)   # words that travel: world blocks of THIS batch's layout
        if self.transport == "rccl":  # world 1 included: the send/recv pair with oneself goes through RCCL too
            abi.check(abi.lib().kaamer_rccl_alltoall(self.comm.handle, self.send.data_ptr(), self.recv.data_ptr(),
                                                     4 * int(self.wire.block_words), self.world, C.c_void_p(raw)))
            return
        if self.world == 1:
            return
        if self.transport == "torch":
            with torch.cuda.stream(stream):
                dist.all_to_all_single(self.recv[:n], self.send[:n], group=self.group)
            return
        with torch.cuda.stream(stream):
            self.h_send[:n].copy_(self.send[:n], non_blocking=True)
        stream.synchronize()
        dist.all_to_all_single(self.h_recv[:n], self.h_send[:n], group=self.group)
        with torch.cuda.stream(stream):
            self.recv[:n].copy_(self.h_recv[:n], non_blocking=True)

    def step(self, d_seqs_ptr, d_off_ptr, n_seqs, seq_bytes, stream, topn=None, timed=False, full=False):
        """`stream`: a torch.cuda.Stream.  -> DeviceResult of the merged, owned queries (query i of the result = query
        rank + i * world of the batch).  topn: dict of kaamer_topn_device options to run the post-steps too (result in
        self.last_topn).  timed: bracket every phase with events; their ms are added to self.phase_ms after a sync."""
        raw = stream.cuda_stream
        ev = []

        def mark():
            if timed:
                e = torch.cuda.Event(enable_timing=True)
                e.record(stream)
                ev.append(e)
        if self.direct:
            mark()
            self.last_search = r = self.ws.search_device(d_seqs_ptr, d_off_ptr, n_seqs, seq_bytes, stream=raw)
            mark(); mark(); mark(); mark()
            self.last_topn = self.topn(stream, **topn) if topn is not None else None
            mark()
            self._account(ev, stream, timed)
            return r
        if not full:
            self.wire = self._choose_layout()
        else:
            self.wire = self.layout
        self.n_steps += 1
        mark()
        self.last_search = self.ws.search_device(d_seqs_ptr, d_off_ptr, n_seqs, seq_bytes, stream=raw)
        mark()
        self.ws.exchange_pack(self.wire, self.send.data_ptr(), raw)
        mark()
        self._alltoall(stream)
        mark()
        r = self.mws.exchange_merge(self.wire, self.recv.data_ptr(), raw)
        mark()
        self.last_topn = self.topn(stream, **topn) if topn is not None else None
        mark()
        self._account(ev, stream, timed)
        return r

    def _account(self, ev, stream, timed):
        if not timed:
            return
        stream.synchronize()
        if self.phase_ms is None:
            self.phase_ms = dict.fromkeys(PHASES, 0.0)
            self.phase_ms["batches"] = 0
        for i, name in enumerate(PHASES):
            self.phase_ms[name] += ev[i].elapsed_time(ev[i + 1])
        self.phase_ms["batches"] += 1

    def topn(self, stream, min_k_ratio=0.05, min_k_match=10, max_results=10):
        """the post-steps of the reference's drivers on the merged results (SetBestStartCodon for nucleotide input)"""
        if self.direct:
            return self.ws.topn_device(min_k_ratio, min_k_match, max_results, best_start_codon=self.nucl, stream=stream.cuda_stream)
        return self.mws.topn_device(min_k_ratio, min_k_match, max_results, best_start_codon=self.nucl,
                                    orf_source=self.ws, q_first=self.rank, q_stride=self.world, stream=stream.cuda_stream)

    def finish(self, stream):
        """-> (counters of the local search, counters of the merge).  BOTH workspaces are finished before an error of
        either is raised: a rank whose own search failed still learns what its merge saw, and a rank whose merge was
        handed a failed peer's blocks raises too (every rank of the batch raises, none hangs at the next collective)."""
        if self.direct:
            c = self.ws.finish(stream.cuda_stream)
            return c, c
        err = None
        out = []
        for w in (self.ws, self.mws):
            try:
                out.append(w.finish(stream.cuda_stream))
            except abi.KaamerError as e:
                out.append(None)
                err = err or e
        if err is not None:
            if err.code == abi.E_CAPACITY:
                self._full_next = True   # (every rank raises for the same batch: all fall back together)
            raise err
        return out[0], out[1]

    def run(self, d_seqs_ptr, d_off_ptr, n_seqs, seq_bytes, stream, topn=None):
        """step + finish; a batch whose blocks did not fit the adaptive layout is repeated once at full capacity
        (on every rank: the overflow is in every header).  -> (DeviceResult, counters of the search, of the merge)"""
        r = self.step(d_seqs_ptr, d_off_ptr, n_seqs, seq_bytes, stream, topn=topn)
        try:
            c = self.finish(stream)
        except abi.KaamerError as e:
            if e.code != abi.E_CAPACITY or self.wire is self.layout:
                raise
            r = self.step(d_seqs_ptr, d_off_ptr, n_seqs, seq_bytes, stream, topn=topn, full=True)
            c = self.finish(stream)
        return r, c[0], c[1]

    def close(self):
        if self._own_comm:
            self.comm.close()
        self.ws.close()
        self.mws.close()


class ShardedPipeline:
    """Several sharded steps in flight: `depth` ShardedSearchers, each with its own workspaces, exchange buffers, stream
    and (transport "rccl") communicator, take the batches in turn -- the all-to-all of batch i travels while batch i + 1 is
    searched and batch i - 1 is merged.  Every rank enqueues the searchers in the same order, so the collectives of one
    communicator (or of torch's process group) meet in the same order everywhere.  The block layout of a searcher's next
    batch comes from ITS OWN earlier batches (kaamer_exchange_stats): the same figures on every rank."""

    def __init__(self, depth, *args, **kw):
        assert depth >= 1
        kw = dict(kw, concurrent_batches=depth if depth > 1 else 0)
        self.searchers = [ShardedSearcher(*args, **kw) for _ in range(depth)]
        self.streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(depth - 1)]
        self.n = 0
        self.last_k = 0                       # the searcher that took the last batch
        self.dirty = [False] * depth          # searchers with batches enqueued since their last finish

    def __len__(self):
        return len(self.searchers)

    def step(self, d_seqs_ptr, d_off_ptr, n_seqs, seq_bytes, **kw):
        """enqueue one batch on the next searcher -> (searcher index, DeviceResult)"""
        k = self.n % len(self.searchers)
        self.n += 1
        self.last_k, self.dirty[k] = k, True
        return k, self.searchers[k].step(d_seqs_ptr, d_off_ptr, n_seqs, seq_bytes, self.streams[k], **kw)

    def finish(self):
        """finish every searcher that has batches enqueued (all are finished before the first error is raised)
        -> [(search counters, merge counters) or None], by searcher"""
        out, err = [], None
        for k, (s_, st) in enumerate(zip(self.searchers, self.streams)):
            if not self.dirty[k]:
                out.append(None)
                continue
            self.dirty[k] = False
            try:
                out.append(s_.finish(st))
            except abi.KaamerError as e:
                out.append(None)
                err = err or e
        if err is not None:
            raise err
        return out

    def set_adaptive(self, on):
        for s_ in self.searchers:
            s_.adaptive = bool(on)

    def close(self):
        for s_ in self.searchers:
            s_.close()
