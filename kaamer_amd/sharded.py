"""Sharded index over several devices: one process per GPU.

The table is partitioned by hash prefix (`kaamer_shard_of`); rank r holds shard r.  Per
batch, every rank searches the WHOLE query batch against its shard (the kernels skip the
keys they do not own), which yields partial hit lists `(protein id, partial Kmatch, first
position)` per query.  Query q is owned by rank `q % world`: one all-to-all (RCCL over
xGMI: every rank talks to every other rank at once, all seven links busy — not a ring)
moves each partial list to its owner, and the owner merges them on the device
(`kaamer_merge_device`: integer sums, so the result is bit-identical to the one-GPU path).

The product path is `ShardedSearcher`: C-ABI calls only (search, kaamer_exchange_pack,
kaamer_exchange_merge, kaamer_topn_device) around ONE collective, no host synchronisation.
The torch functions before it (`build_send`, `exchange`, `to_query_major`) are a
device-agnostic restatement of the same routing with variable-size messages: they let the
N>1 routing be tested with the gloo backend on CPU, where no kernel can run.
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import abi, api


def owner_perm(nq, world, device):
    """queries grouped by owner rank (q % world), ascending inside a group"""
    return torch.cat([torch.arange(d, nq, world, device=device) for d in range(world)]) if nq else \
        torch.zeros(0, dtype=torch.int64, device=device)


def n_owned(nq, world, rank):
    return len(range(rank, nq, world))


def _ranges_index(starts, counts):
    """index tensor that concatenates [starts[i], starts[i]+counts[i])"""
    total = int(counts.sum())
    if total == 0:
        return torch.zeros(0, dtype=torch.int64, device=starts.device)
    dst = torch.cumsum(counts, 0) - counts
    return torch.repeat_interleave(starts - dst, counts) + torch.arange(total, device=starts.device)


def build_send(hit_off, hit_cnt, pid, km, fp, world):
    """partial hit lists of ALL queries (first hit, count per query) -> buffers ordered by destination rank.

    returns (cnt_p [nq] int64: per-query counts in owner order,
             ents [n, 3] int32-like: the entries in the same order,
             q_splits, e_splits: per-destination numbers of queries / entries)"""
    nq = hit_cnt.numel()
    dev = hit_off.device
    cnt = hit_cnt.to(torch.int64)
    perm = owner_perm(nq, world, dev)
    cnt_p = cnt[perm]
    idx = _ranges_index(hit_off[:nq].to(torch.int64)[perm], cnt_p)
    ents = torch.stack([pid[idx], km[idx], fp[idx]], dim=1) if idx.numel() else \
        torch.zeros((0, 3), dtype=pid.dtype, device=dev)
    q_splits = [n_owned(nq, world, d) for d in range(world)]
    bounds = np.cumsum([0] + q_splits)
    csum = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(cnt_p, 0)])
    edges = csum[torch.as_tensor(bounds, dtype=torch.int64, device=dev)].tolist()  # one host sync for all destinations
    e_splits = [int(edges[d + 1] - edges[d]) for d in range(world)]
    return cnt_p, ents.contiguous(), q_splits, e_splits


def exchange(cnt_p, ents, q_splits, e_splits, rank, world, group=None):
    """the one exchange step: all-to-all of the per-query counts, then of the entries.

    returns (recv_cnt [world, n_owned] int64, recv_ents [m, 3]) with the entries in
    source-major order (source 0's lists for my queries, then source 1's, ...)"""
    mine = q_splits[rank]
    recv_cnt = torch.empty(world * mine, dtype=cnt_p.dtype, device=cnt_p.device)
    dist.all_to_all_single(recv_cnt, cnt_p.contiguous(), output_split_sizes=[mine] * world,
                           input_split_sizes=q_splits, group=group)
    recv_cnt = recv_cnt.view(world, mine)
    in_splits = [int(x) for x in recv_cnt.sum(1).tolist()]
    recv_ents = torch.empty((sum(in_splits), 3), dtype=ents.dtype, device=ents.device)
    dist.all_to_all_single(recv_ents, ents, output_split_sizes=in_splits, input_split_sizes=e_splits, group=group)
    return recv_cnt, recv_ents


def to_query_major(recv_cnt, recv_ents):
    """source-major received entries -> per-query contiguous (what kaamer_merge_device reads).

    returns (ent_off [n_owned + 1] int64, ents [m, 3])"""
    world, mine = recv_cnt.shape
    dev = recv_cnt.device
    src_base = torch.cumsum(recv_cnt.sum(1), 0) - recv_cnt.sum(1)                  # first entry of each source block
    seg_start = src_base[:, None] + torch.cumsum(recv_cnt, 1) - recv_cnt           # [world, mine] start of (source, query)
    # query-major order of the (query, source) segments
    starts = seg_start.t().reshape(-1)
    counts = recv_cnt.t().reshape(-1)
    idx = _ranges_index(starts, counts)
    tot_q = recv_cnt.sum(0)
    ent_off = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(tot_q, 0)])
    return ent_off, (recv_ents[idx] if idx.numel() else recv_ents[:0]).contiguous()


class _DevView:
    """zero-copy torch view of a raw device pointer owned by the C library"""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (int(ptr), False), "version": 3}


def dev_tensor(ptr, n, dtype):
    typestr = {torch.int32: "<i4", torch.int64: "<i8"}[dtype]
    if n == 0:
        return torch.zeros(0, dtype=dtype, device="cuda")
    return torch.as_tensor(_DevView(ptr, n, typestr), device="cuda")


class ShardedSearcher:
    """Rank-local driver of the sharded index: search my shard, pack, all-to-all, merge my queries, post-steps.
    Every step is a C-ABI call enqueued on one stream (kaamer_search_device, kaamer_exchange_pack,
    kaamer_exchange_merge, kaamer_topn_device); the transport is torch.distributed's all_to_all_single on
    the packed blocks with EQUAL splits (backend nccl = RCCL: grouped send/recv over xGMI), so a step has no
    host synchronisation at all.  Works for protein and for nucleotide / reads input (every rank translates;
    the owner's post-steps use its own ORFs: orf_source)."""

    def __init__(self, index, rank, world, max_seq_bytes, max_seqs, seq_type=abi.PROTEIN, max_entries_per_peer=1 << 20,
                 group=None, max_hits=0, g_tier_slots=0, first_pos=None):
        """first_pos: carry the lowest matching position of every hit through the exchange.  Default: as the reference
        fills PositionHits (search.go:416) -- nucleotide / reads input yes (SetBestStartCodon reads it), protein input no
        (a third less to pack, send, unpack and merge)."""
        self.index, self.rank, self.world, self.group = index, rank, world, group
        self.nucl = seq_type in (abi.READS, abi.NUCLEOTIDE)
        if first_pos is None:
            first_pos = self.nucl
        fp = 1 if first_pos else 2
        self.ws = api.Workspace(index, max_seq_bytes, max_seqs, seq_type=seq_type, first_pos=fp, max_hits=max_hits,
                                g_tier_slots=g_tier_slots)
        self.layout = abi.ExchangeLayout()
        abi.check(abi.lib().kaamer_exchange_layout_init(world, rank, self.ws.query_capacity, max_entries_per_peer,
                                                        C.byref(self.layout)))
        L = self.layout
        self.mws = api.Workspace(index, 64, L.q_cap, max_queries=L.q_cap, first_pos=fp, max_hits=world * L.e_cap,
                                 g_tier_slots=g_tier_slots)
        n = world * int(L.block_words)
        self.send = torch.empty(n, dtype=torch.int32, device="cuda")
        self.recv = torch.empty(n, dtype=torch.int32, device="cuda") if world > 1 else self.send

    def step(self, d_seqs_ptr, d_off_ptr, n_seqs, seq_bytes, stream):
        """`stream`: a torch.cuda.Stream (the collective is issued under it).  -> DeviceResult of the merged,
        owned queries (query i of the result = query rank + i * world of the batch)"""
        raw = stream.cuda_stream
        self.ws.search_device(d_seqs_ptr, d_off_ptr, n_seqs, seq_bytes, stream=raw)
        self.ws.exchange_pack(self.layout, self.send.data_ptr(), raw)
        if self.world > 1:
            with torch.cuda.stream(stream):
                dist.all_to_all_single(self.recv, self.send, group=self.group)
        return self.mws.exchange_merge(self.layout, self.recv.data_ptr(), raw)

    def topn(self, stream, min_k_ratio=0.05, min_k_match=10, max_results=10):
        """the post-steps of the reference's drivers on the merged results (SetBestStartCodon for nucleotide input)"""
        return self.mws.topn_device(min_k_ratio, min_k_match, max_results, best_start_codon=self.nucl,
                                    orf_source=self.ws, q_first=self.rank, q_stride=self.world, stream=stream.cuda_stream)

    def finish(self, stream):
        """-> (counters of the local search, counters of the merge); raises on a capacity overflow"""
        c = self.ws.finish(stream.cuda_stream)
        return c, self.mws.finish(stream.cuda_stream)
