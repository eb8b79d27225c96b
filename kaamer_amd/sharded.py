"""Sharded index over several devices: one process per GPU.

The table is partitioned by hash prefix (`kaamer_shard_of`); rank r holds shard r.  Per
batch, every rank searches the WHOLE query batch against its shard (the kernels skip the
keys they do not own), which yields partial hit lists `(protein id, partial Kmatch, first
position)` per query.  Query q is owned by rank `q % world`: one all-to-all (RCCL over
xGMI: every rank talks to every other rank at once, all seven links busy — not a ring)
moves each partial list to its owner, and the owner merges them on the device
(`kaamer_merge_device`: integer sums, so the result is bit-identical to the one-GPU path).

Everything here is plumbing: index arithmetic on torch tensors and `torch.distributed`
calls; the search and the merge are the C-ABI kernels.  The exchange helpers are
device-agnostic so the N>1 path is testable with the gloo backend on CPU.
"""
import numpy as np
import torch
import torch.distributed as dist


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
    """rank-local driver: search my shard, exchange, merge my queries"""

    def __init__(self, index, search_ws, merge_ws, rank, world, group=None):
        self.index, self.ws, self.mws = index, search_ws, merge_ws
        self.rank, self.world, self.group = rank, world, group

    def step(self, d_buf, d_off, n_seqs, seq_bytes, stream):
        """-> (ent_off, merged DeviceResult, counters of the local search)"""
        r = self.ws.search_device(d_buf.data_ptr(), d_off.data_ptr(), n_seqs, seq_bytes, stream=stream)
        c = self.ws.finish(stream)
        cap = int(r.hit_capacity)  # the lists sit where the counting kernel put them (hit_off, hit_cnt)
        hit_off = dev_tensor(r.d_hit_off, n_seqs, torch.int64)
        hit_cnt = dev_tensor(r.d_hit_cnt, n_seqs, torch.int32)
        pid = dev_tensor(r.d_hit_pid, cap, torch.int32)
        km = dev_tensor(r.d_hit_kmatch, cap, torch.int32)
        fp = dev_tensor(r.d_hit_first_pos, cap, torch.int32)
        cnt_p, ents, qs, es = build_send(hit_off, hit_cnt, pid, km, fp, self.world)
        recv_cnt, recv_ents = exchange(cnt_p, ents, qs, es, self.rank, self.world, self.group)
        ent_off, q_ents = to_query_major(recv_cnt, recv_ents)
        cols = [q_ents[:, i].contiguous() for i in range(3)] if q_ents.numel() else \
            [torch.zeros(1, dtype=torch.int32, device=ents.device)] * 3
        self._keep = (ent_off, cols)  # alive until the merge kernels have run
        m = self.mws.merge_device(ent_off.data_ptr(), cols[0].data_ptr(), cols[1].data_ptr(), cols[2].data_ptr(),
                                  ent_off.numel() - 1, int(q_ents.shape[0]), stream=stream)
        return ent_off, m, c
