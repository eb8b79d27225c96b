"""Device-agnostic restatement of the sharded exchange's routing with variable-size messages (torch only), kept
under tests/: it lets the q mod W ownership arithmetic be exercised with the gloo backend on CPU, where no kernel can
run.  The product path is kaamer_amd.sharded.ShardedSearcher (C-ABI calls around one equal-split collective); its
block format is restated in numpy in tests/blockfmt.py and checked against the device's own blocks."""
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
