"""Sharded index: the exchange plumbing with world_size-2 gloo on CPU, and the
device merge kernel with two shards on one GPU."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _partial_csr(oracle, oix, seqs):
    """oracle partial hit lists of one shard -> CSR arrays (hit_off, pid, km, fp)"""
    off, pid, km, fp = [0], [], [], []
    for s in seqs:
        if oracle.size_in_kmer(s) >= 7:
            p, k, pos = oix.search(s, want_positions=True)
            pid += p.tolist(); km += k.tolist(); fp += [int(np.argmax(pos[i])) for i in range(len(p))]
        off.append(len(pid))
    return (np.array(off, np.int64), np.array(pid, np.int32), np.array(km, np.int32), np.array(fp, np.int32))


def _numpy_merge(ent_off, ents):
    out = []
    for q in range(len(ent_off) - 1):
        d = {}
        for pid, km, fp in ents[ent_off[q]:ent_off[q + 1]].tolist():
            a = d.get(pid, (0, 1 << 30))
            d[pid] = (a[0] + km, min(a[1], fp))
        out.append(d)
    return out


def _shard_pairs(klib, oracle, db, shard, n_shards):
    full = oracle.Index.from_proteins(None, packed=db).pairs()
    keys = (full >> 32).astype(np.uint32)
    ids = (full & 0xFFFFFFFF).astype(np.uint32)
    sel = np.array([klib.kaamer_shard_of(int(k), n_shards) == shard for k in np.unique(keys)])
    owned = set(np.unique(keys)[sel].tolist())
    m = np.array([int(k) in owned for k in keys])
    return keys[m], ids[m]


def _gloo_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from kaamer_amd import abi, sharded, workload
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        klib = abi.lib()
        db = workload.make_db(60, seed=3)
        seqs = workload.unpack(workload.make_protein_queries(db, 23, seed=4)) + [b"AAAA", b""]
        k, i = _shard_pairs(klib, O, db, rank, world)
        oix = O.Index.from_pairs(k, i)
        off, pid, km, fp = (torch.from_numpy(x) for x in _partial_csr(O, oix, seqs))
        cnt_p, ents, qs, es = sharded.build_send(off, (off[1:] - off[:-1]).to(torch.int32), pid, km, fp, world)
        recv_cnt, recv_ents = sharded.exchange(cnt_p, ents, qs, es, rank, world)
        ent_off, q_ents = sharded.to_query_major(recv_cnt, recv_ents)
        merged = _numpy_merge(ent_off.numpy(), q_ents.numpy())
        full = O.Index.from_proteins(None, packed=db)
        exp_off, epid, ekm, efp = _partial_csr(O, full, seqs)
        owned = list(range(rank, len(seqs), world))
        assert len(merged) == len(owned)
        n = 0
        for j, q in enumerate(owned):
            exp = {int(p): (int(c), int(f)) for p, c, f in zip(epid[exp_off[q]:exp_off[q + 1]], ekm[exp_off[q]:exp_off[q + 1]],
                                                             efp[exp_off[q]:exp_off[q + 1]])}
            assert merged[j] == exp, (rank, q)
            n += len(exp)
        ret[rank] = n
    finally:
        dist.destroy_process_group()


def test_exchange_gloo_world2(klib, oracle):
    import torch.multiprocessing as mp
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, ret)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(180)
        assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
        assert len(ret) == 2 and sum(ret.values()) > 50


def test_exchange_helpers_single_process():
    """routing identities of build_send / to_query_major without a process group"""
    import torch
    from kaamer_amd import sharded
    rng = np.random.default_rng(0)
    world, nq = 3, 17
    cnt = rng.integers(0, 5, nq)
    off = torch.from_numpy(np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64))
    n = int(off[-1])
    pid = torch.arange(n, dtype=torch.int32) * 7
    km = torch.arange(n, dtype=torch.int32) + 1
    fp = torch.arange(n, dtype=torch.int32) % 5
    cnt_p, ents, qs, es = sharded.build_send(off, torch.from_numpy(cnt.astype(np.int32)), pid, km, fp, world)
    assert qs == [6, 6, 5] and sum(es) == n and cnt_p.tolist() == [int(cnt[q]) for d in range(world) for q in range(d, nq, world)]
    # destination 1 receives, from this single source, the lists of queries 1, 4, 7, ...
    qb, eb = np.cumsum([0] + qs), np.cumsum([0] + es)
    recv_cnt = cnt_p[qb[1]:qb[2]].view(1, -1)
    recv = ents[eb[1]:eb[2]]
    ent_off, q_ents = sharded.to_query_major(recv_cnt, recv)
    exp = torch.cat([pid[off[q]:off[q + 1]] for q in range(1, nq, world)])
    assert torch.equal(q_ents[:, 0], exp) and ent_off[-1] == len(exp)


@pytest.mark.gpu
def test_two_shards_merge_on_one_gpu(klib, oracle, gpu_device):
    """shard the table in two, search both shards, route the partial lists in-process, merge on the device"""
    import torch
    from kaamer_amd import api, sharded, workload
    world = 2
    db = workload.make_db(400, seed=6)
    seqs = workload.unpack(workload.make_protein_queries(db, 150, seed=7))
    seqs += [max(workload.unpack(db), key=len), b"AAAAAAA", b""]
    n_seqs = len(seqs)
    buf, offs = api.pack_sequences(seqs)
    d_buf = torch.from_numpy(buf).cuda()
    d_off = torch.from_numpy(offs.view(np.int64)).cuda()
    st = torch.cuda.current_stream().cuda_stream
    sends, keep = [], []
    total_lookups = 0
    for r in range(world):
        ix = api.Index.from_image(api.Image.from_proteins(packed=db, shard=r, n_shards=world), gpu_device)
        ws = api.Workspace(ix, len(buf), n_seqs, first_pos=1)
        res = ws.search_device(d_buf.data_ptr(), d_off.data_ptr(), n_seqs, len(buf), stream=st)
        c = ws.finish(st)
        total_lookups += c["n_lookup"]
        nh = int(res.hit_capacity)
        hit_off = sharded.dev_tensor(res.d_hit_off, n_seqs, torch.int64)
        sends.append(sharded.build_send(hit_off, sharded.dev_tensor(res.d_hit_cnt, n_seqs, torch.int32),
                                        sharded.dev_tensor(res.d_hit_pid, nh, torch.int32),
                                        sharded.dev_tensor(res.d_hit_kmatch, nh, torch.int32),
                                        sharded.dev_tensor(res.d_hit_first_pos, nh, torch.int32), world))
        keep.append((ix, ws))
    # every k-mer is owned by exactly one shard
    assert total_lookups == sum(max(0, oracle.size_in_kmer(s)) for s in seqs if oracle.size_in_kmer(s) >= 7)
    full = oracle.Index.from_proteins(None, packed=db)
    ix0, ws0 = keep[0]
    mws = api.Workspace(ix0, len(buf), n_seqs, first_pos=1, max_hits=1 << 20)
    for d in range(world):
        rc, re = [], []
        for (cnt_p, ents, qs, es) in sends:
            qb, eb = np.cumsum([0] + qs), np.cumsum([0] + es)
            rc.append(cnt_p[qb[d]:qb[d + 1]])
            re.append(ents[eb[d]:eb[d + 1]])
        ent_off, q_ents = sharded.to_query_major(torch.stack(rc), torch.cat(re))
        cols = [q_ents[:, i].contiguous() for i in range(3)]
        m = mws.merge_device(ent_off.data_ptr(), cols[0].data_ptr(), cols[1].data_ptr(), cols[2].data_ptr(),
                             ent_off.numel() - 1, int(q_ents.shape[0]), stream=st)
        c = mws.finish(st)
        owned = list(range(d, n_seqs, world))
        hit_off = sharded.dev_tensor(m.d_hit_off, len(owned), torch.int64).cpu().numpy()
        hit_cnt = sharded.dev_tensor(m.d_hit_cnt, len(owned), torch.int32).cpu().numpy()
        nh = int(m.hit_capacity)
        assert int(hit_cnt.sum()) == c["n_hits"]
        pid = sharded.dev_tensor(m.d_hit_pid, nh, torch.int32).cpu().numpy().view(np.uint32)
        km = sharded.dev_tensor(m.d_hit_kmatch, nh, torch.int32).cpu().numpy()
        fp = sharded.dev_tensor(m.d_hit_first_pos, nh, torch.int32).cpu().numpy()
        for j, q in enumerate(owned):
            exp = {}
            if oracle.size_in_kmer(seqs[q]) >= 7:
                p, k, pos = full.search(seqs[q], want_positions=True)
                exp = {int(a): (int(b), int(np.argmax(pos[i]))) for i, (a, b) in enumerate(zip(p, k))}
            a, b = int(hit_off[j]), int(hit_off[j]) + int(hit_cnt[j])
            got = {int(x): (int(y), int(z)) for x, y, z in zip(pid[a:b], km[a:b], fp[a:b])}
            assert got == exp, (d, q)
