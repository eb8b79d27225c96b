"""Sharded index.

CPU (`-m "not gpu"`): world_size-2 gloo runs of the exchange -- the product's fixed-size block format (restated in
numpy in tests/blockfmt.py, sized by the library's own kaamer_exchange_layout_init) carried by the product's transport
pattern (one all_to_all_single with EQUAL splits), and the variable-size routing restatement of tests/sharded_ref.py.
No kernel can run there, so the partial lists come from the oracle.

GPU (`-m gpu`): the PRODUCT path -- kaamer_search_device -> kaamer_exchange_pack -> all-to-all -> kaamer_exchange_merge
-> kaamer_topn_device -- between two real processes that share GPU 0 (blocks staged through the host and carried by
gloo: RCCL cannot put two ranks on one device), through kaamer_rccl_alltoall on a real one-rank ncclComm_t, and in one
process with the blocks routed by slicing.  Results against the oracle on the WHOLE database."""
import os
import socket
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


def _partial_csr_orfs(oracle, oix, orfs):
    off, pid, km, fp = [0], [], [], []
    for o in orfs:
        p, k, pos = oix.search(o["seq"], want_positions=True)
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


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    return port


def _spawn(target, world, args=(), timeout=600):
    """world processes of `target(rank, world, port, ret, *args)`; -> ret as a dict"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    port = _free_port()
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        procs = [ctx.Process(target=target, args=(r, world, port, ret) + tuple(args)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout)
        alive = [p for p in procs if p.is_alive()]
        for p in alive:
            p.terminate()
        assert not alive, "a rank did not finish"
        assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
        return dict(ret)


def _init_gloo(rank, world, port):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _gloo_worker_blocks(rank, world, port, ret, reads):
    """CPU: the block format + the equal-split collective between two ranks; partial lists from the oracle"""
    for p_ in (ROOT, os.path.join(ROOT, "tests")):
        sys.path.insert(0, p_)
    import ctypes as C
    import torch
    import torch.distributed as dist
    import blockfmt
    from kaamer_amd import abi, workload
    from oracle import oracle as O
    _init_gloo(rank, world, port)
    try:
        klib = abi.lib()
        db = workload.make_db(80, seed=3)
        if reads:
            rl = workload.unpack(workload.make_reads(db, 40, seed=5))
            queries = [o for r in rl for o in O.get_orfs(r)]          # every rank translates: same ORFs, same order
        else:
            queries = workload.unpack(workload.make_protein_queries(db, 23, seed=4)) + [b"AAAA", b""]
        k, i = _shard_pairs(klib, O, db, rank, world)
        oix = O.Index.from_pairs(k, i)
        off, pid, km, fp = _partial_csr_orfs(O, oix, queries) if reads else _partial_csr(O, oix, queries)
        nq = len(queries)
        L = abi.ExchangeLayout()
        abi.check(klib.kaamer_exchange_layout_init(world, rank, nq, 1 << 12, C.byref(L)))
        send = blockfmt.pack_blocks(L, nq, off, np.diff(off), pid.view(np.uint32), km.view(np.uint32), fp.view(np.uint32), True)
        t_send = torch.from_numpy(send.view(np.int32))
        t_recv = torch.empty_like(t_send)
        dist.all_to_all_single(t_recv, t_send)                       # equal splits: world blocks of block_words words
        merged = blockfmt.unpack_merge(L, t_recv.numpy().view(np.uint32), True)
        full = O.Index.from_proteins(None, packed=db)
        owned = list(range(rank, nq, world))
        assert len(merged) == len(owned)
        n = 0
        for j, q in enumerate(owned):
            exp_full, exp_rep = _oracle_report(O, full, queries[q], reads)
            assert merged[j] == exp_full, (rank, q)
            if reads:  # post-steps on the merged list, host entry points of the product library
                o = queries[q]
                ids = np.array(sorted(merged[j]), np.uint32)
                kms = np.array([merged[j][int(p)][0] for p in ids], np.uint32)
                fps = np.array([merged[j][int(p)][1] for p in ids], np.uint32)
                order = np.zeros(len(ids), np.uint32)
                klib.kaamer_sort_hits(ids.ctypes.data, kms.ctypes.data, len(ids), order.ctypes.data)
                ids, kms, fps = ids[order], kms[order], fps[order]
                keep = 0
                if len(kms) and kms[0] >= 10:                      # search_fastq.go:119
                    sa = np.array(o["starts"], np.int32)
                    aa = o["seq"].encode("latin-1")
                    sp, sz = C.c_int32(o["start"]), C.c_int32(O.size_in_kmer(o["seq"]))
                    klib.kaamer_set_best_start_codon(kms.ctypes.data, fps.ctypes.data, len(kms), sa.ctypes.data, len(sa), int(o["plus"]),
                                                     aa, len(aa), C.byref(sp), C.byref(sz))
                    keep = int(klib.kaamer_filter_results(kms.ctypes.data, len(kms), sz.value, 0.05, 10, 10))
                assert list(zip(ids[:keep].tolist(), kms[:keep].tolist())) == exp_rep, (rank, q)
            n += len(exp_full)
        # a sender whose search failed: every owner refuses its blocks
        bad = blockfmt.pack_blocks(L, nq, off, np.diff(off), pid.view(np.uint32), km.view(np.uint32), fp.view(np.uint32), True,
                                   src_status=1 if rank == 1 else 0)
        t_bad = torch.from_numpy(bad.view(np.int32))
        dist.all_to_all_single(t_recv, t_bad)
        with pytest.raises(ValueError):
            blockfmt.unpack_merge(L, t_recv.numpy().view(np.uint32), True)
        ret[rank] = n
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("reads", [False, True], ids=["protein", "reads"])
def test_exchange_gloo_world2(klib, oracle, reads):
    ret = _spawn(_gloo_worker_blocks, 2, (reads,), timeout=180)
    assert len(ret) == 2 and sum(ret.values()) > 50


def test_exchange_helpers_single_process():
    """routing identities of build_send / to_query_major without a process group"""
    import torch
    import sharded_ref as sharded
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




def _route(sends, layouts, d):
    """what rank d receives: block s = the block rank s packed for rank d (an all-to-all with equal splits)"""
    import torch
    bw = int(layouts[0].block_words)
    return torch.cat([sends[s][d * bw:(d + 1) * bw] for s in range(len(sends))])


def _oracle_report(oracle, oix, orf, reads):
    """the reference's per-query block on the WHOLE database: (hit map with first positions, reported hits)"""
    if reads:
        seq, starts, plus, start = orf["seq"], orf["starts"], orf["plus"], orf["start"]
    else:
        seq = orf
    size = oracle.size_in_kmer(seq)
    if size < 7 and not reads:
        return {}, []
    pid, km, pos = oix.search(seq, want_positions=True)
    full = {int(p): (int(k), int(np.argmax(pos[i]))) for i, (p, k) in enumerate(zip(pid, km))}
    keep = 0
    if reads:
        if len(km) and km[0] >= 10:
            _, _, size = oracle.set_best_start_codon(km, pos, size, starts, plus, seq, start)
            keep = oracle.filter_results(km, size)
    elif len(km):
        keep = oracle.filter_results(km, size)
    return full, list(zip(pid[:keep].tolist(), km[:keep].tolist()))


@pytest.mark.gpu
@pytest.mark.parametrize("reads,first_pos", [(False, True), (False, False), (True, True)])
def test_sharded_through_the_c_abi(klib, oracle, gpu_device, reads, first_pos):
    """W = 2 shards on one GPU, every step through the C ABI: search each shard, kaamer_exchange_pack, route the
    blocks as the all-to-all would, kaamer_exchange_merge on each owner, kaamer_topn_device with orf_source --
    for protein queries (with first positions, and without: the protein default, where the exchange leaves that
    third of the entries alone) and for reads (every rank translates; SetBestStartCodon on the owner).  Also W = 1
    (send buffer = receive buffer), which must reproduce the unsharded results."""
    import ctypes as C
    import torch
    from kaamer_amd import abi, api, sharded, workload
    db = workload.make_db(600, seed=6)
    full = oracle.Index.from_proteins(None, packed=db)
    if reads:
        q = workload.make_reads(db, 300, seed=12)
        rl = workload.unpack(q)
        queries = [o for r in rl for o in oracle.get_orfs(r)]
        seq_type = abi.READS
    else:
        seqs = workload.unpack(workload.make_protein_queries(db, 150, seed=7)) + [max(workload.unpack(db), key=len), b"AAAAAAA", b""]
        q = api.pack_sequences(seqs)
        queries = seqs
        seq_type = abi.PROTEIN
    buf, offs = q
    n_seqs = len(offs) - 1
    d_buf = torch.from_numpy(buf).cuda()
    d_off = torch.from_numpy(offs.view(np.int64)).cuda()
    st = torch.cuda.current_stream()
    exp = [_oracle_report(oracle, full, x, reads) for x in queries]
    for world in (1, 2, 8):   # 8: the q mod W ownership, the block indexing and the W-way unpack at configs[3]'s width
        ranks = []
        for r in range(world):
            ix = api.Index.from_image(api.Image.from_proteins(packed=db, shard=r, n_shards=world), gpu_device)
            ranks.append((ix, sharded.ShardedSearcher(ix, r, world, len(buf), n_seqs, seq_type=seq_type, max_entries_per_peer=1 << 16,
                                                      first_pos=first_pos)))
        # search + pack on every "rank" (the searcher's own step() does the same, then the collective)
        for ix, ss in ranks:
            ss.ws.search_device(d_buf.data_ptr(), d_off.data_ptr(), n_seqs, len(buf), stream=st.cuda_stream)
            ss.ws.exchange_pack(ss.layout, ss.send.data_ptr(), st.cuda_stream)
            c = ss.ws.finish(st.cuda_stream)
            assert c["n_queries"] == (len(queries) if reads else sum(1 for x in queries if oracle.size_in_kmer(x) >= 7))
        n_checked = 0
        for d, (ix, ss) in enumerate(ranks):
            recv = _route([x[1].send for x in ranks], [x[1].layout for x in ranks], d)
            m = ss.mws.exchange_merge(ss.layout, recv.data_ptr(), st.cuda_stream)
            t = ss.topn(st)
            c = ss.mws.finish(st.cuda_stream)
            owned = list(range(d, len(queries), world))
            hit_off = sharded.dev_tensor(m.d_hit_off, len(owned), torch.int64).cpu().numpy()
            hit_cnt = sharded.dev_tensor(m.d_hit_cnt, len(owned), torch.int32).cpu().numpy()
            nh = int(m.hit_capacity)
            assert int(hit_cnt.sum()) == c["n_hits"]
            pid = sharded.dev_tensor(m.d_hit_pid, nh, torch.int32).cpu().numpy().view(np.uint32)
            km = sharded.dev_tensor(m.d_hit_kmatch, nh, torch.int32).cpu().numpy()
            fp = sharded.dev_tensor(m.d_hit_first_pos, nh, torch.int32).cpu().numpy()
            tc = sharded.dev_tensor(t.d_top_cnt, len(owned), torch.int32).cpu().numpy()
            tp = sharded.dev_tensor(t.d_top_pid, len(owned) * 10, torch.int32).cpu().numpy().view(np.uint32).reshape(-1, 10)
            tk = sharded.dev_tensor(t.d_top_kmatch, len(owned) * 10, torch.int32).cpu().numpy().reshape(-1, 10)
            for j, qi in enumerate(owned):
                a, b = int(hit_off[j]), int(hit_off[j]) + int(hit_cnt[j])
                got = {int(x): (int(y), int(z)) for x, y, z in zip(pid[a:b], km[a:b], fp[a:b])}
                if first_pos:
                    assert got == exp[qi][0], (world, d, qi)
                else:  # no first positions asked for: the field reads as zeros
                    assert got == {k_: (v_[0], 0) for k_, v_ in exp[qi][0].items()}, (world, d, qi)
                k = len(exp[qi][1])
                assert int(tc[j]) == k, (world, d, qi)
                assert list(zip(tp[j, :k].tolist(), tk[j, :k].tolist())) == exp[qi][1], (world, d, qi)
                n_checked += len(got)
        assert n_checked > 500
    # a block too small for the partial lists is reported, never a partial result
    ix, ss = ranks[0]
    small = sharded.ShardedSearcher(ix, 0, 2, len(buf), n_seqs, seq_type=seq_type, max_entries_per_peer=64, first_pos=first_pos)
    small.ws.search_device(d_buf.data_ptr(), d_off.data_ptr(), n_seqs, len(buf), stream=st.cuda_stream)
    small.ws.exchange_pack(small.layout, small.send.data_ptr(), st.cuda_stream)
    small.mws.exchange_merge(small.layout, small.send.data_ptr(), st.cuda_stream)
    with pytest.raises(abi.KaamerError) as e:
        small.mws.finish(st.cuda_stream)
    assert e.value.code == abi.E_CAPACITY
    if not first_pos:
        # blocks packed without first positions handed to an owner that wants them: reported, not merged as zeros
        ix, ss = ranks[0]
        want = sharded.ShardedSearcher(ix, 0, 1, len(buf), n_seqs, seq_type=seq_type, max_entries_per_peer=1 << 16, first_pos=True)
        ss1 = sharded.ShardedSearcher(ix, 0, 1, len(buf), n_seqs, seq_type=seq_type, max_entries_per_peer=1 << 16, first_pos=False)
        ss1.ws.search_device(d_buf.data_ptr(), d_off.data_ptr(), n_seqs, len(buf), stream=st.cuda_stream)
        ss1.ws.exchange_pack(ss1.layout, ss1.send.data_ptr(), st.cuda_stream)
        ss1.ws.finish(st.cuda_stream)
        want.mws.exchange_merge(want.layout, ss1.send.data_ptr(), st.cuda_stream)
        with pytest.raises(abi.KaamerError):
            want.mws.finish(st.cuda_stream)


@pytest.mark.gpu
def test_two_shards_merge_on_one_gpu(klib, oracle, gpu_device):
    """shard the table in two, search both shards, route the partial lists in-process, merge on the device"""
    import torch
    import sharded_ref
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
        sends.append(sharded_ref.build_send(hit_off, sharded.dev_tensor(res.d_hit_cnt, n_seqs, torch.int32),
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
        ent_off, q_ents = sharded_ref.to_query_major(torch.stack(rc), torch.cat(re))
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


def _gpu_worker(rank, world, port, ret, scenario):
    """GPU: one of `world` processes that share GPU 0.  The product's ShardedSearcher end to end; blocks staged through
    the host and carried by gloo.  scenario: "protein" | "reads" | "fail" (rank 1's G-tier arena is too small)."""
    for p_ in (ROOT, os.path.join(ROOT, "tests")):
        sys.path.insert(0, p_)
    import torch
    import torch.distributed as dist
    import blockfmt
    from kaamer_amd import abi, api, sharded, workload
    from oracle import oracle as O
    _init_gloo(rank, world, port)
    try:
        torch.cuda.set_device(0)
        reads = scenario == "reads"
        g_slots = 0
        if scenario == "fail":
            rng = np.random.default_rng(4)
            alpha = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)
            core = bytes(alpha[rng.integers(0, 20, 40)])
            seqs_db = [bytes(alpha[rng.integers(0, 20, 8)]) + core[(i % 5):] + bytes(alpha[rng.integers(0, 20, 8)]) for i in range(9000)]
            db = api.pack_sequences(seqs_db)
            queries = [core, seqs_db[7], b"ACDEFGHIKLMNPQRSTVWY"]
            q = api.pack_sequences(queries)
            seq_type = abi.PROTEIN
            g_slots = 4096 if rank == 1 else 0   # 9000 distinct hits per shard for `core`: needs the HBM tier
        else:
            db = workload.make_db(600, seed=6)
            if reads:
                q = workload.make_reads(db, 300, seed=12)
                queries = [o for r in workload.unpack(q) for o in O.get_orfs(r)]
                seq_type = abi.READS
            else:
                queries = workload.unpack(workload.make_protein_queries(db, 150, seed=7)) + [max(workload.unpack(db), key=len), b"AAAAAAA", b""]
                q = api.pack_sequences(queries)
                seq_type = abi.PROTEIN
        buf, offs = q
        n_seqs = len(offs) - 1
        ix = api.Index.from_image(api.Image.from_proteins(packed=db, shard=rank, n_shards=world), 0)
        ss = sharded.ShardedSearcher(ix, rank, world, len(buf), n_seqs, seq_type=seq_type, max_entries_per_peer=1 << 16,
                                     transport="host", g_tier_slots=g_slots, first_pos=True)
        d_buf = torch.from_numpy(buf).cuda()
        d_off = torch.from_numpy(offs.view(np.int64)).cuda()
        st = torch.cuda.current_stream()
        if scenario == "grow":
            # three small batches size the adaptive blocks; the full batch then outgrows them on EVERY rank (the overflow is
            # in every header), run() repeats it at full capacity, and the result is the whole batch's
            few = api.pack_sequences(queries[:4])
            fb, fo = torch.from_numpy(few[0]).cuda(), torch.from_numpy(few[1].view(np.int64)).cuda()
            for _ in range(3):
                ss.run(fb.data_ptr(), fo.data_ptr(), 4, len(few[0]), st, topn={})
            assert int(ss.wire.e_cap) < int(ss.layout.e_cap)
            m, cs, cm = ss.run(d_buf.data_ptr(), d_off.data_ptr(), n_seqs, len(buf), st, topn={})
            assert ss.wire is ss.layout, "the batch should have been repeated with capacity-sized blocks"
        else:
            m = ss.step(d_buf.data_ptr(), d_off.data_ptr(), n_seqs, len(buf), st, topn={})
        if scenario == "fail":
            # rank 1's search ran out of its arena: BOTH ranks report it (rank 0 through the blocks it received), none hangs
            try:
                ss.finish(st)
                ret[rank] = "no error"
            except abi.KaamerError as e:
                ret[rank] = e.code
            dist.barrier()
            return
        if scenario != "grow":
            cs, cm = ss.finish(st)
            # three more batches: from the third on the blocks are sized from what the batch before last needed (the same
            # figure on every rank, out of the received headers), so what travels is payload, not capacity
            for _ in range(3):
                m = ss.step(d_buf.data_ptr(), d_off.data_ptr(), n_seqs, len(buf), st, topn={})
                cs, cm = ss.finish(st)
            assert int(ss.wire.e_cap) < int(ss.layout.e_cap) and int(ss.wire.q_cap) <= int(ss.layout.q_cap)
            seq_, nq_, need_, ovf_ = ss.mws.exchange_stats(0)
            assert ovf_ == 0 and nq_ == len(queries) and need_ <= int(ss.wire.e_cap) <= int(need_ * 1.25) + 1028
            ret["need%d" % rank] = need_
        # the blocks the device packed == the numpy restatement of the format applied to the device's own partial lists
        r = ss.last_search
        nq = cs["n_queries"] if reads else n_seqs
        assert nq == len(queries)
        cap = int(r.hit_capacity)
        h_off = sharded.dev_tensor(r.d_hit_off, nq, torch.int64).cpu().numpy()
        h_cnt = sharded.dev_tensor(r.d_hit_cnt, nq, torch.int32).cpu().numpy()
        h_pid, h_km, h_fp = (sharded.dev_tensor(x, cap, torch.int32).cpu().numpy().view(np.uint32)
                             for x in (r.d_hit_pid, r.d_hit_kmatch, r.d_hit_first_pos))
        want = blockfmt.pack_blocks(ss.wire, nq, h_off, h_cnt, h_pid, h_km, h_fp, True)
        got = ss.send.cpu().numpy().view(np.uint32)[:len(want)]
        mask = blockfmt.defined_words(ss.wire, want, True)
        assert np.array_equal(got[mask], want[mask]), "device blocks differ from the documented format"
        # merged + reported results of the owned queries vs the oracle on the WHOLE database
        full = O.Index.from_proteins(None, packed=db)
        owned = list(range(rank, len(queries), world))
        t = ss.last_topn
        hit_off = sharded.dev_tensor(m.d_hit_off, len(owned), torch.int64).cpu().numpy()
        hit_cnt = sharded.dev_tensor(m.d_hit_cnt, len(owned), torch.int32).cpu().numpy()
        nh = int(m.hit_capacity)
        assert int(hit_cnt.sum()) == cm["n_hits"]
        pid = sharded.dev_tensor(m.d_hit_pid, nh, torch.int32).cpu().numpy().view(np.uint32)
        km = sharded.dev_tensor(m.d_hit_kmatch, nh, torch.int32).cpu().numpy()
        fp = sharded.dev_tensor(m.d_hit_first_pos, nh, torch.int32).cpu().numpy()
        tc = sharded.dev_tensor(t.d_top_cnt, len(owned), torch.int32).cpu().numpy()
        tp = sharded.dev_tensor(t.d_top_pid, len(owned) * 10, torch.int32).cpu().numpy().view(np.uint32).reshape(-1, 10)
        tk = sharded.dev_tensor(t.d_top_kmatch, len(owned) * 10, torch.int32).cpu().numpy().reshape(-1, 10)
        n = 0
        for j, qi in enumerate(owned):
            exp_full, exp_rep = _oracle_report(O, full, queries[qi], reads)
            a, b = int(hit_off[j]), int(hit_off[j]) + int(hit_cnt[j])
            got_q = {int(x): (int(y), int(z)) for x, y, z in zip(pid[a:b], km[a:b], fp[a:b])}
            assert got_q == exp_full, (rank, qi)
            k = len(exp_rep)
            assert int(tc[j]) == k, (rank, qi)
            assert list(zip(tp[j, :k].tolist(), tk[j, :k].tolist())) == exp_rep, (rank, qi)
            n += len(got_q)
        ret[rank] = n
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("scenario,world", [("protein", 2), ("reads", 2), ("protein", 5), ("reads", 5), ("grow", 2)])
def test_ranks_on_one_gpu_product_path(klib, oracle, gpu_device, scenario, world):
    """configs[3] in small: `world` ranks (OS processes on GPU 0; five ranks + the test runner are the six processes this pool lets one job put on a card),
    each with its hash-prefix shard; the product's search -> kaamer_exchange_pack -> all-to-all -> kaamer_exchange_merge
    -> kaamer_topn_device(orf_source) with the q mod W ownership arithmetic across REAL ranks; four batches, the later
    ones in blocks sized from the earlier ones' need (every rank derives the same size from its received headers);
    merged hit maps, first positions and reported hits of every owned query vs the oracle on the whole database; the
    device's blocks vs the documented format.  "grow": a batch that outgrows its adaptive blocks is repeated at full
    capacity by every rank."""
    ret = _spawn(_gpu_worker, world, (scenario,))
    needs = {v for k, v in ret.items() if str(k).startswith("need")}
    hits = [v for k, v in ret.items() if not str(k).startswith("need")]
    assert len(hits) == world and sum(hits) > 500, ret
    assert len(needs) <= 1, "the ranks disagree on the block size the batch needed: %s" % ret


@pytest.mark.gpu
def test_failed_peer_is_reported_by_every_rank(klib, oracle, gpu_device):
    """rank 1's search exceeds its G-tier arena: its blocks say so, and rank 0 -- whose own search and merge were
    fine -- raises KAAMER_E_CAPACITY too instead of keeping truncated results and hanging at the next collective"""
    from kaamer_amd import abi
    ret = _spawn(_gpu_worker, 2, ("fail",))
    assert ret == {0: abi.E_CAPACITY, 1: abi.E_CAPACITY}, ret


@pytest.mark.gpu
def test_rccl_alltoall_on_a_real_communicator(klib, oracle, gpu_device):
    """kaamer_rccl_alltoall with an ncclComm_t made by ncclGetUniqueId / ncclCommInitRank (one rank: all one GPU
    allows): the block really goes through RCCL's grouped send/recv, and the sharded step over that transport equals
    the unsharded result"""
    import ctypes as C
    import torch
    from kaamer_amd import abi, api, sharded, workload
    comm = sharded.RcclComm(0, 1)
    assert comm.world == 1
    st = torch.cuda.current_stream()
    src = torch.randint(-2**31, 2**31 - 1, (1 << 18,), dtype=torch.int32, device="cuda")
    dst = torch.zeros_like(src)
    abi.check(klib.kaamer_rccl_alltoall(comm.handle, src.data_ptr(), dst.data_ptr(), src.numel() * 4, 1, C.c_void_p(st.cuda_stream)))
    st.synchronize()
    assert torch.equal(src, dst)
    # bad arguments are refused, not passed to RCCL
    assert klib.kaamer_rccl_alltoall(None, src.data_ptr(), dst.data_ptr(), 16, 1, C.c_void_p(st.cuda_stream)) == abi.E_ARG
    db = workload.make_db(400, seed=6)
    seqs = workload.unpack(workload.make_protein_queries(db, 120, seed=7)) + [b"AAAAAAA", b""]
    buf, offs = api.pack_sequences(seqs)
    ix = api.Index.from_image(api.Image.from_proteins(packed=db), gpu_device)
    ss = sharded.ShardedSearcher(ix, 0, 1, len(buf), len(seqs), max_entries_per_peer=1 << 16, transport="rccl", comm=comm)
    d_buf = torch.from_numpy(buf).cuda()
    d_off = torch.from_numpy(offs.view(np.int64)).cuda()
    m = ss.step(d_buf.data_ptr(), d_off.data_ptr(), len(seqs), len(buf), st)
    ss.finish(st)
    full = oracle.Index.from_proteins(None, packed=db)
    hit_off = sharded.dev_tensor(m.d_hit_off, len(seqs), torch.int64).cpu().numpy()
    hit_cnt = sharded.dev_tensor(m.d_hit_cnt, len(seqs), torch.int32).cpu().numpy()
    pid = sharded.dev_tensor(m.d_hit_pid, int(m.hit_capacity), torch.int32).cpu().numpy().view(np.uint32)
    km = sharded.dev_tensor(m.d_hit_kmatch, int(m.hit_capacity), torch.int32).cpu().numpy()
    n = 0
    for i, s in enumerate(seqs):
        exp = {}
        if oracle.size_in_kmer(s) >= 7:
            p, k, _ = full.search(s)
            exp = dict(zip(p.tolist(), k.tolist()))
        a = int(hit_off[i])
        assert dict(zip(pid[a:a + int(hit_cnt[i])].tolist(), km[a:a + int(hit_cnt[i])].tolist())) == exp, i
        n += len(exp)
    assert n > 500
    ss.close()
    comm.close()


@pytest.mark.gpu
@pytest.mark.parametrize("reads", [False, True], ids=["protein", "reads"])
def test_single_process_sharded_handle(klib, oracle, gpu_device, reads):
    """kaamer_index_open_sharded_images + kaamer_sharded_search_batch_top: ONE process drives W shards (here all placed
    on device 0: peer copies become device-to-device copies), the reported queries come back in batch order and equal
    both the unsharded kaamer_search_batch_top and the oracle's per-query block on the whole database; W = 1, 2, 3,
    and the shards saved to files and opened by path."""
    import tempfile
    from kaamer_amd import abi, api, workload
    db = workload.make_db(600, seed=6)
    full = oracle.Index.from_proteins(None, packed=db)
    if reads:
        q = workload.make_reads(db, 300, seed=12)
        queries = [o for r in workload.unpack(q) for o in oracle.get_orfs(r)]
        seq_type = abi.READS
    else:
        seqs = workload.unpack(workload.make_protein_queries(db, 150, seed=7)) + [max(workload.unpack(db), key=len), b"AAAAAAA", b""]
        q = api.pack_sequences(seqs)
        queries = seqs
        seq_type = abi.PROTEIN
    exp = [_oracle_report(oracle, full, x, reads)[1] for x in queries]
    ix1 = api.Index.from_image(api.Image.from_proteins(packed=db), gpu_device)
    ref = ix1.search_top(packed=q, seq_type=seq_type)
    rpid, rkm = ref.dense()
    for world in (1, 2, 3, 8):
        imgs = [api.Image.from_proteins(packed=db, shard=r, n_shards=world) for r in range(world)]
        if world == 2:   # through files: kaamer_index_open_sharded
            with tempfile.TemporaryDirectory() as td:
                paths = []
                for r, im in enumerate(imgs):
                    paths.append("%s/shard%d.kgi" % (td, r))
                    im.save(paths[-1])
                sx = api.ShardedIndex.open(paths, [gpu_device] * world)
        else:
            sx = api.ShardedIndex.from_images(imgs, [gpu_device] * world)
        for rep in range(3):   # the second call reuses every buffer and sizes its exchange blocks from the first call's need
            if rep == 2:       # the two-halves form, struct entry point underneath the flat one
                top = sx.submit_top(packed=q, seq_type=seq_type).wait()
            else:
                top = sx.search_top(packed=q, seq_type=seq_type, flat=(rep == 0))
            info = sx.exchange_info()
            assert info["adaptive"] == (rep > 0) and info["queries"] == len(queries)
            if rep > 0:
                words = 3 if reads else 2
                payload = 4 * (8 + (len(queries) + world - 1) // world + words * info["need_entries"])
                assert info["block_bytes"] <= 1.5 * payload + 20000, (world, info)   # (+ the additive margins, which dominate on a batch this small)
            assert top.n_queries == ref.n_queries == len(queries)
            assert top.rep_query.tolist() == ref.rep_query.tolist(), world
            assert top.top_off.tolist() == ref.top_off.tolist()
            assert top.top_pid.tolist() == ref.top_pid.tolist() and top.top_kmatch.tolist() == ref.top_kmatch.tolist()
            assert top.trim.tolist() == ref.trim.tolist()
            for f in ("src_seq", "size_in_kmer", "start_position", "end_position", "plus_strand", "aa_len", "aa_off"):
                assert top.meta[f].tolist() == ref.meta[f].tolist(), (world, f)
            if reads:
                assert bytes(top.orf_aa) == bytes(ref.orf_aa)
                assert top.top_first_pos.tolist() == ref.top_first_pos.tolist()
            pid, km = top.dense()
            n = 0
            for i, e in enumerate(exp):
                k = int(top.top_cnt[i])
                assert list(zip(pid[i, :k].tolist(), km[i, :k].tolist())) == e, (world, i)
                n += k
            assert n > 100
            assert top.counters["n_lookup"] == ref.counters["n_lookup"] and top.counters["n_post"] == ref.counters["n_post"]
            assert top.counters["n_hits"] == ref.counters["n_hits"]
        sx.close()
    # a handle whose images are not shard i of n is refused
    with pytest.raises(abi.KaamerError) as e:
        api.ShardedIndex.from_images([api.Image.from_proteins(packed=db, shard=1, n_shards=2)] * 2, [gpu_device] * 2)
    assert e.value.code == abi.E_FORMAT


@pytest.mark.gpu
def test_sharded_handle_concurrent_callers(klib, oracle, gpu_device):
    """Several threads call kaamer_sharded_search_batch_top on ONE handle at once (the reference server's goroutines against
    one set of stores, api/server.go:47-65): every call takes a free set of per-shard workspaces and streams -- more
    threads than sets, protein and read batches mixed -- and every result equals the unsharded call's."""
    import threading
    from kaamer_amd import abi, api, workload
    db = workload.make_db(500, seed=16)
    ix1 = api.Index.from_image(api.Image.from_proteins(packed=db), gpu_device)
    world = 2
    sx = api.ShardedIndex.from_images([api.Image.from_proteins(packed=db, shard=r, n_shards=world) for r in range(world)], [gpu_device] * world)
    batches = [workload.make_protein_queries(db, 50 + 11 * i, seed=60 + i) for i in range(4)] + \
              [workload.make_reads(db, 100 + 20 * i, seed=80 + i) for i in range(2)]
    kinds = [abi.PROTEIN] * 4 + [abi.READS] * 2
    refs = [ix1.search_top(packed=q, seq_type=k) for q, k in zip(batches, kinds)]
    errors = []

    def same(top, ref):
        assert top.n_queries == ref.n_queries
        assert top.rep_query.tolist() == ref.rep_query.tolist()
        assert top.top_off.tolist() == ref.top_off.tolist()
        assert top.top_pid.tolist() == ref.top_pid.tolist() and top.top_kmatch.tolist() == ref.top_kmatch.tolist()
        assert top.trim.tolist() == ref.trim.tolist()

    def worker(i):
        try:
            for r in range(3):
                j = (i + 2 * r) % len(batches)
                same(sx.search_top(packed=batches[j], seq_type=kinds[j]), refs[j])
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    th = [threading.Thread(target=worker, args=(i,)) for i in range(6)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    sx.close()
    assert not errors, errors[:3]


@pytest.mark.gpu
@pytest.mark.parametrize("reads", [False, True])
def test_sharded_steps_in_flight(klib, oracle, gpu_device, reads):
    """ShardedPipeline: two searchers (workspaces, exchange buffers, streams of their own) take the batches in turn, so a
    batch's exchange overlaps the next batch's search.  Eight batches of two different inputs through a pipeline of depth
    two, at world 1 with the exchange code running (pack -> blocks -> merge): the reported hits of every batch equal the
    oracle's, whichever searcher took it, and the adaptive block layout settles on each searcher by itself."""
    import torch
    from kaamer_amd import abi, api, sharded, workload
    from oracle import oracle as O
    torch.cuda.set_device(gpu_device)
    db = workload.make_db(600, seed=6)
    ix = api.Index.from_image(api.Image.from_proteins(packed=db), gpu_device)
    full = O.Index.from_proteins(None, packed=db)
    inputs = []
    for seed in (12, 13):
        if reads:
            q = workload.make_reads(db, 250, seed=seed)
            queries = [o for r in workload.unpack(q) for o in O.get_orfs(r)]
        else:
            queries = workload.unpack(workload.make_protein_queries(db, 120 + seed, seed=seed)) + [b"AAAAAAA", b""]
            q = api.pack_sequences(queries)
        inputs.append((q, queries, torch.from_numpy(q[0]).cuda(), torch.from_numpy(q[1].view(np.int64)).cuda()))
    seq_type = abi.READS if reads else abi.PROTEIN
    max_bytes = max(len(i[0][0]) for i in inputs)
    max_seqs = max(len(i[0][1]) - 1 for i in inputs)
    pipe = sharded.ShardedPipeline(2, ix, 0, 1, max_bytes, max_seqs, seq_type=seq_type, max_entries_per_peer=1 << 17,
                                   transport="torch", first_pos=True)
    assert len(pipe) == 2 and not pipe.searchers[0].direct
    expected = [[_oracle_report(O, full, qq, reads)[1] for qq in queries] for _, queries, _, _ in inputs]

    def check(k, which):
        s_ = pipe.searchers[k]
        queries = inputs[which][1]
        t = s_.last_topn
        tc = sharded.dev_tensor(t.d_top_cnt, len(queries), torch.int32).cpu().numpy()
        tp = sharded.dev_tensor(t.d_top_pid, len(queries) * 10, torch.int32).cpu().numpy().view(np.uint32).reshape(-1, 10)
        tk = sharded.dev_tensor(t.d_top_kmatch, len(queries) * 10, torch.int32).cpu().numpy().reshape(-1, 10)
        for j in range(len(queries)):
            exp_rep = expected[which][j]
            assert int(tc[j]) == len(exp_rep), (k, which, j)
            assert list(zip(tp[j, :len(exp_rep)].tolist(), tk[j, :len(exp_rep)].tolist())) == exp_rep, (k, which, j)

    order = [0, 1, 1, 0, 0, 1, 0, 0]          # which input batch i carries: both searchers see both inputs
    for i in range(0, len(order), 2):
        took = []
        for w in order[i:i + 2]:
            q, queries, d_buf, d_off = inputs[w]
            k, _ = pipe.step(d_buf.data_ptr(), d_off.data_ptr(), len(q[1]) - 1, len(q[0]), topn={})
            took.append((k, w))
        assert [k for k, _ in took] == [0, 1]
        out = pipe.finish()
        assert all(o is not None for o in out)
        for k, w in took:
            check(k, w)
    for s_ in pipe.searchers:                # from its third batch on a searcher's blocks carry payload, not capacity
        assert int(s_.wire.e_cap) < int(s_.layout.e_cap)
    assert pipe.finish() == [None, None]     # nothing enqueued: nothing to finish
    pipe.close()


def _gpu_pipeline_worker(rank, world, port, ret):
    """GPU: ranks that share GPU 0, each driving a ShardedPipeline of depth 2 (host transport over gloo): six batches of two
    inputs; the reported hits of every batch equal those of a plain ShardedSearcher of the same rank on the same input."""
    for p_ in (ROOT, os.path.join(ROOT, "tests")):
        sys.path.insert(0, p_)
    import torch
    import torch.distributed as dist
    from kaamer_amd import abi, api, sharded, workload
    _init_gloo(rank, world, port)
    try:
        torch.cuda.set_device(0)
        db = workload.make_db(600, seed=6)
        ix = api.Index.from_image(api.Image.from_proteins(packed=db, shard=rank, n_shards=world), 0)
        inputs = []
        for seed in (21, 22):
            q = workload.make_protein_queries(db, 140 + seed, seed=seed)
            inputs.append((q, torch.from_numpy(q[0]).cuda(), torch.from_numpy(q[1].view(np.int64)).cuda()))
        max_bytes = max(len(i[0][0]) for i in inputs)
        max_seqs = max(len(i[0][1]) - 1 for i in inputs)
        kw = dict(seq_type=abi.PROTEIN, max_entries_per_peer=1 << 16, transport="host", first_pos=True)

        def reported(s_, n_q):
            owned = len(range(rank, n_q, world))
            t = s_.last_topn
            tc = sharded.dev_tensor(t.d_top_cnt, owned, torch.int32).cpu().numpy().copy()
            tp = sharded.dev_tensor(t.d_top_pid, owned * 10, torch.int32).cpu().numpy().reshape(-1, 10).copy()
            tk = sharded.dev_tensor(t.d_top_kmatch, owned * 10, torch.int32).cpu().numpy().reshape(-1, 10).copy()
            return [(int(tc[j]), tp[j, :tc[j]].tolist(), tk[j, :tc[j]].tolist()) for j in range(owned)]
        ref = sharded.ShardedSearcher(ix, rank, world, max_bytes, max_seqs, **kw)
        want = []
        for q, d_buf, d_off in inputs:
            ref.run(d_buf.data_ptr(), d_off.data_ptr(), len(q[1]) - 1, len(q[0]), torch.cuda.current_stream(), topn={})
            want.append(reported(ref, len(q[1]) - 1))
        ref.close()
        pipe = sharded.ShardedPipeline(2, ix, rank, world, max_bytes, max_seqs, **kw)
        n = 0
        for pair in ((0, 1), (1, 0), (0, 0)):
            took = []
            for w in pair:
                q, d_buf, d_off = inputs[w]
                k, _ = pipe.step(d_buf.data_ptr(), d_off.data_ptr(), len(q[1]) - 1, len(q[0]), topn={})
                took.append((k, w))
            pipe.finish()
            for k, w in took:
                got = reported(pipe.searchers[k], len(inputs[w][0][1]) - 1)
                assert got == want[w], (rank, k, w)
                n += sum(c for c, _, _ in got)
        pipe.close()
        ret[rank] = n
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_pipelined_sharded_steps_across_ranks(klib, oracle, gpu_device):
    """two ranks (processes on GPU 0), each with two sharded steps in flight: the collectives of the two searchers meet in
    the same order on both ranks (no hang), and every batch's reported hits are those of the one-step-at-a-time searcher"""
    ret = _spawn(_gpu_pipeline_worker, 2)
    assert len(ret) == 2 and all(v > 100 for v in ret.values()), ret
