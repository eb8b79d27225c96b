"""Differential fuzz: many small random batches through ONE pair of reused workspaces (protein and
reads), every result compared with the oracle.  Catches state leaking between batches (the kernels
leave all per-batch device state clean themselves), ticket/scheduling races and boundary cases the
fixed tests do not enumerate: random batch sizes, query lengths 0..3000, reads 0..700 nt, queries
drawn from the DB (many hits) or random (few), unknown residues, '*' ends."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _from_ptr(ptr, n, dtype):
    from test_gpu_protein import _from_ptr as f
    return f(ptr, n, dtype)


def test_fuzz_reused_workspaces(klib, oracle, gpu_device):
    import torch
    from kaamer_amd import abi, api, workload
    rng = np.random.default_rng(20261003)
    db = workload.make_db(3000, seed=5)
    prots = workload.unpack(db)
    ix = api.Index.from_image(api.Image.from_proteins(packed=db), gpu_device)
    oix = oracle.Index.from_proteins(None, packed=db)
    st = torch.cuda.current_stream().cuda_stream
    pws = api.Workspace(ix, 1 << 20, 600, first_pos=1)
    rws = api.Workspace(ix, 1 << 19, 600, seq_type=abi.READS)
    aa = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWYXBZU*", dtype=np.uint8)
    nt = np.frombuffer(b"ACGTacgtNn", dtype=np.uint8)

    def rand_protein():
        k = rng.integers(0, 10)
        if k < 5:   # a DB member, mutated, maybe truncated
            p = bytearray(prots[int(rng.integers(0, len(prots)))])
            for _ in range(int(rng.integers(0, 6))):
                p[int(rng.integers(0, len(p)))] = int(aa[rng.integers(0, 20)])
            a = int(rng.integers(0, max(1, len(p) // 2)))
            return bytes(p[a:a + int(rng.integers(0, len(p) - a + 1))])
        if k < 8:   # random
            return bytes(aa[rng.integers(0, 25 if k == 7 else 20, int(rng.integers(0, 400)))])
        return bytes(aa[rng.integers(0, 20, int(rng.integers(0, 3000)))])

    def rand_read():
        k = rng.integers(0, 10)
        n = int(rng.integers(0, 700 if k == 0 else 260))
        if k < 6 and n >= 30:   # back-translated window of a DB protein (hits), random strand via make_reads-like path
            r = workload.unpack(workload.make_reads(db, 1, read_len=max(30, n), seed=int(rng.integers(1 << 30))))[0]
            return r
        return bytes(nt[rng.integers(0, 10 if k == 9 else 4, n)])

    n_hits = 0
    for it in range(120):
        if it % 2 == 0:
            seqs = [rand_protein() for _ in range(int(rng.integers(1, 120)))]
            buf, offs = api.pack_sequences(seqs)
            d_buf = torch.from_numpy(buf if len(buf) else np.zeros(1, np.uint8)).cuda()
            d_off = torch.from_numpy(offs.view(np.int64)).cuda()
            r = pws.search_device(d_buf.data_ptr(), d_off.data_ptr(), len(seqs), len(buf), stream=st)
            c = pws.finish(st)
            off = _from_ptr(r.d_hit_off, len(seqs), np.uint64)
            cnt = _from_ptr(r.d_hit_cnt, len(seqs), np.uint32)
            pid = _from_ptr(r.d_hit_pid, int(r.hit_capacity), np.uint32)
            km = _from_ptr(r.d_hit_kmatch, int(r.hit_capacity), np.uint32)
            fp = _from_ptr(r.d_hit_first_pos, int(r.hit_capacity), np.uint32)
            for i, s in enumerate(seqs):
                size = oracle.size_in_kmer(s)
                exp = {}
                if size >= 7:
                    e_pid, e_km, e_pos = oix.search(s, size=size, want_positions=True)
                    exp = {int(p): (int(k), int(np.argmax(e_pos[j]))) for j, (p, k) in enumerate(zip(e_pid, e_km))}
                a, b = int(off[i]), int(off[i]) + int(cnt[i])
                got = {int(p): (int(k), int(f)) for p, k, f in zip(pid[a:b], km[a:b], fp[a:b])}
                assert got == exp, (it, i)
                n_hits += len(exp)
        else:
            reads = [rand_read() for _ in range(int(rng.integers(1, 100)))]
            buf, offs = api.pack_sequences(reads)
            d_buf = torch.from_numpy(buf if len(buf) else np.zeros(1, np.uint8)).cuda()
            d_off = torch.from_numpy(offs.view(np.int64)).cuda()
            r = rws.search_device(d_buf.data_ptr(), d_off.data_ptr(), len(reads), len(buf), stream=st)
            c = rws.finish(st)
            nq = c["n_queries"]
            off = _from_ptr(r.d_hit_off, max(nq, 1), np.uint64)
            cnt = _from_ptr(r.d_hit_cnt, max(nq, 1), np.uint32)
            pid = _from_ptr(r.d_hit_pid, int(r.hit_capacity), np.uint32)
            km = _from_ptr(r.d_hit_kmatch, int(r.hit_capacity), np.uint32)
            qi = 0
            for rd in reads:
                for o in oracle.get_orfs(rd):
                    e_pid, e_km, _ = oix.search(o["seq"])
                    a, b = int(off[qi]), int(off[qi]) + int(cnt[qi])
                    assert dict(zip(pid[a:b].tolist(), km[a:b].tolist())) == dict(zip(e_pid.tolist(), e_km.tolist())), (it, qi)
                    n_hits += len(e_pid)
                    qi += 1
            assert qi == nq
    assert n_hits > 8000
