"""GPU parity tests of the protein path: the HIP kernels (through the C ABI)
against the CPU oracle, bit-exact on {protein id -> Kmatch} per query."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _oracle_hits(oix, oracle, seqs):
    out = []
    for s in seqs:
        size = oracle.size_in_kmer(s)
        if size < 7:                       # search_protein.go:74-76
            out.append(({}, {}))
            continue
        pid, km, pos = oix.search(s, want_positions=True)
        first = {int(p): int(np.argmax(pos[i])) for i, p in enumerate(pid)}
        out.append((dict(zip(pid.tolist(), km.tolist())), first))
    return out


def _check(res, exp):
    assert res.n_queries == len(exp)
    for q, (hits, first) in enumerate(exp):
        assert res.hits(q) == hits, "query %d" % q
        assert res.first_pos(q) == first, "query %d first positions" % q


@pytest.fixture(scope="module")
def small(klib, oracle, gpu_device):
    """BASELINE config 1: DB-S (1000 proteins) + 100 protein queries."""
    from kaamer_amd import api, workload
    db = workload.make_db(1000)
    img = api.Image.from_proteins(packed=db)
    ix = api.Index.from_image(img, gpu_device)
    oix = oracle.Index.from_proteins(None, packed=db)
    return db, img, ix, oix


def test_config1_parity(small, oracle):
    from kaamer_amd import workload
    db, img, ix, oix = small
    q = workload.make_protein_queries(db, 100)
    res = ix.search(packed=q)
    exp = _oracle_hits(oix, oracle, workload.unpack(q))
    _check(res, exp)
    c = res.counters
    assert c["n_queries"] == 100
    assert c["n_lookup"] == sum(oracle.size_in_kmer(s) for s in workload.unpack(q))
    assert c["n_hits"] == sum(len(h) for h, _ in exp)
    assert c["n_probe"] >= c["n_lookup"] and c["n_found"] <= c["n_lookup"] and c["n_overflow"] <= 5
    # Σ_p Kmatch(p) = Σ_pos |index[key(pos)]| = n_post
    assert c["n_post"] == sum(sum(h.values()) for h, _ in exp)
    # meta: SizeInKmer / Location as GetQueriesFasta sets them (search.go:290-294)
    seqs = workload.unpack(q)
    assert res.meta["size_in_kmer"].tolist() == [oracle.size_in_kmer(s) for s in seqs]
    assert res.meta["end_position"].tolist() == [len(s) for s in seqs]
    assert (res.meta["start_position"] == 1).all() and (res.meta["plus_strand"] == 1).all()


def test_self_hits(small, oracle):
    """every DB protein queried against its own DB scores len-6 on itself (docs/client.md:120-156)"""
    from kaamer_amd import workload
    db, img, ix, oix = small
    seqs = workload.unpack(db)[:200]
    res = ix.search(seqs)
    for q, s in enumerate(seqs):
        assert res.hits(q)[q] == len(s) - 6
        assert res.first_pos(q)[q] == 0


def test_docs_worked_example(klib, oracle, gpu_device):
    from kaamer_amd import api
    ex = json.load(open(os.path.join(GOLD, "docs_example.json")))
    img = api.Image.from_proteins([ex["db_sequence"]], ids=[ex["hit_key"]])
    ix = api.Index.from_image(img, gpu_device)
    res = ix.search([ex["query"]])
    assert int(res.meta["size_in_kmer"][0]) == ex["size_in_kmer"]
    assert res.hits(0) == {ex["hit_key"]: ex["kmatch"]}
    assert (int(res.meta["start_position"][0]), int(res.meta["end_position"][0])) == (1, ex["end_position"])


def test_edge_cases(small, oracle):
    from kaamer_amd import workload
    db, img, ix, oix = small
    base = workload.unpack(db)
    seqs = [b"", b"A", b"AAAAAA", b"AAAAAAA", base[0][:12], base[0][:13], base[0][:13] + b"*",
            base[0][:14] + b"*", base[1] + b"*", base[2].lower(), b"X" * 40, b"*" * 30,
            base[3][:64 + 6], base[3][:65 + 6], base[3][:63 + 6], base[4][:128 + 6], base[4][:129 + 6],
            base[5][:20] + b"XX" + base[5][22:], bytes(range(256)) * 2, base[6] * 3,
            max(base, key=len), min(base, key=len)]
    res = ix.search(seqs)
    _check(res, _oracle_hits(oix, oracle, seqs))
    assert res.hits(0) == {} and res.hits(4) == {}          # SizeInKmer < 7 -> no result
    assert res.hits(5) != {}                                # 13 aa -> SizeInKmer 7


def test_empty_batch(small):
    db, img, ix, oix = small
    res = ix.search([])
    assert res.n_queries == 0 and len(res.hit_pid) == 0 and res.hit_off.tolist() == [0]


def test_ragged_many_queries(small, oracle):
    """more queries than resident waves, ragged lengths, every chunk boundary"""
    from kaamer_amd import workload
    db, img, ix, oix = small
    rng = np.random.default_rng(8)
    base = workload.unpack(db)
    seqs = []
    for i in range(3000):
        s = base[int(rng.integers(0, len(base)))]
        a = int(rng.integers(0, max(1, len(s) - 13)))
        n = int(rng.integers(0, 200))
        seqs.append(s[a:a + n])
    res = ix.search(seqs)
    _check(res, _oracle_hits(oix, oracle, seqs))


def test_shared_kmers_long_lists(klib, oracle, gpu_device):
    """postings lists longer than the in-lane limit take the wave-cooperative path"""
    from kaamer_amd import api
    rng = np.random.default_rng(3)
    alpha = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)
    core = bytes(alpha[rng.integers(0, 20, 40)])
    db = [bytes(alpha[rng.integers(0, 20, 10)]) + core[: 10 + (i % 31)] + bytes(alpha[rng.integers(0, 20, 10)])
          for i in range(300)]
    ids = (np.arange(300) * 7 + 11).astype(np.uint32)
    img = api.Image.from_proteins(db, ids=ids)
    assert img.stats()["max_list"] >= 200
    ix = api.Index.from_image(img, gpu_device)
    oix = oracle.Index.from_proteins(db, ids=ids)
    seqs = [core, db[0], db[150], core[5:30], b"AAAAAAAAAAAAAAAA"]
    res = ix.search(seqs)
    _check(res, _oracle_hits(oix, oracle, seqs))


def _device_search(ix, seqs, **ws_opts):
    """device-resident call with explicit workspace options -> (BatchResult-like dicts, counters)"""
    import torch
    from kaamer_amd import api
    buf, offs = api.pack_sequences(seqs)
    d_buf = torch.from_numpy(buf).cuda()
    d_off = torch.from_numpy(offs.view(np.int64)).cuda()
    ws_opts.setdefault("first_pos", 1)
    ws = api.Workspace(ix, len(buf), len(seqs), **ws_opts)
    st = torch.cuda.current_stream().cuda_stream
    r = ws.search_device(d_buf.data_ptr(), d_off.data_ptr(), len(seqs), len(buf), stream=st)
    c = ws.finish(st)
    hit_off = _from_ptr(r.d_hit_off, len(seqs) + 1, np.uint64)
    hit_cnt = _from_ptr(r.d_hit_cnt, len(seqs), np.uint32)
    n = int(r.hit_capacity)
    if ws_opts.get("compact"):  # CSR in query order
        assert hit_off[0] == 0 and (np.diff(hit_off.astype(np.int64)) == hit_cnt).all()
        assert int(hit_off[-1]) == c["n_hits"]
    pid, km, fp = (_from_ptr(x, n, np.uint32) for x in (r.d_hit_pid, r.d_hit_kmatch, r.d_hit_first_pos))
    sp = [(int(hit_off[i]), int(hit_off[i]) + int(hit_cnt[i])) for i in range(len(seqs))]
    hits = [dict(zip(pid[a:b].tolist(), km[a:b].tolist())) for a, b in sp]
    first = [dict(zip(pid[a:b].tolist(), fp[a:b].tolist())) for a, b in sp]
    return hits, first, c


def test_tier_escalation_is_exact(klib, oracle, gpu_device):
    """queries whose distinct hits exceed a counting tier move to the next tier (S: one wave,
    L: 16-wave workgroup with 4096 slots, G: exactly sized table in HBM) with identical results"""
    from kaamer_amd import abi, api
    rng = np.random.default_rng(4)
    alpha = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)
    core = bytes(alpha[rng.integers(0, 20, 40)])
    long_q = bytes(alpha[rng.integers(0, 20, 300)]) + core + bytes(alpha[rng.integers(0, 20, 400)])
    # 9000 proteins share the core with ragged ends -> 9000 distinct hits for a query holding the core: past the group
    # tables (4096 slots) and past the G tier's LDS table (6144 distinct hits), so the HBM table is what counts them
    db = [bytes(alpha[rng.integers(0, 20, 8)]) + core[(i % 5):] + bytes(alpha[rng.integers(0, 20, 8)]) for i in range(9000)]
    ids = (np.arange(9000, dtype=np.uint32) * 3 + 1)
    img = api.Image.from_proteins(db, ids=ids)
    ix = api.Index.from_image(img, gpu_device)
    oix = oracle.Index.from_proteins(db, ids=ids)
    seqs = [core, long_q, db[7], core[:20], b"ACDEFGHIKLMNPQRSTVWY"]
    exp = _oracle_hits(oix, oracle, seqs)
    assert len(exp[0][0]) == 9000
    for opts in (dict(), dict(compact=True), dict(g_tier_slots=1 << 20)):
        hits, first, c = _device_search(ix, seqs, **opts)
        for i, (h, f) in enumerate(exp):
            assert hits[i] == h and first[i] == f, (opts, i)
        assert c["n_overflow"] >= 2 and c["n_queries"] == len(seqs)
        assert c["n_lookup"] == sum(oracle.size_in_kmer(s) for s in seqs)
        assert c["n_post"] == sum(sum(h.values()) for h, _ in exp)
    # an arena too small for the G tier is an error, never a partial result
    with pytest.raises(abi.KaamerError) as e:
        _device_search(ix, seqs, g_tier_slots=4096)
    assert e.value.code == abi.E_CAPACITY
    # the host-buffer call takes the same path
    res = ix.search(seqs)
    _check(res, exp)


def test_device_resident_call_and_reuse(small, oracle):
    """torch owns the device buffers and the stream; the workspace is reused across batches"""
    import ctypes as C
    import torch
    from kaamer_amd import api, workload
    db, img, ix, oix = small
    stream = torch.cuda.Stream()
    for seed in (1, 2, 3):
        ws = api.Workspace(ix, 1 << 20, 500, compact=(seed == 2))
        ws.set_timing(1)
        q = workload.make_protein_queries(db, 300, seed=seed)
        buf, offs = q
        with torch.cuda.stream(stream):
            d_buf = torch.from_numpy(buf).cuda()
            d_off = torch.from_numpy(offs.view(np.int64)).cuda()
            r = ws.search_device(d_buf.data_ptr(), d_off.data_ptr(), 300, len(buf), stream=stream.cuda_stream)
            c = ws.finish(stream.cuda_stream)
        exp = _oracle_hits(oix, oracle, workload.unpack(q))
        n_hits = sum(len(h) for h, _ in exp)
        assert c["n_hits"] == n_hits

        # read results back through torch from the raw device pointers
        hit_off = _from_ptr(r.d_hit_off, 301, np.uint64)
        hit_cnt = _from_ptr(r.d_hit_cnt, 300, np.uint32)
        assert int(hit_cnt.sum()) == n_hits
        if seed == 2:
            assert int(hit_off[300]) == n_hits
        pid = _from_ptr(r.d_hit_pid, int(r.hit_capacity), np.uint32)
        km = _from_ptr(r.d_hit_kmatch, int(r.hit_capacity), np.uint32)
        for qi, (hits, _) in enumerate(exp):
            a, b = int(hit_off[qi]), int(hit_off[qi]) + int(hit_cnt[qi])
            assert dict(zip(pid[a:b].tolist(), km[a:b].tolist())) == hits
        t = ws.kernel_ms_sum()
        assert 0 < t["probe_ms"] and 0 < t["count_ms"] and t["probe_ms"] + t["count_ms"] <= t["total_ms"]


def _from_ptr(ptr, n, dtype):
    """copy n items of dtype from a raw device pointer (hipMemcpy through the HIP runtime torch loaded)"""
    import ctypes as C
    import torch
    out = np.empty(n, dtype=dtype)
    if n == 0:
        return out
    hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    torch.cuda.synchronize()
    rc = hip.hipMemcpy(out.ctypes.data, C.c_void_p(ptr), out.nbytes, 2)
    assert rc == 0
    return out


def test_position_bitmaps(small, oracle):
    """ExtractPositions: PositionHits[id][pos] (search.go:442-452) for every hit, bit-exact"""
    from kaamer_amd import workload
    db, img, ix, oix = small
    q = workload.make_protein_queries(db, 60, seed=9)
    seqs = workload.unpack(q) + [workload.unpack(db)[3], b"ACDEFGHIKLMNP", b"AAAA", max(workload.unpack(db), key=len)]
    res = ix.search(seqs, want_positions=True)
    exp = _oracle_hits(oix, oracle, seqs)
    _check(res, exp)
    n_bits = 0
    for qi, s in enumerate(seqs):
        size = oracle.size_in_kmer(s)
        got = res.positions(qi)
        if size < 7:
            assert got == {}
            continue
        pid, km, pos = oix.search(s, want_positions=True)
        assert sorted(got) == sorted(pid.tolist())
        for i, p in enumerate(pid.tolist()):
            assert got[p].tolist() == pos[i].tolist(), (qi, p)
            assert int(got[p].sum()) == int(km[i])
            n_bits += int(km[i])
    assert n_bits == res.counters["n_post"]


def test_position_bitmaps_g_tier(klib, oracle, gpu_device):
    """PositionHits of queries counted in the HBM tier (more distinct hits than an LDS table holds)"""
    from kaamer_amd import api
    rng = np.random.default_rng(14)
    alpha = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)
    core = bytes(alpha[rng.integers(0, 20, 70)])
    db = [bytes(alpha[rng.integers(0, 20, 6)]) + core[(i % 9):] + bytes(alpha[rng.integers(0, 20, 6)]) for i in range(3000)]
    ids = rng.permutation(20000)[:3000].astype(np.uint32)
    ix = api.Index.from_image(api.Image.from_proteins(db, ids=ids), gpu_device)
    oix = oracle.Index.from_proteins(db, ids=ids)
    seqs = [core, bytes(alpha[rng.integers(0, 20, 130)]) + core + bytes(alpha[rng.integers(0, 20, 70)]), db[11], core[:25]]
    res = ix.search(seqs, want_positions=True)
    assert res.counters["n_overflow"] >= 2
    for qi, s in enumerate(seqs):
        pid, km, pos = oix.search(s, want_positions=True)
        got = res.positions(qi)
        assert sorted(got) == sorted(pid.tolist())
        if qi < 2:
            assert len(pid) == 3000
        for i, p in enumerate(pid.tolist()):
            assert got[p].tolist() == pos[i].tolist(), (qi, p)


def test_counting_tables_follow_the_previous_batch(klib, oracle, gpu_device):
    """A database whose proteins are built from a small pool of segments: a query meets more proteins than it has k-mers,
    so tables of 1.5 x SizeInKmer slots crowd and queries leave them for the G tier.  The finalize step of a batch leaves
    the next batch's table scale on the device (hits per k-mer x 1.9, within what max_hits provisioned): the second run
    of the same batch sends fewer queries there, and the results are the oracle's every time."""
    import torch
    from kaamer_amd import api
    rng = np.random.default_rng(12)
    alpha = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)
    pool = [bytes(alpha[rng.integers(0, 20, 25)]) for _ in range(40)]
    db = [b"".join(pool[int(k)] for k in rng.integers(0, 40, 10)) for _ in range(1500)]
    img = api.Image.from_proteins(db)
    ix = api.Index.from_image(img, gpu_device)
    oix = oracle.Index.from_proteins(db)
    seqs = [db[int(i)] for i in rng.integers(0, len(db), 120)]
    exp = _oracle_hits(oix, oracle, seqs)
    sizes = sum(oracle.size_in_kmer(s) for s in seqs)
    assert sum(len(h) for h, _ in exp) > 2 * sizes          # more than two distinct proteins per k-mer
    buf, offs = api.pack_sequences(seqs)
    d_buf = torch.from_numpy(buf).cuda()
    d_off = torch.from_numpy(offs.view(np.int64)).cuda()
    st = torch.cuda.current_stream().cuda_stream

    def run(ws):
        r = ws.search_device(d_buf.data_ptr(), d_off.data_ptr(), len(seqs), len(buf), stream=st)
        c = ws.finish(st)
        hit_off = _from_ptr(r.d_hit_off, len(seqs) + 1, np.uint64)
        hit_cnt = _from_ptr(r.d_hit_cnt, len(seqs), np.uint32)
        n = int(r.hit_capacity)
        pid, km, fp = (_from_ptr(x, n, np.uint32) for x in (r.d_hit_pid, r.d_hit_kmatch, r.d_hit_first_pos))
        for i, (h, f) in enumerate(exp):
            a, b = int(hit_off[i]), int(hit_off[i]) + int(hit_cnt[i])
            assert dict(zip(pid[a:b].tolist(), km[a:b].tolist())) == h, i
            assert dict(zip(pid[a:b].tolist(), fp[a:b].tolist())) == f, i
        return c["n_overflow"]

    roomy = api.Workspace(ix, len(buf), len(seqs), first_pos=1, max_hits=40 * len(buf))
    ovf = [run(roomy) for _ in range(3)]
    assert ovf[0] > 0 and ovf[1] < ovf[0] and ovf[2] <= ovf[1], ovf
    # a workspace with default sizes: whatever scale its hit arrays allow, the results are the same
    plain = api.Workspace(ix, len(buf), len(seqs), first_pos=1)
    ovf_p = [run(plain) for _ in range(2)]
    assert ovf_p[0] == ovf[0] and ovf_p[1] <= ovf_p[0]


@pytest.mark.parametrize("first_pos", [1, -1])
def test_both_counting_kernels(small, klib, oracle, gpu_device, first_pos, monkeypatch):
    """Protein batches are counted by count_group_kernel (units of two group windows per barrier cycle; four with a larger
    arena behind an experiment knob) or, with
    KAAMER_COUNT_ASYNC=1, by the barrier-free kernel (count_async.hip.inc: window, stripe and build jobs from LDS counters,
    two groups alive per workgroup): same hit lists, first positions and counters from both,
    equal to the oracle's -- on a ragged batch of many groups per workgroup, with empty and too-short queries, queries
    that leave their table for the G tier, and tables as large as the arena allows."""
    from kaamer_amd import api, workload
    db, img, ix, oix = small
    rng = np.random.default_rng(11)
    recs = workload.unpack(db)
    seqs = workload.unpack(workload.make_protein_queries(db, 700, seed=5))
    seqs += [b"", b"ACDEFG", recs[3][:7], recs[5] * 9, b"".join(recs[int(i)] for i in rng.integers(0, len(recs), 12))]
    seqs += [recs[int(i)][: int(n)] for i, n in zip(rng.integers(0, len(recs), 300), rng.integers(1, 60, 300))]
    seqs += workload.unpack(workload.make_protein_queries(db, 500, seed=6))
    exp = _oracle_hits(oix, oracle, seqs)
    ref_c = None
    # alone: three workgroups per CU; next to other batches: one, with units of two or (KAAMER_UNIT_WINDOWS=4, an experiment
    # knob) four windows; the barrier-free kernel only when told to
    for cb, asyn, uw in ((0, "0", "2"), (3, "0", "2"), (3, "0", "4"), (3, "1", "2")):
        monkeypatch.setenv("KAAMER_COUNT_ASYNC", asyn)
        monkeypatch.setenv("KAAMER_UNIT_WINDOWS", uw)
        hits, first, c = _device_search(ix, seqs, first_pos=first_pos, concurrent_batches=cb)
        for i, (h, f) in enumerate(exp):
            assert hits[i] == h, (cb, i)
            if first_pos == 1:
                assert first[i] == f, (cb, i)
        assert c["n_queries"] == sum(oracle.size_in_kmer(s) >= 7 for s in seqs) and c["n_hits"] == sum(len(h) for h, _ in exp)
        assert c["n_post"] == sum(sum(h.values()) for h, _ in exp)
        if ref_c is None:
            ref_c = c
        else:
            assert {k: v for k, v in c.items() if k != "n_overflow"} == {k: v for k, v in ref_c.items() if k != "n_overflow"}
    # a skewed family: 9 000 proteins behind one motif (tables that fill up, the G tier's three stages) through both kernels
    alpha = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)
    core = bytes(alpha[rng.integers(0, 20, 40)])
    fam = [bytes(alpha[rng.integers(0, 20, 8)]) + core[(i % 5):] + bytes(alpha[rng.integers(0, 20, 8)]) for i in range(9000)]
    fix = api.Index.from_image(api.Image.from_proteins(fam), gpu_device)
    foix = oracle.Index.from_proteins(fam)
    fq = [core, bytes(alpha[rng.integers(0, 20, 300)]) + core + bytes(alpha[rng.integers(0, 20, 400)]), fam[7], core[:20]] * 6
    fq += [bytes(alpha[rng.integers(0, 20, int(n))]) for n in rng.integers(20, 900, 200)]
    fexp = _oracle_hits(foix, oracle, fq)
    for cb, asyn, uw in ((0, "0", "2"), (3, "0", "4"), (3, "1", "2")):
        monkeypatch.setenv("KAAMER_COUNT_ASYNC", asyn)
        monkeypatch.setenv("KAAMER_UNIT_WINDOWS", uw)
        hits, first, c = _device_search(fix, fq, first_pos=first_pos, concurrent_batches=cb)
        for i, (h, f) in enumerate(fexp):
            assert hits[i] == h, (cb, i)
            if first_pos == 1:
                assert first[i] == f, (cb, i)
        assert c["n_overflow"] >= 12


@pytest.mark.parametrize("first_pos", [1, -1])
def test_units_that_do_not_fit_the_arena(small, oracle, first_pos):
    """count_group_kernel counts a UNIT of two group windows per barrier cycle when the unit's tables fit its arena (sized
    for one window's worst case) and in two cycles when they do not.  Runs of small queries followed by a query whose
    table has the largest size (4 096 slots), repeated at shifting offsets of the layout, make units of up to 8 000 slots:
    both forms, the second part's own descriptor loads, and hit lists that start where the layout says."""
    from kaamer_amd import workload
    db, img, ix, oix = small
    rng = np.random.default_rng(17)
    recs = workload.unpack(db)
    big = [b"".join(recs[int(i)] for i in rng.integers(0, len(recs), 9))[:3100] for _ in range(6)]   # 3 094 k-mers: a 4 096-slot table
    seqs = []
    for rep in range(36):
        n_small = 6 + rep % 9
        for _ in range(n_small):
            r = recs[int(rng.integers(0, len(recs)))]
            seqs.append(r[: int(rng.integers(60, 320))])
        seqs.append(big[rep % len(big)])
        if rep % 4 == 0:
            seqs.append(big[(rep + 1) % len(big)])        # two largest tables in a row
    exp = _oracle_hits(oix, oracle, seqs)
    for cb in (0, 3):
        hits, first, c = _device_search(ix, seqs, first_pos=first_pos, concurrent_batches=cb)
        for i, (h, f) in enumerate(exp):
            assert hits[i] == h, (cb, i)
            if first_pos == 1:
                assert first[i] == f, (cb, i)
        assert c["n_hits"] == sum(len(h) for h, _ in exp) and c["n_overflow"] <= 4
