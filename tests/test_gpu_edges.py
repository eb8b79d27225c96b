"""Edge cases of the hot path against the CPU oracle, through the C ABI:
  * a query of more than 65 535 k-mers with first positions (16-bit packed counts/positions of the LDS
    tables do not hold it: the G tier must take it);
  * postings lists of more than 10 000 proteins (the very-long-list expansion, tables that overflow);
  * '.' in the database and in the queries (aaTable[{a,'.'}], k_store.go:48-52,102-103);
  * every byte value below 0x80 in nucleotide input (strings.ToLower + map misses, dna.go:68,106)."""
import numpy as np
import pytest

from kaamer_amd import abi

pytestmark = pytest.mark.gpu


def _expect(oracle, oix, s, positions=True):
    size = oracle.size_in_kmer(s)
    if size < 7:
        return {}, {}
    pid, km, pos = oix.search(s, want_positions=positions)
    fp = {int(p): int(np.argmax(pos[j])) for j, p in enumerate(pid)} if positions else {}
    return dict(zip(pid.tolist(), km.tolist())), fp


def test_query_longer_than_65535_kmers(klib, oracle, gpu_device):
    from kaamer_amd import api, workload
    db = workload.make_db(3000, seed=21)
    ix = api.Index.from_image(api.Image.from_proteins(packed=db), gpu_device)
    oix = oracle.Index.from_proteins(None, packed=db)
    recs = workload.unpack(db)
    rng = np.random.default_rng(5)
    long_q = b"".join(recs[int(i)] for i in rng.integers(0, len(recs), 260))   # ~90 000 residues, hits everywhere
    assert len(long_q) - 6 > 70000
    qs = [recs[3], long_q, recs[17][:40], long_q[1000:70000]]
    res = ix.search(qs)                   # first positions on (PACKED tables): the long queries leave the LDS tier
    assert res.counters["n_overflow"] >= 2
    for i, s in enumerate(qs):
        exp, fp = _expect(oracle, oix, s)
        assert res.hits(i) == exp, "query %d" % i
        assert res.first_pos(i) == fp, "query %d" % i
    assert max(res.hits(1).values()) < 65535 and len(res.hits(1)) > 200
    # and full PositionHits bitmaps for the same batch
    resp = ix.search(qs, want_positions=True)
    for i, s in enumerate(qs):
        size = oracle.size_in_kmer(s)
        pid, km, pos = oix.search(s, want_positions=True)
        got = resp.positions(i)
        assert set(got) == set(pid.tolist())
        for j in range(0, len(pid), max(1, len(pid) // 40)):
            assert (got[int(pid[j])] == pos[j]).all(), (i, int(pid[j]))


def test_postings_lists_of_ten_thousand(klib, oracle, gpu_device):
    """12 000 proteins share one 20-residue motif: its 14 k-mers have postings lists of 12 000 ids."""
    from kaamer_amd import api, workload
    rng = np.random.default_rng(9)
    aa = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)
    motif = bytes(aa[rng.integers(0, 20, 20)])
    motif2 = bytes(aa[rng.integers(0, 20, 12)])
    seqs = []
    for i in range(12000):
        body = bytearray(bytes(aa[rng.integers(0, 20, 60)]))
        body[20:40] = motif
        if i % 3 == 0:
            body[44:56] = motif2              # a second family of 4 000
        seqs.append(bytes(body))
    img = api.Image.from_proteins(seqs)
    assert img.stats()["max_list"] == 12000
    ix = api.Index.from_image(img, gpu_device)
    oix = oracle.Index.from_proteins(seqs)
    fill = lambda n: bytes(aa[rng.integers(0, 20, n)])
    qs = [fill(30) + motif + fill(30),                    # 14 positions x 12 000 ids: > 4096 distinct hits -> G tier
          seqs[7], seqs[9], motif + motif2 + motif,       # several long lists in one window
          fill(200), motif[:13] + fill(5) + motif2,
          (fill(10) + motif) * 20]                        # 280 positions x 12 000 ids = 3.4 M postings for one query
    res = ix.search(qs)
    assert res.counters["n_overflow"] >= 3 and res.counters["n_post"] > 3_000_000
    for i, s in enumerate(qs):
        exp, fp = _expect(oracle, oix, s)
        assert res.hits(i) == exp, "query %d" % i
        assert res.first_pos(i) == fp, "query %d" % i
    top = ix.search_top(qs)                               # 12 000 hits with massive ties through the device top-N
    tp, tk = top.dense()
    for i, s in enumerate(qs):
        size = oracle.size_in_kmer(s)
        pid, km, _ = oix.search(s)
        keep = oracle.filter_results(km, size) if len(km) else 0
        assert int(top.top_cnt[i]) == keep
        assert tp[i, :keep].tolist() == pid[:keep].tolist() and tk[i, :keep].tolist() == km[:keep].tolist()


def test_dot_in_database_and_queries(klib, oracle, gpu_device):
    from kaamer_amd import api
    rng = np.random.default_rng(13)
    alpha = np.frombuffer(b"ACDEFGHIKLMNPQRSTUVWY...XB*", dtype=np.uint8)   # '.' over-represented, a few non-letters
    seqs = [bytes(alpha[rng.integers(0, len(alpha), int(n))]) for n in rng.integers(20, 200, 400)]
    seqs += [b"C.AAAAA" * 4, b"Y.Y.Y.Y.Y.Y.Y", b"......." * 3, b"AAAAAA.AAAAAA."]
    ix = api.Index.from_image(api.Image.from_proteins(seqs), gpu_device)
    oix = oracle.Index.from_proteins(seqs)
    assert abi.lib().kaamer_encode_kmer(b"C.AAAAA") == 0x008582C0 == oracle.encode_kmer("C.AAAAA")
    qs = [seqs[i] for i in range(0, len(seqs), 7)] + [b"C.AAAAA", b"Y.Y.Y.Y", b"CAAAAAAAA", b"A.A.A.A.A.A.A.A"]
    qs += [bytes(alpha[rng.integers(0, len(alpha), 80)]) for _ in range(40)]
    res = ix.search(qs)
    n = 0
    for i, s in enumerate(qs):
        exp, fp = _expect(oracle, oix, s)
        assert res.hits(i) == exp, "query %d %r" % (i, s)
        assert res.first_pos(i) == fp
        n += len(exp)
    assert n > 200


def test_reads_with_every_byte_value(klib, oracle, gpu_device):
    """Nucleotide input with every byte below 0x80 in it (upper / lower case, N, digits, punctuation): ORFs as the
    reference finds them (ToLower, then any codon with a byte other than a/c/g/t is a map miss)."""
    from kaamer_amd import api, workload
    from test_gpu_reads import _check_reads
    db = workload.make_db(500, seed=3)
    ix = api.Index.from_image(api.Image.from_proteins(packed=db), gpu_device)
    oix = oracle.Index.from_proteins(None, packed=db)
    base = workload.unpack(workload.make_reads(db, 256, seed=31))
    rng = np.random.default_rng(2)
    reads = []
    for i, r in enumerate(base):
        b = bytearray(r.lower() if i % 2 else r)
        for _ in range(3):                                    # three foreign bytes per read, all 128 values covered
            b[int(rng.integers(0, len(b)))] = (i // 2 + int(rng.integers(0, 128))) % 128
        b[int(rng.integers(0, len(b)))] = i % 128
        reads.append(bytes(b))
    assert len({c for r in reads for c in r}) == 128
    res = ix.search(reads, seq_type=abi.READS)
    assert _check_reads(res, reads, oracle, oix, check_hits=True) > 300


@pytest.mark.parametrize("seq_type", ["protein", "reads"])
def test_device_offsets_that_disagree_with_the_batch_size(klib, oracle, gpu_device, seq_type):
    """The offsets array lives in device memory and sizes every later access.  Offsets that end beyond the byte count
    the call was given (or run backwards) are an error reported by kaamer_workspace_finish -- nothing is read past
    the buffers, and the workspace is usable afterwards."""
    import torch
    from kaamer_amd import api, workload
    db = workload.make_db(300, seed=5)
    ix = api.Index.from_image(api.Image.from_proteins(packed=db), gpu_device)
    oix = oracle.Index.from_proteins(None, packed=db)
    if seq_type == "protein":
        seqs, st = workload.unpack(workload.make_protein_queries(db, 40, seed=6)), abi.PROTEIN
    else:
        seqs, st = workload.unpack(workload.make_reads(db, 200, seed=6)), abi.READS
    buf, offs = api.pack_sequences(seqs)
    d_buf = torch.from_numpy(buf).cuda()
    ws = api.Workspace(ix, len(buf), len(seqs), seq_type=st)
    stream = torch.cuda.current_stream().cuda_stream
    bad_end = offs.copy(); bad_end[-1] += 4096            # the last sequence claims bytes the batch does not have
    bad_mid = offs.copy(); bad_mid[5] = bad_mid[7] + 3    # not ascending
    huge = offs.copy(); huge[3:] += np.uint64(1 << 40)
    # overlapping ranges: every other "sequence" is 2 000 bytes long and starts 10 bytes after the previous one -- far more
    # long sequences (and residues) than a batch of this size can hold
    over = offs.copy()
    over[0::2] = np.arange(len(over[0::2]), dtype=np.uint64) * 10
    over[1::2] = over[0::2][:len(over[1::2])] + np.uint64(min(2000, len(buf) - 10 * len(over)))
    over = np.minimum(over, np.uint64(len(buf)))
    for bad in (bad_end, bad_mid, huge, over):
        d_off = torch.from_numpy(bad.view(np.int64)).cuda()
        ws.search_device(d_buf.data_ptr(), d_off.data_ptr(), len(seqs), len(buf), stream=stream)
        with pytest.raises(abi.KaamerError) as e:
            ws.finish(stream)
        assert e.value.code == abi.E_CAPACITY
    # and the same workspace still gives the right answer for the consistent batch
    d_off = torch.from_numpy(offs.view(np.int64)).cuda()
    r = ws.search_device(d_buf.data_ptr(), d_off.data_ptr(), len(seqs), len(buf), stream=stream)
    c = ws.finish(stream)
    if seq_type == "protein":
        assert c["n_queries"] == sum(1 for s in seqs if oracle.size_in_kmer(s) >= 7)
        assert c["n_lookup"] == sum(oracle.size_in_kmer(s) for s in seqs if oracle.size_in_kmer(s) >= 7)
    else:
        assert c["n_queries"] == sum(len(oracle.get_orfs(s)) for s in seqs)


def test_zipf_database_parity(klib, oracle, gpu_device):
    """a skewed database (motifs drawn from a power law: postings lists of thousands of proteins): most queries leave
    the LDS counting tables, half of them have more distinct hits than the G tier's LDS table holds (up to 30 000) and
    are counted in several passes over ranges of protein ids; every {protein id -> Kmatch} and first position vs the
    oracle, through the full-list call and (reported hits) the top-N call"""
    from kaamer_amd import abi, api, workload
    db = workload.make_db_zipf(40000, seed=11, n_motifs=1500, zipf_a=1.0, per_residues=60)
    ix = api.Index.from_image(api.Image.from_proteins(packed=db), gpu_device)
    oix = oracle.Index.from_proteins(None, packed=db)
    q = workload.make_protein_queries(db, 200, seed=12)
    res = ix.search(packed=q)
    top = ix.search_top(packed=q)
    tp, tk = top.dense()
    n_big = 0
    for i, s in enumerate(workload.unpack(q)):
        size = oracle.size_in_kmer(s)
        exp, fp = {}, {}
        keep = 0
        if size >= 7:
            pid, km, pos = oix.search(s, want_positions=True)
            exp = dict(zip(pid.tolist(), km.tolist()))
            fp = {int(p): int(np.argmax(pos[j])) for j, p in enumerate(pid)}
            keep = oracle.filter_results(km, size) if len(km) else 0
            assert tp[i, :keep].tolist() == pid[:keep].tolist() and tk[i, :keep].tolist() == km[:keep].tolist(), i
        assert int(top.top_cnt[i]) == keep, i
        assert res.hits(i) == exp, "query %d" % i
        assert res.first_pos(i) == fp, "query %d" % i
        n_big += len(exp) > 6144
    assert n_big >= 50 and res.counters["n_overflow"] >= 100
    assert res.counters["n_post"] == sum(int(v) for i in range(res.n_queries) for v in res.hits(i).values())
