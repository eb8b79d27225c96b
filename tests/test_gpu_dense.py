"""A dense database: every k-mer of a query meets several proteins, a query has more hits than k-mers, and the counting
tables of the NEXT batch scale with what this batch found (finalize_body -> fit_table_scale, count_group.hip.inc).  The
results are the oracle's whatever the scale: at scale 1 the queries leave their LDS tables for the G tier, at the
adapted scale they stay; a workspace with little room for tables clamps the scale to what the batch's own positions
leave (and a batch that does not fit at any scale still reports the capacity error, never wrong lists)."""
import numpy as np
import pytest

from kaamer_amd import abi

pytestmark = pytest.mark.gpu

ALPHA = np.frombuffer(b"ACDEFG", dtype=np.uint8)          # 6^7 = 279 936 k-mers
CODON = {65: b"GCT", 67: b"TGT", 68: b"GAT", 69: b"GAA", 70: b"TTT", 71: b"GGT"}


@pytest.fixture(scope="module")
def dense(klib, oracle, gpu_device):
    from kaamer_amd import api
    rng = np.random.default_rng(77)
    seqs = [bytes(ALPHA[rng.integers(0, 6, 100)]) for _ in range(9000)]   # 846 000 k-mer instances: ~3 proteins per k-mer
    ix = api.Index.from_image(api.Image.from_proteins(seqs), gpu_device)
    oix = oracle.Index.from_proteins(seqs)
    return seqs, ix, oix, rng


def _check(res, qs, oix, oracle, step=1):
    n_hits = 0
    for i in range(0, len(qs), step):
        pid, km, pos = oix.search(qs[i], want_positions=True)
        assert res.hits(i) == dict(zip(pid.tolist(), km.tolist())), "query %d" % i
        assert res.first_pos(i) == {int(p): int(np.argmax(pos[j])) for j, p in enumerate(pid)}, "query %d" % i
        n_hits += len(pid)
    return n_hits


def test_protein_tables_follow_the_previous_batch(dense, oracle):
    seqs, ix, oix, rng = dense
    qs = [bytes(ALPHA[rng.integers(0, 6, int(n))]) for n in rng.integers(60, 140, 600)]
    first = ix.search(qs)                       # a fresh workspace: scale 1, tables of 1.5 x SizeInKmer
    c1 = dict(first.counters)
    assert c1["n_hits"] > 2 * c1["n_lookup"]    # more than two distinct proteins per k-mer
    assert c1["n_overflow"] > len(qs) // 2
    n = _check(first, qs, oix, oracle, step=3)
    assert n > 10000
    second = ix.search(qs)                      # the same workspace: tables of 1.5 x SizeInKmer x (hits per k-mer) x 1.9
    c2 = dict(second.counters)
    assert c2["n_overflow"] * 8 < c1["n_overflow"], (c1, c2)
    assert c2["n_hits"] == c1["n_hits"] and c2["n_post"] == c1["n_post"]
    _check(second, qs, oix, oracle, step=2)
    top1 = ix.search_top(qs)
    tp, tk = top1.dense()
    for i in range(0, len(qs), 5):
        pid, km, _ = oix.search(qs[i])
        keep = oracle.filter_results(km, oracle.size_in_kmer(qs[i])) if len(km) else 0
        assert int(top1.top_cnt[i]) == keep
        assert tp[i, :keep].tolist() == pid[:keep].tolist() and tk[i, :keep].tolist() == km[:keep].tolist()


def test_orf_tables_follow_the_previous_batch(dense, oracle, gpu_device, monkeypatch):
    """reads whose forward frame back-translates database-like protein: ORFs of 100 residues with ~280 hits each.  The
    other five frames are ORFs without hits (no stop codon among these codons), so the BATCH's mean is 0.5 hits per
    k-mer and the shipped margin of 1.9 leaves the scale at 1: the test widens the margin (KAAMER_SLOT_MARGIN, read
    when a workspace is created) to see the ORF prep kernel apply a scale"""
    from test_gpu_reads import _check_reads
    from kaamer_amd import api
    seqs, _, oix, rng = dense
    monkeypatch.setenv("KAAMER_SLOT_MARGIN", "8")
    ix = api.Index.from_image(api.Image.from_proteins(seqs), gpu_device)
    reads = []
    for _ in range(400):
        aa = ALPHA[rng.integers(0, 6, 100)]
        reads.append(b"".join(CODON[int(a)] for a in aa))
    first = ix.search(reads, seq_type=abi.READS)
    c1 = dict(first.counters)
    assert c1["n_hits"] * 5 > 2 * c1["n_lookup"] and c1["n_overflow"] > 200
    second = ix.search(reads, seq_type=abi.READS)
    c2 = dict(second.counters)
    assert c2["n_overflow"] * 8 < c1["n_overflow"], (c1, c2)
    assert c2["n_hits"] == c1["n_hits"]
    _check_reads(second, reads, oracle, oix, check_hits=False)       # every ORF of every read
    for res, step in ((second, 3), (first, 11)):                     # hit lists and first positions of a sample
        n = 0
        for q in range(0, res.n_queries, step):
            m = res.meta[q]
            aa = bytes(res.orf_aa[int(m["aa_off"]):int(m["aa_off"]) + int(m["aa_len"])]).decode("latin-1")
            pid, km, pos = oix.search(aa, want_positions=True)
            assert res.hits(q) == dict(zip(pid.tolist(), km.tolist())), "orf %d" % q
            assert res.first_pos(q) == {int(p): int(np.argmax(pos[i])) for i, p in enumerate(pid)}
            n += len(pid)
        assert n > 4000
    ix.close()


def test_scale_clamped_by_the_room_of_the_hit_arrays(dense, oracle):
    """a device workspace with hit arrays just large enough for the batch at scale ~2: the scale the previous batch asked
    for (5-6) is cut to what THIS batch's positions leave, the lists are still the oracle's and nothing is a capacity error"""
    import torch
    from test_gpu_protein import _from_ptr
    from kaamer_amd import api, workload
    seqs, ix, oix, rng = dense
    qs = [bytes(ALPHA[rng.integers(0, 6, 100)]) for _ in range(2000)]
    buf, offs = api.pack_sequences(qs)
    d_buf = torch.from_numpy(np.ascontiguousarray(buf)).cuda()
    d_off = torch.from_numpy(np.ascontiguousarray(offs).view(np.int64)).cuda()
    st = torch.cuda.current_stream().cuda_stream
    overflow = []
    for max_hits in (200_000, 40_000_000):         # room for tables at ~2.7 x, and for everything the batch asks
        ws = api.Workspace(ix, len(buf), len(qs), max_hits=max_hits, g_tier_slots=4 << 20)
        for rep in range(3):
            r = ws.search_device(d_buf.data_ptr(), d_off.data_ptr(), len(qs), len(buf), stream=st)
            c = ws.finish(st)                      # (raises on a capacity error)
            overflow.append(c["n_overflow"])
            if rep == 0:
                continue
            hit_off = _from_ptr(r.d_hit_off, len(qs) + 1, np.uint64)
            hit_cnt = _from_ptr(r.d_hit_cnt, len(qs), np.uint32)
            pid = _from_ptr(r.d_hit_pid, int(r.hit_capacity), np.uint32)
            km = _from_ptr(r.d_hit_kmatch, int(r.hit_capacity), np.uint32)
            for q in range(0, len(qs), 37):
                p, k, _ = oix.search(qs[q])
                a, b = int(hit_off[q]), int(hit_off[q]) + int(hit_cnt[q])
                assert dict(zip(pid[a:b].tolist(), km[a:b].tolist())) == dict(zip(p.tolist(), k.tolist())), (max_hits, rep, q)
        ws.close()
    small, large = overflow[:3], overflow[3:]
    assert small[0] == large[0] and small[0] > 1000          # scale 1 both times
    assert large[1] * 8 < large[0]                           # the full scale fits
    assert large[1] <= small[1] < small[0]                   # the clamped one still helps, and never hurts
