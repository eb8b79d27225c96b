"""GPU parity tests of the nucleotide / reads path: 6-frame translation + ORF
segmentation on the device (GetORFs, dna.go:65-181) and the per-ORF k-mer search,
against the CPU oracle."""
import numpy as np
import pytest

from kaamer_amd import abi

pytestmark = pytest.mark.gpu


def _gpu_orfs(res):
    """BatchResult -> per input sequence, list of dict like oracle.get_orfs"""
    out = {}
    for i in range(res.n_queries):
        m = res.meta[i]
        aa = bytes(res.orf_aa[int(m["aa_off"]):int(m["aa_off"]) + int(m["aa_len"])]).decode("latin-1")
        sa = res.starts_alt[int(m["sa_off"]):int(m["sa_off"]) + int(m["sa_len"])].tolist()
        out.setdefault(int(m["src_seq"]), []).append(
            dict(seq=aa, start=int(m["start_position"]), end=int(m["end_position"]), plus=bool(m["plus_strand"]),
                 starts=sa, size=int(m["size_in_kmer"]), q=i))
    return out


def _check_reads(res, reads, oracle, oix, check_hits=True):
    got = _gpu_orfs(res)
    n_orfs = 0
    for r, read in enumerate(reads):
        exp = oracle.get_orfs(read)
        g = got.get(r, [])
        assert [{k: o[k] for k in ("seq", "start", "end", "plus", "starts")} for o in g] == exp, "read %d" % r
        n_orfs += len(exp)
        for o in g:
            assert o["size"] == oracle.size_in_kmer(o["seq"])
            if check_hits:
                pid, km, pos = oix.search(o["seq"], want_positions=True)
                assert res.hits(o["q"]) == dict(zip(pid.tolist(), km.tolist())), "read %d orf %d" % (r, o["q"])
                assert res.first_pos(o["q"]) == {int(p): int(np.argmax(pos[i])) for i, p in enumerate(pid)}
    assert res.n_queries == n_orfs
    # queries are grouped by input sequence, in input order
    assert res.meta["src_seq"].tolist() == sorted(res.meta["src_seq"].tolist())
    return n_orfs


@pytest.fixture(scope="module")
def small(klib, oracle, gpu_device):
    from kaamer_amd import api, workload
    db = workload.make_db(1000)
    ix = api.Index.from_image(api.Image.from_proteins(packed=db), gpu_device)
    oix = oracle.Index.from_proteins(None, packed=db)
    return db, ix, oix


def test_orf_kat(small, oracle):
    from test_oracle import ORF_KAT, ORF_KAT_READ
    db, ix, oix = small
    res = ix.search([ORF_KAT_READ], seq_type=abi.READS)
    g = _gpu_orfs(res)[0]
    assert [(o["seq"], o["start"], o["end"], o["plus"], o["starts"]) for o in g] == ORF_KAT


def test_reads_config3_sample(small, oracle):
    """Q-R150 reads (back-translated DB windows, both strands, 1 % substitutions, N's)"""
    from kaamer_amd import workload
    db, ix, oix = small
    reads = workload.unpack(workload.make_reads(db, 400))
    res = ix.search(reads, seq_type=abi.READS)
    n = _check_reads(res, reads, oracle, oix)
    assert n > 400
    c = res.counters
    assert c["n_queries"] == n and c["n_overflow"] == 0
    assert c["n_lookup"] == int(res.meta["size_in_kmer"].sum())
    assert c["n_hits"] > 300      # most reads come from the DB


def test_reads_edge_cases(small, oracle):
    import random
    db, ix, oix = small
    rng = random.Random(11)

    def rnd(n, alpha="ACGT"):
        return "".join(rng.choice(alpha) for _ in range(n)).encode()

    reads = [b"", b"A", b"AC", b"ACG", b"ACGT"] + [rnd(n) for n in range(5, 70)]
    reads += [rnd(n) for n in (63, 64, 65, 66, 126, 127, 128, 129, 189, 190, 191, 192, 193, 194, 195, 250, 251, 252)]
    reads += [rnd(rng.randint(60, 400), "ACGTacgtNn") for _ in range(60)]          # lower case, N
    reads += [rnd(rng.randint(60, 300), "ACGTRYKM-*") for _ in range(20)]          # IUPAC / junk bytes
    reads += [b"ATG" + b"GCA" * 19 + b"TAA",        # 20 aa + '*' = 21 -> emitted
              b"ATG" + b"GCA" * 18 + b"TAA",        # 20 -> dropped
              b"GCA" * 21, b"GCA" * 20,             # stop-free frame of exactly 21 / 20 codons
              b"TAA" * 40, b"ATG" * 70, b"TAAATG" * 30,
              b"GCA" * 200,                          # one ORF spanning four 64-codon chunks, no stop
              (b"ATG" + b"GCA" * 30 + b"TAG") * 8,  # repeated ORFs
              b"N" * 150, b"GCANNN" * 40]
    # the lane-per-read kernel's limits (reads of up to 192 nt, three ORFs queued per frame of 64 codons): three
    # 21-codon ORFs back to back in one frame, on either strand, in every frame; two ORFs that touch (the second
    # opens on the codon after the first one's stop); unknown codons inside an ORF; a 64-codon ORF with 64 starts
    def revcomp(s):
        return s[::-1].translate(bytes.maketrans(b"ACGT", b"TGCA"))
    orf21 = b"ATG" + b"GCA" * 19 + b"TAA"
    three = orf21 * 3
    reads += [three + b"AC", b"A" + three + b"C", b"AC" + three + b"A", three + b"ACG",
              revcomp(three + b"AC"), revcomp(b"A" + three + b"C"), revcomp(b"AC" + three + b"A"),
              orf21 + b"TTG" + b"GCA" * 19 + b"TGA" + b"CTG" + b"GCA" * 20,         # alternative starts, open last ORF
              b"ATG" + b"GCANNA" * 4 + b"GCA" * 14 + b"TAA" + b"GC",               # unknown codons do not count
              b"ATG" * 64, revcomp(b"ATG" * 64), b"ATGTTG" * 32, (b"GTG" + b"GCA" * 20 + b"TAG") * 2 + b"GTGGCA" * 9]
    assert max(len(r) for r in reads[-13:]) <= 192
    res = ix.search(reads, seq_type=abi.READS)
    n = _check_reads(res, reads, oracle, oix)
    assert sum(len(oracle.get_orfs(r)) >= 3 for r in reads[-13:]) >= 6


def test_contig_many_orfs(small, oracle):
    """NUCLEOTIDE input: one long contig, hundreds of ORFs, the reference's position order"""
    from kaamer_amd import workload
    db, ix, oix = small
    rng = np.random.default_rng(5)
    contig = bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 60000)])
    # plant DB genes on both strands
    prot = workload.unpack(db)
    reads = workload.unpack(workload.make_reads(db, 40, read_len=600, seed=3))
    c = bytearray(contig)
    for i, r in enumerate(reads):
        c[1000 + i * 1400:1000 + i * 1400 + len(r)] = r
    contigs = [bytes(c), contig[:5000]]
    res = ix.search(contigs, seq_type=abi.NUCLEOTIDE)
    n = _check_reads(res, contigs, oracle, oix)
    assert n > 500
    assert len(prot) == 1000


def test_reads_device_call_reuse(small, oracle):
    """device-resident call for reads, workspace reused across batches of different size"""
    import torch
    from test_gpu_protein import _from_ptr
    from kaamer_amd import api, workload
    db, ix, oix = small
    ws = api.Workspace(ix, 200 * 150, 200, seq_type=abi.READS)
    st = torch.cuda.current_stream().cuda_stream
    for n, seed in ((200, 1), (50, 2), (200, 3)):
        reads = workload.make_reads(db, n, seed=seed)
        buf, offs = reads
        d_buf = torch.from_numpy(buf).cuda()
        d_off = torch.from_numpy(offs.view(np.int64)).cuda()
        r = ws.search_device(d_buf.data_ptr(), d_off.data_ptr(), n, len(buf), stream=st)
        c = ws.finish(st)
        exp_orfs = [oracle.get_orfs(x) for x in workload.unpack(reads)]
        nq = int(_from_ptr(r.d_n_queries, 1, np.uint32)[0])
        assert nq == sum(len(e) for e in exp_orfs) == c["n_queries"]
        hit_off = _from_ptr(r.d_hit_off, nq + 1, np.uint64)
        hit_cnt = _from_ptr(r.d_hit_cnt, nq, np.uint32)
        pid = _from_ptr(r.d_hit_pid, int(r.hit_capacity), np.uint32)
        km = _from_ptr(r.d_hit_kmatch, int(r.hit_capacity), np.uint32)
        q = 0
        for e in exp_orfs:
            for o in e:
                p, k, _ = oix.search(o["seq"])
                a, b = int(hit_off[q]), int(hit_off[q]) + int(hit_cnt[q])
                assert dict(zip(pid[a:b].tolist(), km[a:b].tolist())) == dict(zip(p.tolist(), k.tolist()))
                q += 1


def test_reads_position_bitmaps(small, oracle):
    """reads always carry PositionHits in the reference (search.go:416); optional here"""
    from kaamer_amd import workload
    db, ix, oix = small
    reads = workload.unpack(workload.make_reads(db, 120, seed=4))
    res = ix.search(reads, seq_type=abi.READS, want_positions=True)
    got_orfs = _gpu_orfs(res)
    n = 0
    for r, read in enumerate(reads):
        for o in got_orfs.get(r, []):
            pid, km, pos = oix.search(o["seq"], want_positions=True)
            got = res.positions(o["q"])
            assert sorted(got) == sorted(pid.tolist())
            for i, p in enumerate(pid.tolist()):
                assert got[p].tolist() == pos[i].tolist()
                n += 1
    assert n > 100


def test_contig_piece_boundaries(small, oracle):
    """long sequences are translated in pieces of 4096 codons per frame: ORFs that start in one piece
    and close in a later one, stops on the last / first codon of a piece, a stop-free frame longer
    than two pieces, unknown bases across a boundary, lengths around the piece size"""
    db, ix, oix = small
    rng = np.random.default_rng(77)

    def rnd(n):
        return bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n)])

    P = 4096
    orf = lambda n_codons: b"ATG" + b"GCA" * (n_codons - 2) + b"TAA"   # n_codons incl. start and stop
    contigs = [
        orf(P) + orf(30) + rnd(3000),                       # stop on the last codon of piece 0
        orf(P + 1) + orf(30) + rnd(3000),                   # stop on the first codon of piece 1
        orf(P - 1) + orf(25) + rnd(100),                    # next ORF starts on the last codon of piece 0
        b"GCA" * (2 * P + 500),                             # stop-free: one ORF over three pieces, open at both ends
        rnd(5000) + orf(9000) + rnd(5000),                  # an ORF crossing two boundaries inside random sequence
        b"AC" + orf(P) + rnd(64),                           # frame 3 carries the construct
        rnd(3 * P - 40) + b"N" * 90 + rnd(2000),            # unknown bases across the boundary
        rnd(3 * P), rnd(3 * P + 1), rnd(3 * P + 2), rnd(3 * P + 3), rnd(6 * P + 5),
        rnd(193), rnd(192), rnd(200000),
    ]
    res = ix.search(contigs, seq_type=abi.NUCLEOTIDE)
    n = _check_reads(res, contigs, oracle, oix)
    assert n > 1000


@pytest.mark.parametrize("wide", ["0", "1"])
def test_both_lane_per_read_kernels(small, oracle, monkeypatch, wide):
    """translate_reads_kernel exists for reads of up to 192 nt (64 codons per frame, three ORFs queued per frame) and of
    up to 384 nt (128 codons: two mask words, six ORFs per frame); the library picks per batch from the mean read length
    and KAAMER_WIDE_READS forces either.  The same reads through both: lengths around both limits (what is longer goes a
    wave per frame through translate_kernel), six 21-codon ORFs back to back in one frame on either strand, ORFs that
    start in the first mask word and end in the second, 128 start codons in a row."""
    import random
    from kaamer_amd import workload
    db, ix, oix = small
    monkeypatch.setenv("KAAMER_WIDE_READS", wide)
    rng = random.Random(23)

    def rnd(n, alphabet="ACGT"):
        return "".join(rng.choice(alphabet) for _ in range(n)).encode()

    def revcomp(s):
        return s[::-1].translate(bytes.maketrans(b"ACGT", b"TGCA"))
    reads = [rnd(n) for n in (189, 190, 191, 192, 193, 194, 195, 196, 250, 251, 252, 299, 300, 301, 381, 382, 383, 384, 385, 386, 387, 500)]
    reads += workload.unpack(workload.make_reads(db, 60, read_len=250, seed=5)) + workload.unpack(workload.make_reads(db, 60, read_len=300, seed=6))
    reads += workload.unpack(workload.make_reads(db, 40, read_len=384, seed=7)) + workload.unpack(workload.make_reads(db, 40, read_len=150, seed=8))
    reads += [rnd(rng.randint(180, 400), "ACGTacgtNn") for _ in range(60)]
    orf21 = b"ATG" + b"GCA" * 19 + b"TAA"
    six = orf21 * 6
    reads += [six + b"AC", b"A" + six + b"C", b"AC" + six + b"A", six + b"ACGTAC", revcomp(six + b"AC"), revcomp(b"A" + six + b"C"),
              revcomp(b"AC" + six + b"A"),
              b"GCA" * 128, b"GCA" * 127 + b"TA", b"ATG" * 128, revcomp(b"ATG" * 128), b"ATGTTG" * 64,
              b"TAA" + b"ATG" + b"GCA" * 100 + b"TAG" + b"GTG" + b"GCA" * 20,      # an ORF across the two mask words, a last open one
              b"GCA" * 50 + b"GCANNA" * 10 + b"GCA" * 40 + b"TGA" + b"AC",          # unknown codons inside an ORF of 120 codons
              (b"TTG" + b"GCA" * 20 + b"TAG") * 5 + b"CTGGCA" * 9]
    assert max(len(r) for r in reads[-15:]) <= 384
    res = ix.search(reads, seq_type=abi.READS)
    _check_reads(res, reads, oracle, oix)
    assert sum(len(oracle.get_orfs(r)) >= 6 for r in reads[-15:]) >= 6
