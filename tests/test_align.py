"""The alignment step (`-aln`): pkg/align/align.go:46-161 + matrixScores.go, called per reported hit by
QueryResultHandler (search.go:483-494).

What the reference's files pin is tested against fixtures parsed from them (tests/golden/matrix_scores.json): the
(matrix, open, extend) -> (lambda, K) table, the key format, AAPosInMatrix; the BitScore / EValue / identity arithmetic
against values worked out by hand from align.go's formulas.  The aligner itself is a third-party library that is not in the
reference tree (biogo v1.0.1): its recurrence is restated (oracle/align_oracle.c), PARITY UNPINNED; the GPU path
(kaamer_align_pairs) is held to that restatement bit for bit -- score, coordinates, every column of the alignment."""
import json
import math
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ALPHA = "-ABCDEFGHIJKLMNPQRSTVWXYZ*"


def _fixture():
    return json.load(open(os.path.join(HERE, "golden", "matrix_scores.json")))


def test_matrix_scores_table_is_the_references(klib, oracle):
    """every row of AllMatrixScores (matrixScores.go:22-105), in both restatements; anything else: "No matrix found" """
    import ctypes as C
    from kaamer_amd import abi
    fx = _fixture()["lambda_k"]
    assert len(fx) == 82
    for key, (lam, k) in fx.items():
        m, go, ge = key.rsplit("_", 2)
        for name in (m, m.upper()):                                   # strings.ToLower on the matrix name
            assert oracle.matrix_scores(name, int(go), int(ge)) == (lam, k), key
            a, b = C.c_double(), C.c_double()
            assert klib.kaamer_align_matrix_scores(name.encode(), int(go), int(ge), C.byref(a), C.byref(b)) == 0
            assert (a.value, b.value) == (lam, k), key
    for bad in (("blosum62", 11, 3), ("blosum62", 12, 2), ("blosum63", 11, 1), ("", 11, 1), ("blosum62_11", 1, 1)):
        assert oracle.matrix_scores(*bad) is None
        a, b = C.c_double(), C.c_double()
        assert klib.kaamer_align_matrix_scores(bad[0].encode(), bad[1], bad[2], C.byref(a), C.byref(b)) == abi.E_ARG


def test_alphabet_and_blosum62(klib, oracle):
    """AAPosInMatrix (matrixScores.go:117) is the order both tables are indexed in; BLOSUM62's published values: symmetric,
    the well-known diagonal, a few off-diagonal entries; product table == restatement for every pair of letters"""
    pos = _fixture()["aa_pos_in_matrix"]
    assert "".join(sorted(pos, key=pos.get)) == ALPHA
    diag = dict(A=4, R=5, N=6, D=6, C=9, Q=5, E=5, G=6, H=8, I=4, L=4, K=5, M=5, F=6, P=7, S=4, T=5, W=11, Y=7, V=4, B=4, Z=4, X=-1)
    for a, v in diag.items():
        assert oracle.b62(a, a) == v, a
    assert oracle.b62("*", "*") == 1 and oracle.b62("A", "*") == -4
    for (a, b), v in {("W", "F"): 1, ("I", "V"): 3, ("D", "E"): 2, ("K", "R"): 2, ("C", "W"): -2, ("G", "I"): -4, ("N", "B"): 3, ("E", "Z"): 4}.items():
        assert oracle.b62(a, b) == oracle.b62(b, a) == v, (a, b)
    for a in ALPHA:
        for b in ALPHA:
            assert oracle.b62(a, b) == oracle.b62(b, a)
            assert klib.kaamer_align_matrix_entry(ord(a), ord(b)) == oracle.b62(a, b), (a, b)
    assert klib.kaamer_align_matrix_entry(ord("a"), ord("A")) == 0        # GetAlnScoreAA: a map miss reads as index 0 ('-')


def test_align_arithmetic_by_hand(oracle):
    """align.go's own arithmetic on alignments small enough to work out by hand"""
    q = "MKTAYIAKQRQISFVKSHFSRQLEERLGLIEVQ"
    r = oracle.align(q, q, 1000)
    raw = sum(oracle.b62(c, c) for c in q)
    assert r["raw"] == raw and r["length"] == len(q) and r["mismatches"] == 0 and r["gap_openings"] == 0
    assert r["identity"] == 100.0 and r["similarity"] == 100.0
    assert (r["q_start"], r["q_end"], r["s_start"], r["s_end"]) == (1, len(q), 1, len(q))
    assert r["aln"] == (q, q, q)
    lam, k = 0.267, 0.041                                            # blosum62_11_1
    bits = (lam * raw - math.log(k)) / math.log(2)                   # align.go:137
    assert r["bitscore"] == bits and r["evalue"] == len(q) * 1000 / 2 ** bits   # align.go:142
    # a five-residue deletion in the subject: one gap feature of length 5 -> raw = matches - 11 - 4 * GapExtend (align.go:127-131)
    s = q[:15] + q[20:]                                              # (a cut whose placement is unique: no letter repeats across it)
    r = oracle.align(q, s, 1000)
    assert r["gap_openings"] == 1 and r["mismatches"] == 0 and r["length"] == len(q)
    assert r["raw"] == sum(oracle.b62(c, c) for c in s) - 11 - 4 * 1
    assert r["aln"][0] == q and r["aln"][2] == q[:15] + "-----" + q[20:] and r["aln"][1] == q[:15] + "     " + q[20:]
    assert r["identity"] == np.float32(np.float32(len(s)) / np.float32(len(q))) * np.float32(100)
    r2 = oracle.align(q, s, 1000, gap_extend=2)                      # blosum62_11_2: the same alignment, other statistics
    assert r2["raw"] == r["raw"] - 4 and r2["bitscore"] == (0.297 * r2["raw"] - math.log(0.082)) / math.log(2)
    # substitutions: '+' where BLOSUM62 is positive, ' ' elsewhere; U reads as '*' (align.go:54-55)
    assert q[5] == "I" and q[12] == "S"
    s = q[:5] + "U" + q[6:12] + "T" + q[13:]                          # I against '*' (-4), S against T (+1)
    r = oracle.align(q, s, 1000)
    assert r["aln"][2][5] == "*" and r["mismatches"] == 2 and r["aln"][1][4] == "Y" and r["aln"][1][5] == " " and r["aln"][1][12] == "+"
    assert r["raw"] == sum(oracle.b62(c, c) for c in q) - 4 - 4 + 1 - 4 and r["similarity"] > r["identity"]
    # "No matrix found": the caller keeps an empty AlignmentResult (align.go:50-52)
    assert oracle.align(q, q, 1000, gap_open=12, gap_extend=2) is None
    with pytest.raises(ValueError):
        oracle.align(q, q[:5] + "O" + q[6:], 1000)


def _mutate(rng, s, n_sub, n_indel):
    aa = "ACDEFGHIKLMNPQRSTVWY"
    s = list(s)
    for _ in range(n_sub):
        s[int(rng.integers(0, len(s)))] = aa[int(rng.integers(0, 20))]
    for _ in range(n_indel):
        at = int(rng.integers(1, len(s) - 1))
        ln = int(rng.integers(1, 7))
        if rng.random() < 0.5:
            del s[at:at + ln]
        else:
            s[at:at] = [aa[int(rng.integers(0, 20))] for _ in range(ln)]
    return "".join(s)


@pytest.mark.gpu
def test_device_alignments_equal_the_restatement(klib, oracle, gpu_device, monkeypatch):
    """kaamer_align_pairs against oracle/align_oracle.c on related, unrelated, nested and degenerate pairs: every number and
    every column of the three rows; both device kernels; several launches (a small direction-array budget)"""
    from kaamer_amd import api, workload
    rng = np.random.default_rng(17)
    db = workload.unpack(workload.make_db(120, seed=5))
    seqs, pairs = [], []
    for i in range(150):
        a = db[int(rng.integers(0, len(db)))].decode()[:int(rng.integers(20, 400))]
        kind = i % 6
        if kind == 0:
            b = _mutate(rng, a, len(a) // 10, 2)
        elif kind == 1:
            b = _mutate(rng, a, len(a) // 3, 4)
        elif kind == 2:
            b = db[int(rng.integers(0, len(db)))].decode()[:int(rng.integers(20, 400))]       # unrelated
        elif kind == 3:
            b = a[len(a) // 4: 3 * len(a) // 4]                                               # nested
        elif kind == 4:
            b = _mutate(rng, a, 3, 1).lower() if rng.random() < 0.5 else _mutate(rng, a, 3, 1).replace("C", "U")
        else:
            b = "X" * 5 + _mutate(rng, a, 5, 1) + "*"
        seqs += [a.encode(), b.encode()]
        pairs.append((len(seqs) - 2, len(seqs) - 1))
    seqs += [b"A", b"", b"MKTAYIAKQRQISFVKSHFSRQLEER", b"MKTAYIOKQRQISFVKSHFSRQLEER"]
    n = len(seqs)
    pairs += [(n - 4, n - 4), (n - 3, n - 2), (n - 2, n - 3), (n - 2, n - 1), (0, 1), (1, 0)]
    n_aa = 2 * 10 ** 8
    # the wave-per-pair kernel (default), several launches of it, the lane-per-pair kernel alone (the fallback for subjects
    # beyond the LDS row buffer), and both in one call
    for budget, wave_ns in ((None, None), ("200000", None), (None, "0"), ("3000000", "100")):
        for k_, v_ in (("KAAMER_ALIGN_DIR_BYTES", budget), ("KAAMER_ALIGN_WAVE_NS", wave_ns)):
            if v_ is None:
                monkeypatch.delenv(k_, raising=False)
            else:
                monkeypatch.setenv(k_, v_)
        got = api.align_pairs(seqs=seqs, pairs=pairs, number_of_aa=n_aa, device=gpu_device)
        n_gaps = 0
        for (qi, si), g in zip(pairs, got):
            q, s = seqs[qi], seqs[si]
            try:
                exp = oracle.align(q, s, n_aa)
            except ValueError:
                assert g["status"] == 2, (qi, si)
                continue
            assert g["status"] == 0
            for k in ("length", "mismatches", "gap_openings", "raw", "bitscore", "evalue"):
                assert g[k] == exp[k] or (isinstance(exp[k], float) and math.isnan(exp[k]) and math.isnan(g[k])), (qi, si, k, g[k], exp[k])
            for k, ke in (("query_start", "q_start"), ("query_end", "q_end"), ("subject_start", "s_start"), ("subject_end", "s_end")):
                assert g[k] == exp[ke], (qi, si, k)
            for k in ("identity", "similarity"):
                assert g[k] == exp[k] or (math.isnan(g[k]) and math.isnan(exp[k])), (qi, si, k)
            assert g["aln"] == exp["aln"], (qi, si)
            n_gaps += exp["gap_openings"]
        assert n_gaps > 50
    # options without a row in the table: the reference keeps empty results
    assert api.align_pairs(seqs=seqs, pairs=pairs[:3], number_of_aa=n_aa, gap_open=12, gap_extend=2, device=gpu_device) == [None] * 3


@pytest.mark.gpu
def test_search_then_align_like_query_result_handler(klib, oracle, gpu_device):
    """search.go:483-494 end to end: ProteinSearch -> FetchHitsInformation -> one alignment per reported hit (the query
    against the hit's Sequence, NumberOfAA from the table's KStats) -> hits re-sorted by BitScore; against the restatement"""
    from kaamer_amd import api, search, workload
    db = workload.make_db(400, seed=41)
    recs = workload.unpack(db)
    fasta = "".join(">sp|P%05d|N%d\n%s\n" % (i, i, s.decode()) for i, s in enumerate(recs))
    prot = api.Proteins.from_fasta(fasta.encode())
    ix = api.Index.from_image(prot.image(device=gpu_device), gpu_device)
    qs = workload.unpack(workload.make_protein_queries(db, 30, seed=42))
    qtext = "".join(">q%d\n%s\n" % (i, s.decode()) for i, s in enumerate(qs))
    res = search.FetchHitsInformation(search.ProteinSearch(ix, qtext, search.SearchOptions(MaxResults=5)), prot)
    res = search.AlignHits(res, prot, search.SearchOptions(MaxResults=5), device=gpu_device)
    n_aa = prot.stats()["NumberOfAA"]
    n = 0
    for qr in res:
        hits = qr["SearchResults"]["Hits"]
        bits = [h["Alignment"]["BitScore"] for h in hits]
        assert bits == sorted(bits, reverse=True)
        for h in hits:
            exp = oracle.align(qr["Query"]["Sequence"], qr["HitEntries"][h["Key"]]["Sequence"], n_aa)
            a = h["Alignment"]
            assert (a["Raw"], a["Length"], a["Mismatches"], a["GapOpenings"]) == (exp["raw"], exp["length"], exp["mismatches"], exp["gap_openings"])
            assert a["BitScore"] == exp["bitscore"] and a["EValue"] == exp["evalue"] and a["Identity"] == exp["identity"]
            assert a["AlnString"] == "\n".join(exp["aln"])
            assert (a["QueryStart"], a["QueryEnd"], a["SubjectStart"], a["SubjectEnd"]) == (exp["q_start"], exp["q_end"], exp["s_start"], exp["s_end"])
            n += 1
    assert n > 60


@pytest.mark.gpu
def test_alignment_properties_at_batch_size(klib, gpu_device):
    """size-independent properties on a batch as large as a real one (~12 000 pairs): a sequence against itself aligns end to
    end with identity 100 and Raw = the sum of its BLOSUM62 diagonal (U counted as '*', align.go:54-55); swapping query and
    subject of a gap-free alignment leaves Raw alone; a subject that contains the query as a substring scores the query's
    self score.  (Sequences with X are left out: X against X scores -1, a local alignment drops one at an end.)"""
    from kaamer_amd import api, workload
    db = [s for s in workload.unpack(workload.make_db(5000, seed=12)) if b"X" not in s]
    rng = np.random.default_rng(3)
    seqs = list(db)
    pairs = [(i, i) for i in range(len(db))]
    for i in range(0, len(db) - 1, 2):
        pairs += [(i, i + 1), (i + 1, i)]
    sub = []
    for i in range(0, min(2500, len(db))):
        s = db[i]
        a = int(rng.integers(0, max(1, len(s) // 3)))
        b = int(rng.integers(2 * len(s) // 3, len(s))) + 1
        seqs.append(s[a:b])
        sub.append((len(seqs) - 1, i))
    pairs += sub
    got = api.align_pairs(seqs=seqs, pairs=pairs, number_of_aa=10 ** 8, device=gpu_device)
    n = len(db)
    diag = lambda s: sum(int(klib.kaamer_align_matrix_entry(c, c)) for c in s.replace(b"U", b"*"))
    for i in range(n):
        g = got[i]
        assert g["raw"] == diag(db[i]) and g["length"] == len(db[i]) and g["identity"] == 100.0 and g["gap_openings"] == 0
        assert (g["query_start"], g["query_end"], g["subject_start"], g["subject_end"]) == (1, len(db[i]), 1, len(db[i]))
    for k in range(n, n + 2 * (n // 2), 2):
        if got[k]["gap_openings"] == 0 and got[k + 1]["gap_openings"] == 0:   # (Raw charges gap extensions after the fact: ties may differ)
            assert got[k]["raw"] == got[k + 1]["raw"]        # the optimal local score does not depend on which sequence is the query
    for (qi, si), g in zip(sub, got[-len(sub):]):
        assert g["raw"] == diag(seqs[qi]) and g["identity"] == 100.0 and g["query_end"] == len(seqs[qi])
