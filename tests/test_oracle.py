"""Pins the CPU oracle: reference-text fixtures, hand-derived KATs, and an
independent pure-Python restatement (tests/pyref.py)."""
import json
import os
import random

import numpy as np
import pytest

import pyref

GOLD = os.path.join(os.path.dirname(__file__), "golden")

# hand-derived from pkg/kvstore/k_store.go:39-117
KATS = [("AAAAAAA", 0x0B0582C0), ("YYYYYYY", 0xE773B9D4), ("MELPNIM", 0x75B7E08A),
        ("ACDEFGH", 0x0B90CDE6), ("XAAAAAA", 0x000582C0), ("UUUUUUU", 0xC6633191),
        ("MKVLAAG", 0x786642C5), ("AAAAAAX", 0x0B0582C0), ("AAAAAA*", 0x0B0582C0),
        ("XXXXXXX", 0x00000000), ("aaaaaaa", 0x00000000), ("AXAAAAA", 0x000582C0),
        # aaTable[{a,'.'}] = j (k_store.go:48-52) is also found by the PAIR lookup (k_store.go:102-103):
        # a pair whose second letter is '.' encodes as the index of its first letter
        ("C.AAAAA", 0x008582C0), ("Y.Y.Y.Y", 0x0A050294), ("A.AAAAA", 0x000582C0), (".AAAAAA", 0x000582C0),
        ("AAAAAA.", 0x0B0582C0), ("AA..AAA", 0x0B0002C0), ("AAAAY.A", 0x0B058280)]

ORF_KAT_READ = ("GCTAAAGACAATTACATAACATACACGTCAGCACGAAACTTGTTGGCCCAGTGTGAATCGCTTAAGGGTTAAGTAAGTGTGATGCATACGCC"
                "TTTACTTGCTGTGTCCACCCCATCGGACTGGCATTTTTATTACACTCAGAAACAGAAC")
ORF_KAT = [("AKDNYITYTSARNLLAQCESLKG*", 1, 72, True, [5, 13, 14]),
           ("IRLYLLCPPHRTGIFITLRNR", 86, 148, True, [0, 4, 5, 13, 15]),
           ("SVSECNKNASPMGWTQQVKAYASHLLNP*", 148, 62, False, [11]),
           ("VMHTPLLAVSTPSDWHFYYTQKQN", 79, 150, True, [0, 1, 8])]


@pytest.mark.parametrize("kmer,key", KATS)
def test_codec_kat(oracle, kmer, key):
    assert oracle.encode_kmer(kmer) == key
    assert pyref.encode_kmer(kmer) == key
    assert oracle.create_bytes_key(kmer) == key.to_bytes(4, "big")  # k_store.go:69-70


def test_codec_roundtrip_and_random(oracle):
    rng = random.Random(1)
    alpha = "ACDEFGHIKLMNPQRSTUVWY"
    for _ in range(2000):
        k = "".join(rng.choice(alpha) for _ in range(7))
        key = oracle.encode_kmer(k)
        assert key == pyref.encode_kmer(k)
        assert oracle.decode_kmer(key) == k
        assert key <= 0xE773B9D4
    noisy = alpha + "BJOXZ*abc-"
    for _ in range(2000):
        k = "".join(rng.choice(noisy) for _ in range(7))
        assert oracle.encode_kmer(k) == pyref.encode_kmer(k)


def test_codec_all_bytes_every_position(oracle, klib):
    """oracle == pyref == the product's kaamer_encode_kmer for every byte value in every position and,
    exhaustively, for every byte PAIR in each of the three pair positions (k_store.go:100-110)."""
    base = bytearray(b"MKVLAAG")
    for pos in range(7):
        for b in range(256):
            k = bytes(base[:pos]) + bytes([b]) + bytes(base[pos + 1:])
            e = pyref.encode_kmer(k)
            assert oracle.encode_kmer(k) == e, (pos, b)
            assert klib.kaamer_encode_kmer(k) == e, (pos, b)
    # pyref's table as arrays, so the 3 x 65 536 pair sweep compares two C implementations against it cheaply
    pair = np.zeros((256, 256), np.uint32)
    for (a, b), v in pyref._AA_TABLE.items():
        pair[ord(a), ord(b)] = v
    for s, sh in ((0, 23), (2, 14), (4, 5)):
        rest = pyref.encode_kmer(bytes(base)) & ~(0x1FF << sh) & 0xFFFFFFFF
        for a in range(256):
            for b in range(256):
                k = bytes(base[:s]) + bytes([a, b]) + bytes(base[s + 2:])
                e = rest | (int(pair[a, b]) << sh)
                assert oracle.encode_kmer(k) == e, (s, a, b)
                assert klib.kaamer_encode_kmer(k) == e, (s, a, b)
    rng = random.Random(5)
    for _ in range(20000):  # and random byte soup, '.' over-represented
        k = bytes(rng.choice(b"ACDEFGHIKLMNPQRSTUVWY....XBZ*a") if rng.random() < 0.8 else rng.randrange(256) for _ in range(7))
        e = pyref.encode_kmer(k)
        assert oracle.encode_kmer(k) == e and klib.kaamer_encode_kmer(k) == e, k


def test_gcode_all_bytes(oracle):
    """gcodeBacteria is a map keyed by the lower-cased codon string (dna.go:68,106): any byte other than
    t/c/a/g in any position is a map miss -> AminoAcid{} ("", false, false).  All 256 values, every position."""
    table = json.load(open(os.path.join(GOLD, "gcode_bacteria.json")))
    for pos in range(3):
        for b in range(256):
            for rest in ("tt", "ca", "ag", "gg", "at", "tg"):
                codon = bytearray(rest[:pos].encode() + bytes([b]) + rest[pos:].encode())
                exp = ("", False, False)
                txt = codon.decode("latin-1")
                if txt in table:
                    exp = tuple(table[txt])
                assert oracle.gcode_bacteria(bytes(codon)) == exp, codon
                assert pyref.GCODE_BACTERIA.get(txt, ("", False, False)) == exp


def test_gcode_matches_reference_text(oracle):
    table = json.load(open(os.path.join(GOLD, "gcode_bacteria.json")))
    assert len(table) == 64
    for codon, (aa, start, stop) in table.items():
        assert oracle.gcode_bacteria(codon) == (aa, start, stop), codon
        assert pyref.GCODE_BACTERIA[codon] == (aa, start, stop), codon
    for codon in ("nnn", "atn", "ATG", "a-g", "xyz"):
        assert oracle.gcode_bacteria(codon) == ("", False, False)


def test_gcode_fixture_is_current():
    ref = "/root/reference/pkg/search/gcode.go"
    if not os.path.exists(ref):
        pytest.skip("reference tree not present (GPU box)")
    import importlib.util
    spec = importlib.util.spec_from_file_location("mk", os.path.join(GOLD, "make_reference_fixtures.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    assert mk.gcode() == {k: v for k, v in json.load(open(os.path.join(GOLD, "gcode_bacteria.json"))).items()}
    assert mk.docs_example() == json.load(open(os.path.join(GOLD, "docs_example.json")))


def test_docs_worked_example(oracle):
    """docs/client.md:131-180: 270 aa, SizeInKmer 264, self hit Kmatch 264, all positions."""
    ex = json.load(open(os.path.join(GOLD, "docs_example.json")))
    assert len(ex["query"]) == 270 and ex["db_length"] == 270
    assert oracle.size_in_kmer(ex["query"]) == ex["size_in_kmer"] == 264
    ix = oracle.Index.from_proteins([ex["db_sequence"]], ids=[ex["hit_key"]])
    pid, km, pos = ix.search(ex["query"], want_positions=True)
    assert pid.tolist() == [ex["hit_key"]] and km.tolist() == [ex["kmatch"]]
    assert pos.shape == (1, ex["n_positions"]) and pos.all() == ex["all_positions_true"]
    assert oracle.filter_results(km, 264, 0.05, 10, 10) == 1


def test_orf_kat(oracle):
    got = oracle.get_orfs(ORF_KAT_READ)
    assert [(o["seq"], o["start"], o["end"], o["plus"], o["starts"]) for o in got] == ORF_KAT
    assert got == pyref.get_orfs(ORF_KAT_READ)


def _rand_dna(rng, n, alphabet="ACGT"):
    return "".join(rng.choice(alphabet) for _ in range(n))


def test_orfs_random_vs_pyref(oracle):
    rng = random.Random(7)
    for n in list(range(0, 30)) + [63, 64, 65, 66, 100, 149, 150, 151, 250, 1000]:
        for _ in range(6):
            d = _rand_dna(rng, n)
            assert oracle.get_orfs(d) == pyref.get_orfs(d), d
    for _ in range(100):  # N, lower case, junk
        d = _rand_dna(rng, rng.randint(60, 300), "ACGTacgtNn")
        assert oracle.get_orfs(d) == pyref.get_orfs(d), d
    # stop-free frames: whole-frame ORFs, and ORF length exactly at the 21 threshold
    d = "ATG" + "GCA" * 19 + "TAA"       # 20 aa + '*' = 21 -> emitted
    assert any(o["seq"] == "M" + "A" * 19 + "*" for o in oracle.get_orfs(d))
    d = "ATG" + "GCA" * 18 + "TAA"       # 19 aa + '*' = 20 -> dropped
    assert not any(o["plus"] and o["start"] == 1 for o in oracle.get_orfs(d))


def _rand_prot(rng, n, alpha="ACDEFGHIKLMNPQRSTVWY"):
    return "".join(rng.choice(alpha) for _ in range(n))


def test_search_random_vs_pyref(oracle):
    rng = random.Random(3)
    founders = [_rand_prot(rng, rng.randint(5, 120)) for _ in range(20)]
    db = []
    for f in founders:
        db.append(f)
        for _ in range(3):
            s = list(f)
            for i in range(len(s)):
                if rng.random() < 0.15:
                    s[i] = rng.choice("ACDEFGHIKLMNPQRSTVWYXUB")
            db.append("".join(s))
    ids = list(range(100, 100 + len(db)))
    ix = oracle.Index.from_proteins(db, ids=ids)
    pix = pyref.build_index(db, ids=ids)
    pairs = ix.pairs()
    assert len(pairs) == sum(len(v) for v in pix.values())
    for q in db[:30] + [_rand_prot(rng, 50), db[0] + "*", db[3][:12], "AAAAAAA", "AAAAAA", ""]:
        size = oracle.size_in_kmer(q)
        assert size == pyref.size_in_kmer(q)
        pid, km, pos = ix.search(q, want_positions=True)
        counter, positions = pyref.kmer_search(pix, q)
        assert dict(zip(pid.tolist(), km.tolist())) == counter
        for i, p in enumerate(pid.tolist()):
            assert pos[i].tolist() == positions[p]
        assert list(km) == sorted(km, reverse=True)
        hits = list(zip(pid.tolist(), km.tolist()))
        for opts in ((0.05, 10, 10), (0.5, 1, 3), (0.0, 0, 1000), (0.9, 50, 2)):
            if size > 0:
                assert oracle.filter_results(km, size, *opts) == len(pyref.filter_results(hits, size, *opts))


def test_self_hit_invariant_and_sum(oracle):
    """docs invariant: a DB protein scores len-6 on itself; sum_p Kmatch = sum_pos |index[key]|."""
    from kaamer_amd import workload
    db = workload.make_db(200, seed=11)
    seqs = workload.unpack(db)
    ix = oracle.Index.from_proteins(None, packed=db)
    for p in (0, 5, 77, 199):
        pid, km, _ = ix.search(seqs[p])
        d = dict(zip(pid.tolist(), km.tolist()))
        assert d[p] == len(seqs[p]) - 6
        tot = sum(len(ix.get(oracle.encode_kmer(seqs[p][k:k + 7]))) for k in range(len(seqs[p]) - 6))
        assert sum(d.values()) == tot


def test_fasta_id_rule(oracle):
    """inputFASTA.go:95-124: record k gets id k+1, the last two records share id N."""
    assert oracle.fasta_ids(1).tolist() == [1]
    assert oracle.fasta_ids(2).tolist() == [2, 2]
    assert oracle.fasta_ids(5).tolist() == [2, 3, 4, 5, 5]


def test_best_start_codon_vs_pyref(oracle):
    rng = random.Random(5)
    for _ in range(300):
        n = rng.randint(21, 60)
        seq = _rand_prot(rng, n - 1) + rng.choice(["*", "A"])
        size = oracle.size_in_kmer(seq)
        nh = rng.randint(1, 4)
        km = sorted((rng.randint(1, size) for _ in range(nh)), reverse=True)
        pos = np.zeros((nh, size), dtype=bool)
        for h in range(nh):
            pos[h, rng.sample(range(size), km[h])] = True
        starts = sorted(rng.sample(range(n), rng.randint(0, 4)))
        plus = rng.random() < 0.5
        start_position = rng.randint(1, 500)
        hits = [(h, km[h]) for h in range(nh)]
        q = dict(seq=seq, start=start_position, plus=plus, starts=starts, size=size)
        q2, trimmed = pyref.set_best_start_codon(hits, {h: pos[h].tolist() for h in range(nh)}, q)
        t, sp, so = oracle.set_best_start_codon(km, pos, size, starts, plus, seq, start_position)
        assert (t, sp, so) == (trimmed, q2["start"], q2["size"])
