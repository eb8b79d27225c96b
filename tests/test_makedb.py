"""makedb acceptance and id rules (pkg/makedb/inputFASTA.go, inputTSV.go) and the protein table behind
FetchHitsInformation (search.go:454-470), against the oracle's literal restatement.  CPU only."""
import os

import numpy as np
import pytest

from kaamer_amd import abi, api


def _makedb_ref():
    from oracle import makedb_ref
    return makedb_ref


def _pairs_of(records, oracle):
    """what the reference's KmerStore ends up with: {key -> sorted distinct ids} as (keys, ids) arrays"""
    oix = oracle.Index.from_proteins([r[2] for r in records], ids=np.array([r[0] for r in records], np.uint32))
    return oix


def _image_pairs(img, oix):
    pr = oix.pairs()                       # (key << 32 | id), sorted, distinct
    keys, ids = (pr >> np.uint64(32)).astype(np.uint32), (pr & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    last = None
    for k in np.unique(keys):
        got = img.get(int(k))
        exp = np.unique(ids[keys == k])
        assert got.tolist() == exp.tolist(), hex(int(k))
        last = k
    assert img.stats()["n_keys"] == len(np.unique(keys))
    return last


FASTA = b""">sp|P1|A first protein OS=Somewhere
MKTAYIAKQR
qistfvkshf
>sp|P2|B hypothetical protein, partial
MKTAYIAKQRQISTFVKSHFSRQ
>P3
MKTAY
>P4 lower case and a dot
mktayiakqr.istfvk\r
>P5 two from the end
ACDEFGHIKLMNPQRSTVWY
>P6 the last record shares its id with P5
YWVTSRQPNMLKIHGFEDCA
"""


def test_fasta_rules_match_the_reference(klib, oracle):
    ref = _makedb_ref().run_fasta(FASTA)
    # the id quirk itself: record k -> k + 1, the last two share N (inputFASTA.go:98-124)
    assert [(r[0], r[1]) for r in ref] == [(2, b"sp|P1|A"), (5, b"P4"), (6, b"P5"), (6, b"P6")]
    p = api.Proteins.from_fasta(FASTA)
    buf, offs = p.packed
    got = [(int(i), bytes(buf[int(offs[j]):int(offs[j + 1])])) for j, i in enumerate(p.ids)]
    assert got == [(r[0], r[2]) for r in ref]
    assert got[0][1] == b"MKTAYIAKQRQISTFVKSHF" and got[1][1] == b"MKTAYIAKQR.ISTFVK"
    assert p.feature_names == [b"ProteinName"]
    st = p.stats()
    assert (st["NumberOfProteins"], st["NumberOfAA"]) == (4, sum(len(r[2]) for r in ref))
    assert st["NumberOfKmers"] == sum(len(r[2]) - 6 for r in ref)
    img = p.image()
    _image_pairs(img, _pairs_of(ref, oracle))
    # FetchHitsInformation: id 6 holds the LAST record written under it (ProteinStore overwrite)
    e = p.fetch_hits([2, 6, 5, 3, 99])
    assert e[3] is None and e[4] is None
    assert e[0] == dict(EntryId=b"sp|P1|A", Sequence=b"MKTAYIAKQRQISTFVKSHF", Length=20,
                        Features={b"ProteinName": b"first protein OS=Somewhere"})
    assert e[1]["EntryId"] == b"P6" and e[1]["Sequence"] == b"YWVTSRQPNMLKIHGFEDCA"
    assert e[2]["Features"] == {b"ProteinName": b"lower case and a dot"}


def test_fasta_text_before_the_first_header_and_no_final_newline(klib, oracle):
    text = b"ACDEFGHIKLMN\n>A x\nMKTAYIAKQR\n>B y\nMKTAYIAKQRQISTFVK"
    ref = _makedb_ref().run_fasta(text)
    assert [(r[0], r[1], r[2]) for r in ref] == [(1, b"", b"ACDEFGHIKLMN"), (2, b"A", b"MKTAYIAKQR"), (2, b"B", b"MKTAYIAKQRQISTFVK")]
    p = api.Proteins.from_fasta(text)
    assert p.ids.tolist() == [1, 2, 2]
    _image_pairs(p.image(), _pairs_of(ref, oracle))
    assert len(api.Proteins.from_fasta(b"")) == 0 and len(api.Proteins.from_fasta(b">only a header\n")) == 0


TSV = b"""EntryID\tOrganism\tSEQUENCE\tEC
P1\tE. coli\tMKTAYIAKQRQISTFVK\t1.1.1.1
\tno id\tMKTAYIAKQRQISTFVK\t-
P3\tshort\tMKTAY\t-
P4\tlower stays lower\tmktayiakqrqistfvk\t2.7.7.7
P5\tshort row\tACDEFGHIKLMNPQRSTVWY
P6\textra\tYWVTSRQPNMLKIHGFEDCA\t3.3.3.3\tignored
"""


def test_tsv_rules_match_the_reference(klib, oracle):
    ref = _makedb_ref().run_tsv(TSV)
    assert [(r[0], r[1]) for r in ref] == [(0, b"P1"), (1, b"P4"), (2, b"P5"), (3, b"P6")]   # 0-based over ACCEPTED rows
    p = api.Proteins.from_tsv(TSV)
    buf, offs = p.packed
    got = [(int(i), bytes(buf[int(offs[j]):int(offs[j + 1])])) for j, i in enumerate(p.ids)]
    assert got == [(r[0], r[2]) for r in ref]
    assert got[1][1] == b"mktayiakqrqistfvk"            # no ToUpper on the TSV path: its k-mers encode to 0s
    assert p.feature_names == [b"Organism", b"EC"]
    _image_pairs(p.image(), _pairs_of(ref, oracle))
    ents = p.fetch_hits([r[0] for r in ref])
    for e, r in zip(ents, ref):
        assert (e["EntryId"], e["Sequence"], e["Length"], e["Features"]) == (r[1], r[2], len(r[2]), r[3])
    assert ents[2]["Features"] == {b"Organism": b"short row", b"EC": b""}
    for bad, msg in ((b"Name\tSequence\nx\tMKTAYIAKQR\n", "EntryID"), (b"entryid\tseq\nx\tMKTAYIAKQR\n", "Sequence"), (b"", "EntryID")):
        with pytest.raises(RuntimeError, match=msg):
            api.Proteins.from_tsv(bad)
        with pytest.raises(ValueError, match=msg):
            _makedb_ref().run_tsv(bad)


def test_random_fasta_against_the_restatement(klib, oracle):
    rng = np.random.default_rng(11)
    aa = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWYacdefgXBZ.*", dtype=np.uint8)
    out = []
    for i in range(600):
        name = [b"kinase", b"hypothetical protein, partial", b"", b"transporter, partial cds", b"x y z"][int(rng.integers(0, 5))]
        out.append(b">E%d%s%s" % (i, b" " if name or i % 3 else b"", name))
        n = int(rng.integers(0, 160))
        seq = bytes(aa[rng.integers(0, len(aa), n)])
        w = int(rng.integers(20, 70))
        out += [seq[j:j + w] for j in range(0, n, w)]
    text = b"\n".join(out) + b"\n"
    ref = _makedb_ref().run_fasta(text)
    p = api.Proteins.from_fasta(text)
    buf, offs = p.packed
    assert p.ids.tolist() == [r[0] for r in ref] and len(ref) > 250
    assert [bytes(buf[int(offs[j]):int(offs[j + 1])]) for j in range(len(p))] == [r[2] for r in ref]
    oix = _pairs_of(ref, oracle)
    whole = p.image()
    _image_pairs(whole, oix)
    # the same proteins built as 4 shards hold the same postings between them
    keys = np.unique((oix.pairs() >> np.uint64(32)).astype(np.uint32))
    shards = [p.image(shard=s, n_shards=4) for s in range(4)]
    assert sum(s.stats()["n_keys"] for s in shards) == len(keys)
    for k in keys[:: max(1, len(keys) // 500)]:
        assert sum(len(s.get(int(k))) for s in shards) == len(whole.get(int(k)))
    last = {r[0]: r for r in ref}
    ents = p.fetch_hits(sorted(last))
    for e, i in zip(ents, sorted(last)):
        assert (e["EntryId"], e["Sequence"], e["Features"][b"ProteinName"]) == (last[i][1], last[i][2], last[i][3][b"ProteinName"])


def test_protein_table_file_round_trip_and_validation(klib, tmp_path):
    p = api.Proteins.from_tsv(TSV)
    path = tmp_path / "proteins.kpt"
    p.save(path)
    q = api.Proteins.load(path)
    assert q.ids.tolist() == p.ids.tolist() and q.feature_names == p.feature_names and q.stats() == p.stats()
    assert q.fetch_hits([0, 1, 2, 3, 4]) == p.fetch_hits([0, 1, 2, 3, 4])
    raw = path.read_bytes()
    for bad in (raw[:-3], raw + b"\0", raw[:40], b"\0" * 64):
        path.write_bytes(bad)
        with pytest.raises(RuntimeError):
            api.Proteins.load(path)
    hdr = bytearray(raw)
    hdr[8:16] = (2 ** 40).to_bytes(8, "little")          # a protein count the file cannot hold
    path.write_bytes(bytes(hdr))
    with pytest.raises(RuntimeError):
        api.Proteins.load(path)


def test_fetch_hits_information_on_query_results(klib):
    """FetchHitsInformation (search.go:454-470) over the QueryResult form the drivers return"""
    from kaamer_amd import search
    p = api.Proteins.from_tsv(TSV)
    qrs = [{"Query": {"Name": "q1"}, "SearchResults": {"Hits": [{"Key": 1, "Kmatch": 11}, {"Key": 0, "Kmatch": 10}]}},
           {"Query": {"Name": "q2"}, "SearchResults": {"Hits": [{"Key": 3, "Kmatch": 14}, {"Key": 77, "Kmatch": 12}, {"Key": 2, "Kmatch": 11}]}},
           {"Query": {"Name": "q3"}, "SearchResults": {"Hits": []}}]
    out = search.FetchHitsInformation(qrs, p)
    assert out[0]["HitEntries"][0] == {"EntryId": "P1", "Sequence": "MKTAYIAKQRQISTFVK", "Length": 17,
                                       "Features": {"Organism": "E. coli", "EC": "1.1.1.1"}}
    assert set(out[0]["HitEntries"]) == {0, 1} and out[0]["HitEntries"][1]["EntryId"] == "P4"
    assert set(out[1]["HitEntries"]) == {3}          # the loop returns at the first id without an entry (search.go:461-463)
    assert out[2]["HitEntries"] == {}


def test_random_tsv_and_nasty_fasta_against_the_restatement(klib):
    """random bytes in every role: CR, empty fields, repeated and missing columns, '>' inside lines, no final newline"""
    rng = np.random.default_rng(23)
    pool = [b"MKTAYIAKQRQISTFVK", b"mktayiak", b"ACDEF", b"", b"ACDEFGHIKLMNPQRSTVWY.*XB", b"A" * 7, b"A" * 6, b"  MKTAYIAKQR  "]
    ids = [b"P1", b"", b"sp|Q|x y", b"dup", b"dup"]
    for trial in range(40):
        ncol = int(rng.integers(2, 6))
        header = [b"EntryID", b"Sequence", b"Organism", b"EC", b"Note"][:ncol]
        if trial % 5 == 1:
            header[0] = b"ENTRYID"
        if trial % 5 == 2:
            header = header[::-1]
        rows = []
        for _ in range(int(rng.integers(0, 25))):
            n = int(rng.integers(1, ncol + 2))                      # short rows and rows with one column too many
            cells = []
            for c in range(n):
                role = header[c].lower() if c < ncol else b""
                if role == b"entryid":
                    cells.append(ids[int(rng.integers(0, len(ids)))])
                elif role == b"sequence":
                    cells.append(pool[int(rng.integers(0, len(pool)))])
                else:
                    cells.append([b"x", b"", b"a b", b"1.1.1.1"][int(rng.integers(0, 4))])
            rows.append(b"\t".join(cells))
        eol = b"\r\n" if trial % 3 == 0 else b"\n"
        text = eol.join([b"\t".join(header)] + rows) + (eol if trial % 2 else b"")
        ref = _makedb_ref().run_tsv(text)
        p = api.Proteins.from_tsv(text)
        buf, offs = p.packed
        got = [(int(i), bytes(buf[int(offs[j]):int(offs[j + 1])])) for j, i in enumerate(p.ids)]
        assert got == [(r[0], r[2]) for r in ref], text
        ents = p.fetch_hits([r[0] for r in ref])
        for e, r in zip(ents, ref):
            assert (e["EntryId"], e["Features"]) == (r[1], r[3]), text
    for trial in range(40):
        lines = []
        for _ in range(int(rng.integers(0, 30))):
            k = int(rng.integers(0, 8))
            lines.append([b">id name, partial", b">id2", b"> leading space", b"MKTAYIAKQRQISTFVK", b"mk>tay", b"ACD", b"X" * 30,
                          b">a b c, partial cds"][k])
        eol = b"\r\n" if trial % 3 == 0 else b"\n"
        text = eol.join(lines) + (eol if trial % 2 else b"")
        ref = _makedb_ref().run_fasta(text)
        p = api.Proteins.from_fasta(text)
        buf, offs = p.packed
        got = [(int(i), bytes(buf[int(offs[j]):int(offs[j + 1])])) for j, i in enumerate(p.ids)]
        assert got == [(r[0], r[2]) for r in ref], text
        last = {r[0]: r for r in ref}
        for e, i in zip(p.fetch_hits(sorted(last)), sorted(last)):
            assert (e["EntryId"], e["Features"][b"ProteinName"]) == (last[i][1], last[i][3][b"ProteinName"]), text


# ---- EMBL / GBK (pkg/makedb/inputEMBL.go, inputGBK.go) -----------------------------------------------------
EMBL = b"""ID   001R_FRG3G              Reviewed;         20 AA.
AC   Q6GZX4;
DE   RecName: Full=Putative transcription factor 001R {ECO:0000305};
DE   SubName: Full=Second name {ECO:1}; extra {ECO:2};
GN   Name=tf1; ORFNames=FV3-001R;
OS   Frog virus 3
OS   (isolate Goorha).
OC   Viruses; Varidnaviria;
OC   Bamfordvirae.
OX   NCBI_TaxID=654924;
DR   GO; GO:0046782; P:regulation of viral transcription; IEA:InterPro.
DR   GO; GO:0000001; F:x.
DR   KEGG; vg:2947773; -.
DR   EMBL; AY548484; AAT09660.1; -; Genomic_DNA.
SQ   SEQUENCE   20 AA;  29735 MW;  B4840739BF7D4121 CRC64;
     MAFSAEDVLK EYDRRRRMEA
//
//
ID   FRAG_X   Unreviewed;  30 AA.
DE   SubName: Full=Some fragment;
DE   Flags: Fragment;
SQ   SEQUENCE   30 AA;  1 MW;  0 CRC64;
     MAFSAEDVLK EYDRRRRMEA MAFSAEDVLK
//
ID   SHORT_Y   Unreviewed;  6 AA.
SQ   SEQUENCE   6 AA;  1 MW;  0 CRC64;
     MAFSAE
//
ID   lower_z   Unreviewed;  12 AA.
DE   RecName: Full=lower case stays, partial;
DE            EC=1.1.1.1 {ECO:3};
SQ   SEQUENCE   12 AA;  1 MW;  0 CRC64;
     mafsaedvlk ey
     extra residues beyond the declared length are not indexed
//
ID   NOT_TERMINATED   Unreviewed;  20 AA.
SQ   SEQUENCE   20 AA;  1 MW;  0 CRC64;
     MAFSAEDVLK EYDRRRRMEA
"""

GBK = b"""LOCUS       WP_000000001             20 aa            linear   BCT 01-JAN-2020
DEFINITION  MULTISPECIES: hypothetical protein
            [Escherichia coli].
ACCESSION   WP_000000001
VERSION     WP_000000001.1
KEYWORDS    RefSeq.
SOURCE      Escherichia coli
  ORGANISM  Escherichia coli
            Bacteria; Proteobacteria;
            Gammaproteobacteria.
FEATURES             Location/Qualifiers
     source          1..20
ORIGIN      
        1 mafsaedvlk eydrrrrmea
//
LOCUS       WP_2                     20 aa
DEFINITION  some enzyme, partial [Bacillus].
VERSION     WP_2.1
ORIGIN      
        1 mafsaedvlk eydrrrrmea
//
LOCUS       WP_3                     5 aa
DEFINITION  tiny [X].
VERSION     WP_3.1
ORIGIN      
        1 mafsa
//
LOCUS       WP_4                     12 aa
DEFINITION  last one [Y]. trailing
VERSION     WP_4.2
ORIGIN      
        1 acdefghikl mn
//
"""


def _check_flat(p, ref, names):
    buf, offs = p.packed
    got = [(int(i), bytes(buf[int(offs[j]):int(offs[j + 1])])) for j, i in enumerate(p.ids)]
    assert got == [(r[0], r[2]) for r in ref]
    assert p.feature_names == names
    for e, r in zip(p.fetch_hits([r[0] for r in ref]), ref):
        assert e["EntryId"] == r[1]
        assert e["Sequence"] == r[3] and e["Length"] == len(r[2])      # Protein.Sequence is the whole string, Length what was indexed
        assert {k: v for k, v in e["Features"].items() if v != b""} == {k: v for k, v in r[4].items() if v != b""}


def test_embl_rules_match_the_reference(klib, oracle):
    R = _makedb_ref()
    ref = R.run_embl(EMBL)
    # record 1 -> id 1; the empty record uses up id 2; the fragment (3) and the short one (4) are dropped; 5 is kept as
    # it is (no upper-casing, ", partial" is no filter here) and indexed up to its declared length; the unterminated
    # record is never queued
    assert [(r[0], r[1], r[2]) for r in ref] == [(1, b"001R_FRG3G", b"MAFSAEDVLKEYDRRRRMEA"), (5, b"lower_z", b"mafsaedvlkey")]
    assert ref[0][4] == {b"ProteinName": b"Putative transcription factor 001R;;Second name", b"GeneName": b"tf1", b"Organism": b"Frog virus 3 (isolate Goorha)",
                         b"FullTaxonomy": b"Viruses; Varidnaviria; Bamfordvirae.", b"TaxId": b"54924", b"GO": b"GO:0046782;GO:0000001",
                         b"KEGG_ID": b"vg:2947773"}
    assert ref[1][4] == {b"ProteinName": b"lower case stays, partial", b"EC": b"1.1.1.1"}
    p = api.Proteins.from_embl(EMBL)
    _check_flat(p, ref, R.EMBL_DEF_FTS)
    st = p.stats()
    assert (st["NumberOfProteins"], st["NumberOfAA"], st["NumberOfKmers"]) == (2, 32, 20)
    img = p.image()
    _image_pairs(img, _pairs_of(ref, oracle))
    assert img.get(oracle.encode_kmer("mafsaed")).tolist() == [5]      # lower case: every pair is a map miss -> key 0 region
    assert img.get(oracle.encode_kmer("MAFSAED")).tolist() == [1]


def test_gbk_rules_match_the_reference(klib, oracle):
    R = _makedb_ref()
    ref = R.run_gbk(GBK)
    assert [(r[0], r[1], r[2]) for r in ref] == [(1, b"WP_000000001.1", b"MAFSAEDVLKEYDRRRRMEA"), (4, b"WP_4.2", b"ACDEFGHIKLMN")]
    assert ref[0][4] == {b"ProteinName": b"MULTISPECIES: hypothetical protein", b"Organism": b"Escherichia coli",
                         b"FullTaxonomy": b"Bacteria; Proteobacteria; Gammaproteobacteria."}
    assert ref[1][4][b"ProteinName"] == b"last one trailing"
    p = api.Proteins.from_gbk(GBK)
    _check_flat(p, ref, R.GBK_DEF_FTS)
    _image_pairs(p.image(), _pairs_of(ref, oracle))


def test_random_embl_and_gbk_against_the_restatement(klib):
    """records assembled at random from well-formed and odd lines (repeated tags, tags in the wrong place, CR line ends,
    empty records, no terminator); lines on which the Go would panic are only drawn into entries of their own, where
    restatement (RefPanic) and product (entry dropped) are compared as such"""
    R = _makedb_ref()
    rng = np.random.default_rng(29)
    embl_lines = [b"ID   P%d_X   Reviewed;  %d AA.", b"DE   RecName: Full=Name %d {ECO:1};", b"DE   SubName: Full=Sub %d;;",
                  b"DE            EC=2.7.%d.1;", b"DE   Flags: Precursor;", b"GN   Name=g%d;", b"GN   ORFNames=o%d;", b"OS   Org %d.",
                  b"OC   Tax%d; More.", b"OX   NCBI_TaxID=12345%d;", b"DR   GO; GO:%d; F:x.", b"DR   KEGG; k:%d; -.", b"DR   BioCyc; B:%d; -.",
                  b"DR   HAMAP; MF_%d; x.", b"DR   Pfam; PF%d; x.", b"XX", b"X", b"", b"CC   free text %d", b"     MAFSAEDVLK EYDRRRRMEA",
                  b"     mktay iakqr", b"     A", b"SQ   SEQUENCE   %d AA;  1 MW;  0 CRC64;"]
    embl_panics = [b"ID", b"DE  RecName: short", b"OX   x", b"DR   ", b"SQ   SEQUENCE", b"GN   Name=", b"SQ   SEQUENCE   999 AA;"]
    gbk_lines = [b"LOCUS       L%d   10 aa", b"DEFINITION  protein %d [Org]. tail", b"DEFINITION  enzyme, partial [Org].", b"            continued [Other].",
                 b"ACCESSION   A%d", b"VERSION     V%d.1  GI:1", b"KEYWORDS    .", b"SOURCE      src", b"  ORGANISM  Org %d", b"            Bacteria; X%d.",
                 b"COMMENT     c", b"FEATURES             Location/Qualifiers", b"     source          1..%d", b"ORIGIN      ", b"ORIGIN", b"        1 mafsaedvlk eydrrrrmea",
                 b"       21 acdefghikl", b"        1 mktay", b"REFERENCE   1", b"DBSOURCE    x", b"X", b""]
    gbk_panics = [b"DEFINITION", b"VERSION", b"VERSION     ", b"  ORGANISM"]
    for kind, lines, panics, run, make in ((b"embl", embl_lines, embl_panics, R.run_embl, api.Proteins.from_embl),
                                           (b"gbk", gbk_lines, gbk_panics, R.run_gbk, api.Proteins.from_gbk)):
        n_kept = 0
        for trial in range(60):
            recs = []
            expect_drop = set()
            for r in range(int(rng.integers(0, 8))):
                body = []
                for _ in range(int(rng.integers(0, 12))):
                    t = lines[int(rng.integers(0, len(lines)))]
                    body.append(t.replace(b"%d", b"%d" % int(rng.integers(5, 40))))
                if rng.random() < 0.5:   # a well-formed end, so that a good share of the records is kept
                    if kind == b"embl":
                        body += [b"SQ   SEQUENCE   %d AA;  1 MW;  0 CRC64;" % int(rng.integers(5, 21)), b"     MAFSAEDVLK EYDRRRRMEA"]
                    else:
                        body += [b"VERSION     W%d.1" % r, b"ORIGIN      ", b"        1 mafsaedvlk eydrrrrmea"]
                if trial % 4 == 3 and rng.random() < 0.3:
                    body.insert(int(rng.integers(0, len(body) + 1)), panics[int(rng.integers(0, len(panics)))])
                recs.append(body)
            eol = b"\r\n" if trial % 3 == 0 else b"\n"
            text = b"".join(eol.join(b + [b"//"]) + eol for b in recs)
            if trial % 5 == 4:
                text += eol.join(lines[:3])          # an unterminated tail
            # the restatement entry by entry, so that a RefPanic drops one entry as the product does
            ref = []
            for pid, entry in R._entries(text):
                try:
                    got = (R.process_embl if kind == b"embl" else R.process_gbk)(entry)
                except R.RefPanic:
                    got = None
                if got is not None:
                    entry_id, seq, length, feat = got
                    ref.append((pid, entry_id, seq[:length], seq, feat))
            p = make(text)
            buf, offs = p.packed
            have = [(int(i), bytes(buf[int(offs[j]):int(offs[j + 1])])) for j, i in enumerate(p.ids)]
            assert have == [(r[0], r[2]) for r in ref], text
            for e, r in zip(p.fetch_hits([r[0] for r in ref]), ref):
                assert e["EntryId"] == r[1], text
                assert e["Sequence"] == r[3] and e["Length"] == len(r[2]), text
                assert {k: v for k, v in e["Features"].items() if v != b""} == {k: v for k, v in r[4].items() if v != b""}, text
            n_kept += len(ref)
        assert n_kept > 20, kind


def test_strict_scanner_limit_of_the_database_readers(klib):
    """inputFASTA.go:88-89 (and the same lines of the other readers): bufio.Scanner with a 1 MiB buffer ends the scan at a
    line of 1 048 576 bytes or more; kaamer_makedb_text(strict_scanner = 1) reproduces it, the default reads on."""
    import ctypes as C

    def make(text, fmt, strict):
        h = C.c_void_p()
        abi.check(klib.kaamer_makedb_text(text, len(text), fmt, strict, C.byref(h)))
        return api.Proteins(h.value)
    MAX = 1024 * 1024
    head = b">a one\nMKTAYIAKQRQISFVK\n>b two\nMKTAYIAKQR"
    tail = b"\n>c three\nACDEFGHIKLMNPQRS\n>d four\nACDEFGHIKLMNPQRT\n"
    for n, cut in ((MAX - 1, False), (MAX, True)):
        text = head + b"\n" + b"K" * n + tail
        loose, strict = make(text, 0, 0), make(text, 0, 1)
        same = api.Proteins.from_fasta(text)
        assert len(loose) == len(same) == 4 and loose.ids.tolist() == same.ids.tolist()
        if not cut:
            assert len(strict) == 4 and bytes(strict.packed[0]) == bytes(loose.packed[0])
        else:   # the scan ends inside record b: a and what b had so far, b as the file's last record
            assert len(strict) == 2
            buf, offs = strict.packed
            assert bytes(buf[int(offs[1]):int(offs[2])]) == b"MKTAYIAKQR"
    # TSV: rows behind the long line are never seen
    tsv = b"EntryID\tSequence\np1\tMKTAYIAKQRQISFVK\np2\t" + b"K" * MAX + b"\np3\tACDEFGHIKLMNPQRS\n"
    assert len(make(tsv, 1, 0)) == 3 and len(make(tsv, 1, 1)) == 1
    with pytest.raises(abi.KaamerError):
        make(b">a\nMKTAYIAK\n", 7, 0)


def test_embl_entry_with_more_residues_than_declared_keeps_them(klib, tmp_path):
    """inputEMBL.go:293-312: the k-mers are those of Sequence[:Length], the stored Protein.Sequence is the whole string
    (the table used to cut it); the protein-table file carries the whole string too"""
    text = EMBL.replace(b"SEQUENCE   20 AA;", b"SEQUENCE   12 AA;", 1)
    assert text != EMBL
    R = _makedb_ref()
    ref = R.run_embl(text)
    cut = [r for r in ref if len(r[3]) > len(r[2])]
    assert cut, "the fixture must hold an entry that declares fewer residues than it has"
    p = api.Proteins.from_embl(text)
    _check_flat(p, ref, R.EMBL_DEF_FTS)
    f = tmp_path / "prot.kpt"
    p.save(f)
    q = api.Proteins.load(f)
    _check_flat(q, ref, R.EMBL_DEF_FTS)
    blob = f.read_bytes()
    for bad in (blob[:-3], blob + b"\0" * 5):
        g = tmp_path / "bad.kpt"
        g.write_bytes(bad)
        with pytest.raises(abi.KaamerError):
            api.Proteins.load(g)
