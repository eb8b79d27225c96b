"""End-to-end drop-in check: the reference's drivers (ProteinSearch / FastqSearch,
search_protein.go, search_fastq.go) mirrored over the C ABI against the same flow run on
the CPU restatements (tests/pyref.py readers + oracle search/filter/start-codon)."""
import numpy as np
import pytest

import pyref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env(klib, oracle, gpu_device):
    from kaamer_amd import api, workload
    db = workload.make_db(800, seed=21)
    ix = api.Index.from_image(api.Image.from_proteins(packed=db), gpu_device)
    return db, ix, oracle.Index.from_proteins(None, packed=db)


def _ref_hits(oix, seq, size, opts):
    pid, km, pos = oix.search(seq, size=size, want_positions=True)
    return pid, km, pos


def test_protein_search_flow(env, oracle):
    from kaamer_amd import search, workload
    db, ix, oix = env
    qs = workload.unpack(workload.make_protein_queries(db, 80, seed=5))
    text = "".join(">q%d some description\n%s\n" % (i, s.decode()) for i, s in enumerate(qs))
    text += ">short\nACDEFGHIKLMN\n>lower\n" + qs[0].decode().lower() + "\n"     # SizeInKmer < 7; last record keeps its case
    for opts in (search.SearchOptions(), search.SearchOptions(MaxResults=3, MinKMatch=1, MinKRatio=0.0),
                 search.SearchOptions(MinKMatch=50, MinKRatio=0.5),
                 search.SearchOptions(ExtractPositions=True)):  # the last one takes the host post-steps
        got = search.ProteinSearch(ix, text, opts)
        exp = []
        for q in pyref.get_queries_fasta(text):
            if q["size"] < 7:
                continue
            pid, km, _ = oix.search(q["seq"], size=q["size"])
            keep = oracle.filter_results(km, q["size"], opts.MinKRatio, opts.MinKMatch, opts.MaxResults)
            if keep:
                exp.append((q["name"], q["size"], km[:keep].tolist(), dict(zip(pid.tolist(), km.tolist()))))
        assert len(got) == len(exp)
        for g, (name, size, kms, full) in zip(got, exp):
            assert g["Query"]["Name"] == name and g["Query"]["SizeInKmer"] == size
            hits = g["SearchResults"]["Hits"]
            # the Kmatch sequence is fixed; among ties any member of the reference's set is valid
            assert [h["Kmatch"] for h in hits] == kms
            assert all(full[h["Key"]] == h["Kmatch"] for h in hits)
    assert len(search.ProteinSearch(ix, text)) > 50


def test_fastq_search_flow(env, oracle):
    from kaamer_amd import search, workload
    db, ix, oix = env
    reads = workload.unpack(workload.make_reads(db, 300, seed=12))
    text = "".join("@read%d\n%s\n+\n%s\n" % (i, r.decode(), "I" * len(r)) for i, r in enumerate(reads))
    opts = search.SearchOptions(SequenceType=2)
    got = search.FastqSearch(ix, text, opts)                                         # device post-steps
    got_host = search.FastqSearch(ix, text, search.SearchOptions(SequenceType=2, ExtractPositions=True))  # host post-steps
    assert [(g["Query"], [h["Kmatch"] for h in g["SearchResults"]["Hits"]]) for g in got] == \
        [(g["Query"], [h["Kmatch"] for h in g["SearchResults"]["Hits"]]) for g in got_host]
    exp = []
    for rec in pyref.get_queries_fastq(text):
        for o in oracle.get_orfs(rec["seq"]):
            size = oracle.size_in_kmer(o["seq"])
            pid, km, pos = oix.search(o["seq"], size=size, want_positions=True)
            if len(km) == 0 or km[0] < opts.MinKMatch:
                continue
            t, sp, so = oracle.set_best_start_codon(km, pos, size, o["starts"], o["plus"], o["seq"], o["start"])
            keep = oracle.filter_results(km, so, opts.MinKRatio, opts.MinKMatch, opts.MaxResults)
            if keep:
                exp.append((rec["name"], o["seq"][t:], so, sp, o["end"], o["plus"], km[:keep].tolist(),
                            dict(zip(pid.tolist(), km.tolist()))))
    assert len(exp) > 100 and len(got) == len(exp)
    for g, (name, seq, size, start, end, plus, kms, full) in zip(got, exp):
        q = g["Query"]
        assert (q["Name"], q["Sequence"], q["SizeInKmer"]) == (name, seq, size)
        loc = q["Location"]
        assert (loc["StartPosition"], loc["EndPosition"], loc["PlusStrand"]) == (start, end, plus)
        hits = g["SearchResults"]["Hits"]
        assert [h["Kmatch"] for h in hits] == kms and all(full[h["Key"]] == h["Kmatch"] for h in hits)


def _dev(ptr, n, dtype):
    from test_gpu_protein import _from_ptr
    return _from_ptr(ptr, n, dtype)


def test_device_topn_protein(env, oracle):
    """kaamer_topn_device on a protein batch == oracle search + sortMapByValue order + FilterResults"""
    import torch
    from kaamer_amd import api, workload
    db, ix, oix = env
    q = workload.make_protein_queries(db, 120, seed=9)
    seqs = workload.unpack(q) + [b"ACDEFGHIKLMN", b"", db_first(db) * 3]   # too short, empty, > 512 hits unlikely but long
    buf, offs = api.pack_sequences(seqs)
    d_buf = torch.from_numpy(buf).cuda()
    d_off = torch.from_numpy(offs.view(np.int64)).cuda()
    st = torch.cuda.current_stream().cuda_stream
    for compact in (False, True):
        ws = api.Workspace(ix, len(buf), len(seqs), first_pos=1, compact=compact)
        ws.search_device(d_buf.data_ptr(), d_off.data_ptr(), len(seqs), len(buf), stream=st)
        for (ratio, mink, maxr) in ((0.05, 10, 10), (0.0, 1, 3), (0.5, 50, 10), (0.0, 1, 700)):
            t = ws.topn_device(ratio, mink, maxr, stream=st)
            ws.finish(st)
            cnt = _dev(t.d_top_cnt, len(seqs), np.uint32)
            pid = _dev(t.d_top_pid, len(seqs) * maxr, np.uint32).reshape(len(seqs), maxr)
            km = _dev(t.d_top_kmatch, len(seqs) * maxr, np.uint32).reshape(len(seqs), maxr)
            fp = _dev(t.d_top_first_pos, len(seqs) * maxr, np.uint32).reshape(len(seqs), maxr)
            size_out = _dev(t.d_size_in_kmer, len(seqs), np.int32)
            for i, s in enumerate(seqs):
                size = oracle.size_in_kmer(s)
                assert size_out[i] == size
                if size < 7:
                    assert cnt[i] == 0
                    continue
                epid, ekm, epos = oix.search(s, size=size, want_positions=True)
                keep = oracle.filter_results(ekm, size, ratio, mink, maxr)
                assert cnt[i] == keep, (i, ratio, mink, maxr)
                assert pid[i, :keep].tolist() == epid[:keep].tolist()
                assert km[i, :keep].tolist() == ekm[:keep].tolist()
                assert fp[i, :keep].tolist() == [int(np.argmax(epos[h])) for h in range(keep)]


def db_first(db):
    from kaamer_amd import workload
    return workload.unpack((db[0][:int(db[1][1])], db[1][:2]))[0]


def test_device_topn_reads(env, oracle):
    """reads: SetBestStartCodon + gate + FilterResults on the device == the oracle's literal flow"""
    import torch
    from kaamer_amd import abi, api, workload
    db, ix, oix = env
    reads = workload.make_reads(db, 400, seed=31)
    buf, offs = reads
    d_buf = torch.from_numpy(buf).cuda()
    d_off = torch.from_numpy(offs.view(np.int64)).cuda()
    st = torch.cuda.current_stream().cuda_stream
    ws = api.Workspace(ix, len(buf), 400, seq_type=abi.READS)
    r = ws.search_device(d_buf.data_ptr(), d_off.data_ptr(), 400, len(buf), stream=st)
    n_trim = n_rep = 0
    for (ratio, mink, maxr) in ((0.05, 10, 10), (0.0, 1, 2)):
        t = ws.topn_device(ratio, mink, maxr, best_start_codon=True, stream=st)
        c = ws.finish(st)
        nq = c["n_queries"]
        cnt = _dev(t.d_top_cnt, nq, np.uint32)
        pid = _dev(t.d_top_pid, nq * maxr, np.uint32).reshape(nq, maxr)
        km = _dev(t.d_top_kmatch, nq * maxr, np.uint32).reshape(nq, maxr)
        trim = _dev(t.d_trim, nq, np.int32)
        sp = _dev(t.d_start_position, nq, np.int32)
        so = _dev(t.d_size_in_kmer, nq, np.int32)
        qi = 0
        for read in workload.unpack(reads):
            for o in oracle.get_orfs(read):
                size = oracle.size_in_kmer(o["seq"])
                epid, ekm, epos = oix.search(o["seq"], size=size, want_positions=True)
                if len(ekm) == 0 or ekm[0] < mink:
                    assert cnt[qi] == 0
                else:
                    et, esp, eso = oracle.set_best_start_codon(ekm, epos, size, o["starts"], o["plus"], o["seq"], o["start"])
                    keep = oracle.filter_results(ekm, eso, ratio, mink, maxr)
                    assert (trim[qi], sp[qi], so[qi]) == (et, esp, eso), qi
                    assert cnt[qi] == keep
                    assert pid[qi, :keep].tolist() == epid[:keep].tolist() and km[qi, :keep].tolist() == ekm[:keep].tolist()
                    n_trim += et > 0
                    n_rep += keep > 0
                qi += 1
        assert qi == nq
    assert n_rep > 100
    # SetBestStartCodon needs ORFs: refused on a protein workspace
    pws = api.Workspace(ix, 1024, 4)
    with pytest.raises(abi.KaamerError):
        pws.topn_device(best_start_codon=True, stream=st)


def test_device_topn_long_lists(klib, oracle, gpu_device):
    """> 512 hits per query (the kernel re-scans instead of caching) with massive Kmatch ties:
    the order is (Kmatch desc, protein id asc) and the prefix rule holds"""
    import torch
    from kaamer_amd import api
    rng = np.random.default_rng(8)
    alpha = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)
    core = bytes(alpha[rng.integers(0, 20, 60)])
    db = [bytes(alpha[rng.integers(0, 20, 5)]) + core[(i % 7):] + bytes(alpha[rng.integers(0, 20, 5)]) for i in range(900)]
    ids = rng.permutation(5000)[:900].astype(np.uint32)
    ix = api.Index.from_image(api.Image.from_proteins(db, ids=ids), gpu_device)
    oix = oracle.Index.from_proteins(db, ids=ids)
    seqs = [core, core[:30], db[5]]
    buf, offs = api.pack_sequences(seqs)
    d_buf = torch.from_numpy(buf).cuda()
    d_off = torch.from_numpy(offs.view(np.int64)).cuda()
    st = torch.cuda.current_stream().cuda_stream
    ws = api.Workspace(ix, len(buf), len(seqs), first_pos=1)
    ws.search_device(d_buf.data_ptr(), d_off.data_ptr(), len(seqs), len(buf), stream=st)
    for (ratio, mink, maxr) in ((0.0, 1, 900), (0.9, 1, 900), (0.05, 10, 10)):
        t = ws.topn_device(ratio, mink, maxr, stream=st)
        ws.finish(st)
        cnt = _dev(t.d_top_cnt, len(seqs), np.uint32)
        pid = _dev(t.d_top_pid, len(seqs) * maxr, np.uint32).reshape(len(seqs), maxr)
        km = _dev(t.d_top_kmatch, len(seqs) * maxr, np.uint32).reshape(len(seqs), maxr)
        for i, s in enumerate(seqs):
            size = oracle.size_in_kmer(s)
            epid, ekm, _ = oix.search(s, size=size)
            if i == 0:
                assert len(epid) == 900
            keep = oracle.filter_results(ekm, size, ratio, mink, maxr)
            assert cnt[i] == keep
            assert pid[i, :keep].tolist() == epid[:keep].tolist() and km[i, :keep].tolist() == ekm[:keep].tolist()


def test_makedb_to_hit_entries_end_to_end(klib, oracle, gpu_device):
    """The whole drop-in flow over the C ABI: database FASTA text -> kaamer_makedb_fasta (the reference's id rule,
    inputFASTA.go:95-124: ids run on from the second record's, the last two records share one) -> the table built ON THE DEVICE
    (kaamer_image_build_makedb_device) -> ProteinSearch -> FetchHitsInformation (search.go:454-470): every reported
    Key resolves to the database record the reference's id rule gave that id, and the counts are the oracle's."""
    from kaamer_amd import api, search, workload
    db = workload.make_db(600, seed=77)
    recs = workload.unpack(db)
    fasta = "".join(">sp|P%05d|NAME_%d OS=Test organism GN=g%d\n%s\n" % (i, i, i, s.decode()) for i, s in enumerate(recs))
    prot = api.Proteins.from_fasta(fasta.encode())
    ids = np.asarray(prot.ids)
    assert len(ids) == len(recs) and (np.diff(ids[:-1]) == 1).all() and ids[-1] == ids[-2]   # the id rule and its last-record quirk
    img = prot.image(device=gpu_device)
    assert img.stats() == prot.image().stats()                                           # device builder == host builder
    ix = api.Index.from_image(img, gpu_device)
    qs = workload.unpack(workload.make_protein_queries(db, 60, seed=78))
    qtext = "".join(">query%d\n%s\n" % (i, s.decode()) for i, s in enumerate(qs))
    res = search.FetchHitsInformation(search.ProteinSearch(ix, qtext, search.SearchOptions(MaxResults=5)), prot)
    assert len(res) > 40
    # the oracle over the same (sequence, id) pairs
    oix = oracle.Index.from_proteins(recs, ids=ids.tolist())
    by_id = {}
    for s, i in zip(recs, ids.tolist()):
        by_id.setdefault(i, s)          # two records under the last id: the table keeps the first (makedb.cpp)
    for qr in res:
        q = next(s for i, s in enumerate(qs) if "query%d" % i == qr["Query"]["Name"])
        pid, km, _ = oix.search(q)
        full = dict(zip(pid.tolist(), km.tolist()))
        hits = qr["SearchResults"]["Hits"]
        assert [h["Kmatch"] for h in hits] == km[:len(hits)].tolist()
        assert set(qr["HitEntries"]) == {h["Key"] for h in hits}
        for h in hits:
            assert full[h["Key"]] == h["Kmatch"]
            e = qr["HitEntries"][h["Key"]]
            assert e["Length"] == len(e["Sequence"]) and e["EntryId"].startswith("sp|P")
            if h["Key"] != int(ids[-1]):    # (the shared last id has two candidate records)
                assert e["Sequence"].encode() == by_id[h["Key"]]


def _embl_text(recs, rng):
    """UniProt flat-file records around the given sequences (inputEMBL.go:95-113,189-314 read these tags)"""
    out = []
    for i, s in enumerate(recs):
        seq = s.decode()
        lines = ["ID   P%05d_TEST            Reviewed;       %d AA." % (i, len(seq)),
                 "AC   P%05d;" % i,
                 "DE   RecName: Full=Protein number %d {ECO:0000255|HAMAP-Rule:MF_%05d};" % (i, i),
                 "DE            EC=2.7.%d.1;" % (i % 9 + 1),
                 "GN   Name=gene%d;" % i,
                 "OS   Testus organismus (strain %d)." % (i % 7),
                 "OC   Bacteria; Proteobacteria; Gammaproteobacteria.",
                 "OX   NCBI_TaxID=%d;" % (100000 + i),
                 "DR   GO; GO:%07d; F:binding; IEA:UniProtKB-KW." % i,
                 "DR   KEGG; tst:T%05d; -." % i,
                 "SQ   SEQUENCE   %d AA;  %d MW;  0000000000000000 CRC64;" % (len(seq), 110 * len(seq))]
        if rng.random() < 0.05:
            lines.insert(3, "DE   Flags: Fragment;")         # dropped by the reference (inputEMBL.go:230-233)
        for k in range(0, len(seq), 60):
            row = seq[k:k + 60]
            lines.append("     " + " ".join(row[j:j + 10] for j in range(0, len(row), 10)))
        lines.append("//")
        out.append("\n".join(lines) + "\n")
    return "".join(out).encode()


def _gbk_text(recs, rng):
    """GenPept records (inputGBK.go:94-112,186-301)"""
    out = []
    for i, s in enumerate(recs):
        seq = s.decode().lower()
        name = "hypothetical protein %d" % i + (", partial" if rng.random() < 0.05 else "")   # ", partial" entries are dropped
        lines = ["LOCUS       WP_%09d           %d aa            linear   BCT 01-JAN-2020" % (i, len(seq)),
                 "DEFINITION  %s [Testus organismus]." % name,
                 "ACCESSION   WP_%09d" % i,
                 "VERSION     WP_%09d.1" % i,
                 "KEYWORDS    RefSeq.",
                 "SOURCE      Testus organismus",
                 "  ORGANISM  Testus organismus",
                 "            Bacteria; Proteobacteria.",
                 "FEATURES             Location/Qualifiers",
                 "     source          1..%d" % len(seq),
                 "ORIGIN      "]
        for k in range(0, len(seq), 60):
            row = seq[k:k + 60]
            lines.append("%9d %s" % (k + 1, " ".join(row[j:j + 10] for j in range(0, len(row), 10))))
        lines.append("//")
        out.append("\n".join(lines) + "\n")
    return "".join(out).encode()


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", ["embl", "gbk"])
def test_flat_file_database_to_hit_entries_end_to_end(klib, oracle, gpu_device, fmt):
    """SURVEY 8f row 3 on the GPU: UniProt (EMBL) / GenPept (GBK) flat-file text -> kaamer_makedb_embl / _gbk (record-ordinal
    ids, Fragment / ", partial" filters, the declared-length rule: inputEMBL.go:95-113,189-314, inputGBK.go:94-112,
    186-301) -> the table built ON THE DEVICE (kaamer_image_build_makedb_device) -> ProteinSearch -> kaamer_fetch_hits.
    Ids, indexed residues and entries against the line-by-line restatement (oracle/makedb_ref.py); counts against the
    oracle's index over the restatement's (sequence, id) pairs; once more from the gzipped file (inputEMBL.go:76-84)."""
    import gzip
    import importlib.util
    import os
    from kaamer_amd import api, search, workload
    spec = importlib.util.spec_from_file_location("makedb_ref", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "makedb_ref.py"))
    R = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(R)
    rng = np.random.default_rng(91)
    db = workload.make_db(500, seed=79)
    recs = workload.unpack(db)
    text = (_embl_text if fmt == "embl" else _gbk_text)(recs, rng)
    ref = (R.run_embl if fmt == "embl" else R.run_gbk)(text)
    assert 400 < len(ref) < len(recs)                      # the filters dropped some records; ids keep the record ordinals
    make = api.Proteins.from_embl if fmt == "embl" else api.Proteins.from_gbk
    prot = make(text)
    buf, offs = prot.packed
    assert [int(i) for i in prot.ids] == [r[0] for r in ref]
    assert [bytes(buf[int(offs[j]):int(offs[j + 1])]) for j in range(len(ref))] == [r[2] for r in ref]
    gz = make(gzip.compress(text))
    assert gz.ids.tolist() == prot.ids.tolist() and bytes(gz.packed[0]) == bytes(buf)
    img = prot.image(device=gpu_device)
    assert img.stats() == prot.image().stats()             # device builder == host builder
    ix = api.Index.from_image(img, gpu_device)
    oix = oracle.Index.from_proteins([r[2] for r in ref], ids=[r[0] for r in ref])
    qs = workload.unpack(workload.make_protein_queries(db, 80, seed=80))
    qtext = "".join(">query%d\n%s\n" % (i, s.decode()) for i, s in enumerate(qs))
    res = search.FetchHitsInformation(search.ProteinSearch(ix, qtext, search.SearchOptions(MaxResults=5)), prot)
    assert len(res) > 50
    by_id = {r[0]: r for r in ref}
    n_hits = 0
    for qr in res:
        q = next(s for i, s in enumerate(qs) if "query%d" % i == qr["Query"]["Name"])
        pid, km, _ = oix.search(q)
        full = dict(zip(pid.tolist(), km.tolist()))
        hits = qr["SearchResults"]["Hits"]
        assert [h["Kmatch"] for h in hits] == km[:len(hits)].tolist()
        for h in hits:
            assert full[h["Key"]] == h["Kmatch"]
            e, r = qr["HitEntries"][h["Key"]], by_id[h["Key"]]
            assert e["EntryId"].encode() == r[1] and e["Sequence"].encode() == r[2]
            assert {k: v for k, v in e["Features"].items() if v} == {k.decode(): v.decode() for k, v in r[4].items() if v}
            n_hits += 1
    assert n_hits > 100
