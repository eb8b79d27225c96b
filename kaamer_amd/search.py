"""Host-side mirror of the reference's search drivers (pkg/search/search_protein.go,
search_fastq.go, search_nucleotide.go) over the C ABI: same option names and defaults
(api/server.go:139-152), same per-query flow, results as plain dicts shaped like the
reference's JSON (docs/client.md:131-180).  All compute is in libkaamer_hip.so: readers and
k-mer search; sortMapByValue, SetBestStartCodon and FilterResults run on the device
(kaamer_search_batch_top) unless ExtractPositions asks for the full hit lists + bitmaps, in
which case the host versions of the same C ABI are used.
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import abi, api

PROTEIN_QUERY, DNA_QUERY = "Protein Query", "DNA Query"  # search.go:46-47


@dataclass
class SearchOptions:  # search.go:56-71 (the fields the hot path reads); defaults api/server.go:139-152
    SequenceType: int = abi.PROTEIN
    MaxResults: int = 10
    MinKMatch: int = 10
    MinKRatio: float = 0.05
    ExtractPositions: bool = False
    Align: bool = False           # search.go:65; the alignment step: AlignHits below
    SubMatrix: str = "blosum62"   # api/server.go:149-151
    GapOpen: int = 11
    GapExtend: int = 1


def _sorted_hits(res, q):
    """sortMapByValue (search.go:132-152): Kmatch descending (ties by protein id)"""
    a, b = res.span(q)
    pid = np.ascontiguousarray(res.hit_pid[a:b])
    km = np.ascontiguousarray(res.hit_kmatch[a:b])
    order = np.zeros(b - a, dtype=np.uint32)
    abi.lib().kaamer_sort_hits(pid.ctypes.data, km.ctypes.data, b - a, order.ctypes.data)
    return pid[order], km[order], np.ascontiguousarray(res.hit_first_pos[a:b])[order], order + a


def _filter(km_sorted, size, o):
    """FilterResults (search.go:189-220) -> number of hits kept"""
    km = np.ascontiguousarray(km_sorted, dtype=np.uint32)
    return int(abi.lib().kaamer_filter_results(km.ctypes.data, len(km), size, o.MinKRatio, o.MinKMatch, o.MaxResults))


def ProteinSearch(index, fasta_text, options=None):
    """search_protein.go:27-134 for one FASTA upload -> list of QueryResult dicts"""
    o = options or SearchOptions(SequenceType=abi.PROTEIN)
    queries = api.parse_reads(fasta_text, "fasta")
    if not o.ExtractPositions:
        # sortMapByValue + FilterResults on the device: only the reported hits come back
        top = index.search_top([q["seq"] for q in queries], seq_type=abi.PROTEIN, min_k_ratio=o.MinKRatio,
                               min_k_match=o.MinKMatch, max_results=o.MaxResults)
        out = []
        for r in range(top.n_reported):   # search_protein.go:74-76, :108: the others are not reported
            q = queries[int(top.rep_query[r])]
            a, b = int(top.top_off[r]), int(top.top_off[r + 1])
            out.append({"Query": {"Sequence": q["seq"], "Name": q["name"], "SizeInKmer": q["size"], "Type": PROTEIN_QUERY,
                                  "Location": {"StartPosition": 1, "EndPosition": len(q["seq"]), "PlusStrand": q["plus"],
                                               "StartsAlternative": []}, "Contig": ""},
                        "SearchResults": {"Hits": [{"Key": int(p), "Kmatch": int(k)}
                                                   for p, k in zip(top.top_pid[a:b], top.top_kmatch[a:b])]}})
        return out
    res = index.search([q["seq"] for q in queries], seq_type=abi.PROTEIN, want_positions=o.ExtractPositions)
    out = []
    for i, q in enumerate(queries):
        if q["size"] < 7:  # search_protein.go:74-76
            continue
        pid, km, _, idx = _sorted_hits(res, i)
        keep = _filter(km, q["size"], o)
        if keep == 0:  # search_protein.go:108
            continue
        qr = {"Query": {"Sequence": q["seq"], "Name": q["name"], "SizeInKmer": q["size"], "Type": PROTEIN_QUERY,
                        "Location": {"StartPosition": 1, "EndPosition": len(q["seq"]), "PlusStrand": q["plus"],
                                     "StartsAlternative": []}, "Contig": ""},
              "SearchResults": {"Hits": [{"Key": int(p), "Kmatch": int(k)} for p, k in zip(pid[:keep], km[:keep])]}}
        if o.ExtractPositions:
            pos = res.positions(i)
            qr["SearchResults"]["PositionHits"] = {int(p): pos[int(p)].tolist() for p in pid[:keep]}
        out.append(qr)
    return out


def _orf_results(index, reads, names, o, seq_type):
    if not o.ExtractPositions:
        # SetBestStartCodon, its gate and FilterResults on the device (kaamer_search_batch_top)
        top = index.search_top(reads, seq_type=seq_type, min_k_ratio=o.MinKRatio, min_k_match=o.MinKMatch,
                               max_results=o.MaxResults)
        out = []
        for i in range(top.n_reported):
            a, b = int(top.top_off[i]), int(top.top_off[i + 1])
            m = top.meta[i]
            aa = bytes(top.orf_aa[int(m["aa_off"]):int(m["aa_off"]) + int(m["aa_len"])])  # already trimmed
            out.append({"Query": {"Sequence": aa.decode("latin-1"), "Name": names[int(m["src_seq"])],
                                  "SizeInKmer": int(m["size_in_kmer"]), "Type": DNA_QUERY,
                                  "Location": {"StartPosition": int(m["start_position"]), "EndPosition": int(m["end_position"]),
                                               "PlusStrand": bool(m["plus_strand"]), "StartsAlternative": []},
                                  "Contig": ""},
                        "SearchResults": {"Hits": [{"Key": int(p), "Kmatch": int(k)}
                                                   for p, k in zip(top.top_pid[a:b], top.top_kmatch[a:b])]}})
        return out
    res = index.search(reads, seq_type=seq_type, want_positions=o.ExtractPositions)
    out = []
    for i in range(res.n_queries):
        m = res.meta[i]
        pid, km, fp, idx = _sorted_hits(res, i)
        if len(km) == 0 or int(km[0]) < o.MinKMatch:  # search_fastq.go:119
            continue
        aa = bytes(res.orf_aa[int(m["aa_off"]):int(m["aa_off"]) + int(m["aa_len"])])
        sa = np.ascontiguousarray(res.starts_alt[int(m["sa_off"]):int(m["sa_off"]) + int(m["sa_len"])], dtype=np.int32)
        start, size = C.c_int32(int(m["start_position"])), C.c_int32(int(m["size_in_kmer"]))
        kmc, fpc = np.ascontiguousarray(km, dtype=np.uint32), np.ascontiguousarray(fp, dtype=np.uint32)
        trimmed = abi.lib().kaamer_set_best_start_codon(kmc.ctypes.data, fpc.ctypes.data, len(kmc),
                                                        sa.ctypes.data if len(sa) else None, len(sa),
                                                        int(m["plus_strand"]), aa, len(aa), C.byref(start), C.byref(size))
        keep = _filter(km, size.value, o)  # on the possibly shrunk SizeInKmer (dna.go:263-266)
        if keep == 0:
            continue
        out.append({"Query": {"Sequence": aa[trimmed:].decode("latin-1"), "Name": names[int(m["src_seq"])],
                              "SizeInKmer": size.value, "Type": DNA_QUERY,
                              "Location": {"StartPosition": start.value, "EndPosition": int(m["end_position"]),
                                           "PlusStrand": bool(m["plus_strand"]), "StartsAlternative": []},
                              "Contig": ""},
                    "SearchResults": {"Hits": [{"Key": int(p), "Kmatch": int(k)} for p, k in zip(pid[:keep], km[:keep])]}})
    return out


def FastqSearch(index, fastq_text, options=None):
    """search_fastq.go:27-154 (READS) -> list of QueryResult dicts, one per reported ORF"""
    o = options or SearchOptions(SequenceType=abi.READS)
    recs = api.parse_reads(fastq_text, "fastq")
    return _orf_results(index, [r["seq"] for r in recs], [r["name"] for r in recs], o, abi.READS)


def NucleotideSearch(index, fasta_text, options=None):
    """search_nucleotide.go:27-160 (contigs in FASTA) -> list of QueryResult dicts"""
    o = options or SearchOptions(SequenceType=abi.NUCLEOTIDE)
    recs = api.parse_reads(fasta_text, "fasta")
    out = _orf_results(index, [r["seq"] for r in recs], [r["name"] for r in recs], o, abi.NUCLEOTIDE)
    for qr in out:
        qr["Query"]["Contig"] = qr["Query"]["Name"]  # search.go:305-306
    return out


def FetchHitsInformation(query_results, proteins):
    """search.go:454-470 for a list of QueryResult dicts: HitEntries[Key] = the Protein entry (protein.proto:
    EntryId, Sequence, Length, Features) of every reported hit, from the protein table makedb built
    (api.Proteins: kaamer_fetch_hits) instead of one ProteinStore point read per hit.  Like the reference, a
    query's loop stops at the first id that has no entry."""
    keys = sorted({h["Key"] for qr in query_results for h in qr["SearchResults"]["Hits"]})
    entries = dict(zip(keys, proteins.fetch_hits(keys))) if keys else {}
    for qr in query_results:
        he = qr.setdefault("HitEntries", {})
        for h in qr["SearchResults"]["Hits"]:
            if h["Key"] in he:
                continue
            e = entries.get(h["Key"])
            if e is None:
                break                      # search.go:461-463: return on the first failed read
            he[h["Key"]] = {"EntryId": e["EntryId"].decode("latin-1"), "Sequence": e["Sequence"].decode("latin-1"),
                            "Length": e["Length"],
                            "Features": {k.decode("latin-1"): v.decode("latin-1") for k, v in e["Features"].items()}}
    return query_results


def AlignHits(query_results, proteins, opts, device=0):
    """The alignment step of QueryResultHandler (search.go:483-494) for a list of QueryResult dicts that went through
    FetchHitsInformation: align.Align(Query.Sequence, HitEntries[hit.Key].Sequence, dbStats, SubMatrix, GapOpen,
    GapExtend) for every reported hit -- all pairs of the batch in ONE device call (kaamer_align_pairs) -- then the hits
    of each query re-sorted by BitScore, descending.  A pair for which Align returns an error ("No matrix found") keeps
    the empty AlignmentResult sortMapByValue gave it (search.go:144)."""
    seqs, index, pairs, where = [], {}, [], []

    def seq_id(s):
        if s not in index:
            index[s] = len(seqs)
            seqs.append(s)
        return index[s]
    for qi, qr in enumerate(query_results):
        q = qr["Query"]["Sequence"].encode("latin-1")
        for hi, h in enumerate(qr["SearchResults"]["Hits"]):
            e = qr.get("HitEntries", {}).get(h["Key"])
            if e is None:
                continue
            pairs.append((seq_id(q), seq_id(e["Sequence"].encode("latin-1"))))
            where.append((qi, hi))
    empty = {"Identity": 0.0, "Similarity": 0.0, "Length": 0, "Mismatches": 0, "GapOpenings": 0, "Raw": 0, "BitScore": 0.0,
             "EValue": 0.0, "AlnString": "", "QueryStart": 0, "QueryEnd": 0, "SubjectStart": 0, "SubjectEnd": 0}
    for qr in query_results:
        for h in qr["SearchResults"]["Hits"]:
            h["Alignment"] = dict(empty)
    if pairs:
        n_aa = proteins.stats()["NumberOfAA"]   # KStats.NumberOfAA (kvstore.KStats)
        got = api.align_pairs(seqs=seqs, pairs=pairs, number_of_aa=n_aa, sub_matrix=getattr(opts, "SubMatrix", "blosum62"),
                              gap_open=getattr(opts, "GapOpen", 11), gap_extend=getattr(opts, "GapExtend", 1), device=device)
        for (qi, hi), a in zip(where, got):
            if a is None or a.get("status"):
                continue
            query_results[qi]["SearchResults"]["Hits"][hi]["Alignment"] = {
                "Identity": a["identity"], "Similarity": a["similarity"], "Length": a["length"], "Mismatches": a["mismatches"],
                "GapOpenings": a["gap_openings"], "Raw": a["raw"], "BitScore": a["bitscore"], "EValue": a["evalue"],
                "AlnString": "\n".join(a["aln"]), "QueryStart": a["query_start"], "QueryEnd": a["query_end"],
                "SubjectStart": a["subject_start"], "SubjectEnd": a["subject_end"]}
    for qr in query_results:   # sort.Slice by BitScore descending (search.go:492-494; ties keep their order here)
        qr["SearchResults"]["Hits"].sort(key=lambda h: -h["Alignment"]["BitScore"])
    return query_results
