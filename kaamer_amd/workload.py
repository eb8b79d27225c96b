"""Deterministic synthetic inputs of SURVEY.md §8(d) / BASELINE.md (seed 20261003).

  DB-S / DB-SP : proteins, log-normal length (median 300, sigma 0.6, clip 30..5000),
                 Swiss-Prot background residue frequencies + U/X/B/Z at 1e-4 each,
                 families of 10 (one founder + 9 members at 10..30 % substitutions)
  Q-P          : protein queries, 80 % DB members with 5 % substitutions, 20 % random
  Q-R150       : 150-nt reads, 90 % back-translated 50-aa DB windows (random
                 synonymous codons of table 11, random strand, 1 % substitutions,
                 0.1 % N), 10 % uniform random ACGT

Everything is vectorised numpy so DB-SP (560 k proteins, ~2e8 residues) is
generated in seconds.  Outputs are packed (uint8 buffer, uint64 offsets[n+1]).
"""
import numpy as np

SEED = 20261003

_AA20 = "ACDEFGHIKLMNPQRSTVWY"
# Swiss-Prot release statistics (percent), order of _AA20
_FREQ = np.array([8.25, 1.38, 5.46, 6.72, 3.86, 7.07, 2.27, 5.91, 5.80, 9.65,
                  2.41, 4.06, 4.74, 3.93, 5.53, 6.65, 5.36, 6.85, 1.10, 2.92], dtype=np.float64)
_RARE = "UXBZ"
_ALPHABET = np.frombuffer((_AA20 + _RARE).encode(), dtype=np.uint8)


def _cdf():
    f = _FREQ / _FREQ.sum() * (1.0 - 4e-4)
    p = np.concatenate([f, np.full(4, 1e-4)])
    c = np.cumsum(p)
    c[-1] = 1.0
    return c


_CDF = _cdf()


def _residues(rng, n):
    u = rng.random(n, dtype=np.float32).astype(np.float64)
    return _ALPHABET[np.minimum(np.searchsorted(_CDF, u, side="right"), len(_ALPHABET) - 1)]


def _lengths(rng, n, median=300.0, sigma=0.6, lo=30, hi=5000):
    l = np.exp(rng.normal(np.log(median), sigma, n))
    return np.clip(np.rint(l), lo, hi).astype(np.int64)


def _offsets(lengths):
    o = np.zeros(len(lengths) + 1, dtype=np.uint64)
    o[1:] = np.cumsum(lengths, dtype=np.uint64)
    return o


def _gather_index(src_off, lengths):
    """index array that concatenates slices [src_off[i], src_off[i]+lengths[i])"""
    total = int(lengths.sum())
    dst_off = np.zeros(len(lengths), dtype=np.int64)
    dst_off[1:] = np.cumsum(lengths[:-1])
    return np.repeat(src_off.astype(np.int64) - dst_off, lengths) + np.arange(total, dtype=np.int64)


def make_db(n_proteins, seed=SEED, family=10, chunk_families=20000):
    """-> (buf uint8, offsets uint64[n+1]); protein p belongs to family p // family."""
    rng = np.random.default_rng(seed)
    n_fam = (n_proteins + family - 1) // family
    fam_len = _lengths(rng, n_fam)
    prot_len = np.repeat(fam_len, family)[:n_proteins]
    offs = _offsets(prot_len)
    buf = np.empty(int(offs[-1]), dtype=np.uint8)
    for f0 in range(0, n_fam, chunk_families):
        f1 = min(n_fam, f0 + chunk_families)
        fl = fam_len[f0:f1]
        founders = _residues(rng, int(fl.sum()))
        fo = np.zeros(len(fl), dtype=np.int64)
        fo[1:] = np.cumsum(fl[:-1])
        p0, p1 = f0 * family, min(n_proteins, f1 * family)
        pl = prot_len[p0:p1]
        fam_of = (np.arange(p0, p1) // family) - f0
        idx = _gather_index(fo[fam_of], pl)
        seqs = founders[idx]
        # members (p % family != 0): substitution rate uniform in [0.1, 0.3]
        rate = rng.uniform(0.10, 0.30, p1 - p0)
        rate[(np.arange(p0, p1) % family) == 0] = 0.0
        mask = rng.random(len(seqs), dtype=np.float32) < np.repeat(rate, pl).astype(np.float32)
        seqs[mask] = _residues(rng, int(mask.sum()))
        buf[int(offs[p0]):int(offs[p1])] = seqs
    return buf, offs


def _make_db_span(args):
    n, seed, family = args
    return make_db(n, seed=seed, family=family)


def make_db_parallel(n_proteins, seed=SEED, family=10, workers=None, span=200000):
    """A database with make_db's statistics, generated `span` proteins at a time by a pool of processes (each span is
    make_db under its own seed; families do not cross spans).  For the DB-UR-lite sizes: make_db itself is one thread."""
    import multiprocessing as mp
    import os
    span -= span % family
    jobs = [(min(span, n_proteins - a), seed + 1 + i, family) for i, a in enumerate(range(0, n_proteins, span))]
    workers = workers or min(len(jobs), max(1, (os.cpu_count() or 2) - 1))
    lens = []
    parts = []
    with mp.get_context("fork").Pool(workers) as pool:
        for buf, offs in pool.imap(_make_db_span, jobs):
            parts.append(buf)
            lens.append(np.diff(offs.astype(np.int64)))
    offs = _offsets(np.concatenate(lens))
    out = np.empty(int(offs[-1]), dtype=np.uint8)
    at = 0
    for b in parts:
        out[at:at + len(b)] = b
        at += len(b)
    return out, offs


def make_protein_queries(db, n_queries, seed=SEED + 1, member_frac=0.8, subst=0.05):
    """Q-P.  -> (buf, offsets); all queries >= 13 aa (so SizeInKmer >= 7)."""
    buf, offs = db
    n_db = len(offs) - 1
    rng = np.random.default_rng(seed)
    is_member = rng.random(n_queries) < member_frac
    src = rng.integers(0, n_db, n_queries)
    db_len = (offs[1:] - offs[:-1]).astype(np.int64)
    qlen = np.where(is_member, db_len[src], np.maximum(_lengths(rng, n_queries), 13))
    qoffs = _offsets(qlen)
    out = _residues(rng, int(qoffs[-1]))  # random background everywhere, members overwritten
    m = np.flatnonzero(is_member)
    if len(m):
        idx_src = _gather_index(offs[:-1][src[m]].astype(np.int64), qlen[m])
        idx_dst = _gather_index(qoffs[:-1][m].astype(np.int64), qlen[m])
        vals = buf[idx_src].copy()
        mask = rng.random(len(vals), dtype=np.float32) < subst
        vals[mask] = _residues(rng, int(mask.sum()))
        out[idx_dst] = vals
    return out, qoffs


# ---- back-translation (NCBI table 11) -----------------------------------------------------
_BASES = "tcag"
_T11 = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"


def _codon_tables():
    codons = np.zeros((256, 6, 3), dtype=np.uint8)
    count = np.zeros(256, dtype=np.int64)
    n = 0
    for a in _BASES:
        for b in _BASES:
            for c in _BASES:
                aa = ord(_T11[n])
                codons[aa, count[aa]] = [ord(a), ord(b), ord(c)]
                count[aa] += 1
                n += 1
    for aa in range(256):  # letters without codons (U, X, B, Z, ...) -> "nnn"
        if count[aa] == 0:
            codons[aa, 0] = [ord("n")] * 3
            count[aa] = 1
    return codons, count


_CODONS, _CODON_COUNT = _codon_tables()
_COMP = np.arange(256, dtype=np.uint8)
for _a, _b in zip(b"acgtACGT", b"tgcaTGCA"):
    _COMP[_a] = _b


def make_reads(db, n_reads, read_len=150, seed=SEED + 2, db_frac=0.9, subst=0.01, n_rate=0.001):
    """Q-R150.  -> (buf, offsets) of upper-case nucleotides."""
    buf, offs = db
    n_db = len(offs) - 1
    rng = np.random.default_rng(seed)
    aa_per = read_len // 3
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    reads = acgt[rng.integers(0, 4, (n_reads, read_len))]
    from_db = np.flatnonzero(rng.random(n_reads) < db_frac)
    if len(from_db):
        db_len = (offs[1:] - offs[:-1]).astype(np.int64)
        src = rng.integers(0, n_db, len(from_db))
        ok = db_len[src] >= aa_per
        from_db, src = from_db[ok], src[ok]
        start = (rng.random(len(src)) * (db_len[src] - aa_per + 1)).astype(np.int64)
        idx = (offs[:-1][src].astype(np.int64) + start)[:, None] + np.arange(aa_per)[None, :]
        aa = buf[idx]                                             # (m, aa_per)
        pick = (rng.random(aa.shape) * _CODON_COUNT[aa]).astype(np.int64)
        nt = _CODONS[aa, pick].reshape(len(src), aa_per * 3)      # lower-case
        nt = np.where((nt >= 97) & (nt <= 122), nt - 32, nt).astype(np.uint8)
        sub = rng.random(nt.shape, dtype=np.float32) < subst
        nt[sub] = acgt[rng.integers(0, 4, int(sub.sum()))]
        minus = rng.random(len(src)) < 0.5
        nt[minus] = _COMP[nt[minus][:, ::-1]]
        reads[from_db, :aa_per * 3] = nt
    nmask = rng.random(reads.shape, dtype=np.float32) < n_rate
    reads[nmask] = ord("N")
    lens = np.full(n_reads, read_len, dtype=np.int64)
    return np.ascontiguousarray(reads).reshape(-1), _offsets(lens)


def make_reads_mix(db, n_reads, seed=SEED + 3, db_frac=0.9, subst=0.01, n_rate=0.001):
    """Q-mix of SURVEY 8d (BASELINE configs[4]): read lengths 100 / 150 / 250 nt (30 / 50 / 15 %) plus 5 % long
    reads (log-normal, median 3 kb, sigma 0.5, clip 500..20000).  db_frac of the reads carry a back-translated
    window of a DB protein (as long as the read and the protein allow, random strand and offset inside the
    read); the rest, and everything around the window, is uniform random ACGT.  -> (buf, offsets)."""
    buf, offs = db
    n_db = len(offs) - 1
    rng = np.random.default_rng(seed)
    u = rng.random(n_reads)
    lens = np.where(u < 0.30, 100, np.where(u < 0.80, 150, 250)).astype(np.int64)
    is_long = u >= 0.95
    ll = np.clip(np.rint(np.exp(rng.normal(np.log(3000.0), 0.5, n_reads))), 500, 20000).astype(np.int64)
    lens[is_long] = ll[is_long]
    roffs = _offsets(lens)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    out = acgt[rng.integers(0, 4, int(roffs[-1]))]
    from_db = np.flatnonzero(rng.random(n_reads) < db_frac)
    if len(from_db):
        db_len = (offs[1:] - offs[:-1]).astype(np.int64)
        src = rng.integers(0, n_db, len(from_db))
        aa_n = np.minimum(lens[from_db] // 3, db_len[src])                        # residues back-translated
        start = (rng.random(len(src)) * (db_len[src] - aa_n + 1)).astype(np.int64)
        room = lens[from_db] - 3 * aa_n
        at = (rng.random(len(src)) * (room + 1)).astype(np.int64)                 # where the window sits in the read
        aa = buf[_gather_index(offs[:-1][src].astype(np.int64) + start, aa_n)]
        pick = (rng.random(len(aa)) * _CODON_COUNT[aa]).astype(np.int64)
        nt = _CODONS[aa, pick].reshape(-1)                                       # lower-case, 3 per residue
        nt = np.where((nt >= 97) & (nt <= 122), nt - 32, nt).astype(np.uint8)
        sub = rng.random(len(nt), dtype=np.float32) < subst
        nt[sub] = acgt[rng.integers(0, 4, int(sub.sum()))]
        # minus strand: reverse-complement the window in place (segment-wise reversal through an index)
        seg_len = 3 * aa_n
        seg_off = np.zeros(len(seg_len), dtype=np.int64)
        seg_off[1:] = np.cumsum(seg_len[:-1])
        minus = rng.random(len(src)) < 0.5
        within = np.arange(len(nt), dtype=np.int64) - np.repeat(seg_off, seg_len)
        rev_idx = np.repeat(seg_off + seg_len - 1, seg_len) - within
        is_minus = np.repeat(minus, seg_len)
        nt = np.where(is_minus, _COMP[nt[rev_idx]], nt)
        out[_gather_index(roffs[:-1][from_db].astype(np.int64) + at, seg_len)] = nt
    nmask = rng.random(len(out), dtype=np.float32) < n_rate
    out[nmask] = ord("N")
    return np.ascontiguousarray(out), roffs


def make_db_zipf(n_proteins, seed=SEED + 7, n_motifs=300000, zipf_a=0.7, per_residues=120):
    """A DB of the size and length distribution of DB-SP whose shared content follows a power law, like the
    domain families of a real protein database: proteins are random background with motifs (15..40 residues)
    pasted in, one per `per_residues` residues, the motif drawn with P(rank r) ~ r^-zipf_a from a library of
    `n_motifs`.  The most frequent motifs occur in ~1e4 proteins, so their 7-mers have postings lists of that
    length and a query expands ~25 times the postings of a DB-SP query, most of them past the LDS counting
    tables (the stress case for the postings expansion and the HBM counting tier)."""
    rng = np.random.default_rng(seed)
    lens = _lengths(rng, n_proteins)
    offs = _offsets(lens)
    buf = _residues(rng, int(offs[-1]))
    mlen = rng.integers(15, 41, n_motifs).astype(np.int64)
    moff = _offsets(mlen)
    mbuf = _residues(rng, int(moff[-1]))
    w = np.arange(1, n_motifs + 1, dtype=np.float64) ** (-zipf_a)
    cdf = np.cumsum(w / w.sum())
    n_ins = np.maximum(lens // per_residues, 1)
    total = int(n_ins.sum())
    motif = np.minimum(np.searchsorted(cdf, rng.random(total), side="right"), n_motifs - 1)
    prot = np.repeat(np.arange(n_proteins), n_ins)
    ml = mlen[motif]
    ok = ml <= lens[prot]
    motif, prot, ml = motif[ok], prot[ok], ml[ok]
    at = (rng.random(len(prot)) * (lens[prot] - ml + 1)).astype(np.int64)
    buf[_gather_index(offs[:-1][prot].astype(np.int64) + at, ml)] = mbuf[_gather_index(moff[:-1][motif].astype(np.int64), ml)]
    return buf, offs


def unpack(packed):
    buf, offs = packed
    return [bytes(buf[int(offs[i]):int(offs[i + 1])]) for i in range(len(offs) - 1)]


def fastq_text(reads):
    """packed reads -> FASTQ text (numpy, no per-read Python): "@r\\n<seq>\\n+\\n<quality 'I' x len>\\n" per read."""
    buf, offs = reads
    offs = offs.astype(np.int64)
    lens = offs[1:] - offs[:-1]
    rec = 2 * lens + 7
    start = np.zeros(len(lens) + 1, dtype=np.int64)
    np.cumsum(rec, out=start[1:])
    out = np.full(int(start[-1]), ord("I"), dtype=np.uint8)
    s0 = start[:-1]
    out[s0] = ord("@"); out[s0 + 1] = ord("r"); out[s0 + 2] = 10
    dst = np.repeat(s0 + 3 - offs[:-1], lens) + np.arange(int(offs[-1]), dtype=np.int64)
    out[dst] = buf[:int(offs[-1])]
    e = s0 + 3 + lens
    out[e] = 10; out[e + 1] = ord("+"); out[e + 2] = 10
    out[start[1:] - 1] = 10
    return out.tobytes()
