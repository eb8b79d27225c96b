"""ctypes binding of oracle/kaamer_oracle.c — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product (kaamer_amd/) never does.  See the C file's header
for the reference file:line each function follows and for the "parity
unpinned" statement.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libkaamer_oracle.so")


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("kaamer_oracle.c", "align_oracle.c")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libkaamer_oracle.so"])
    return _SO


class _Orf(C.Structure):
    _fields_ = [("start_position", C.c_int32), ("end_position", C.c_int32),
                ("plus_strand", C.c_int32), ("aa_off", C.c_uint32), ("aa_len", C.c_uint32),
                ("sa_off", C.c_uint32), ("sa_len", C.c_uint32)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_SO)
    u8p, u32p, u64p, i32p, i64p = (C.POINTER(t) for t in
                                   (C.c_uint8, C.c_uint32, C.c_uint64, C.c_int32, C.c_int64))
    L.ko_encode_kmer.restype = C.c_uint32
    L.ko_encode_kmer.argtypes = [C.c_char_p, C.c_int]
    L.ko_decode_kmer.argtypes = [C.c_uint32, C.c_char_p]
    L.ko_create_bytes_key.argtypes = [C.c_char_p, C.c_char_p]
    L.ko_gcode_bacteria_packed.restype = C.c_uint32
    L.ko_gcode_bacteria_packed.argtypes = [C.c_char_p]
    L.ko_orf_list_new.restype = C.c_void_p
    L.ko_orf_list_free.argtypes = [C.c_void_p]
    L.ko_orf_list_clear.argtypes = [C.c_void_p]
    L.ko_orf_list_count.restype = C.c_size_t
    L.ko_orf_list_count.argtypes = [C.c_void_p]
    L.ko_orf_list_orfs.restype = C.POINTER(_Orf)
    L.ko_orf_list_orfs.argtypes = [C.c_void_p]
    L.ko_orf_list_aa.restype = u8p
    L.ko_orf_list_aa.argtypes = [C.c_void_p]
    L.ko_orf_list_sa.restype = i32p
    L.ko_orf_list_sa.argtypes = [C.c_void_p]
    L.ko_orf_list_aa_len.restype = C.c_size_t
    L.ko_orf_list_aa_len.argtypes = [C.c_void_p]
    L.ko_orf_list_sa_len.restype = C.c_size_t
    L.ko_orf_list_sa_len.argtypes = [C.c_void_p]
    L.ko_get_orfs.argtypes = [C.c_char_p, C.c_long, C.c_void_p]
    L.ko_size_in_kmer.restype = C.c_int32
    L.ko_size_in_kmer.argtypes = [C.c_char_p, C.c_long]
    L.ko_index_from_pairs.restype = C.c_void_p
    L.ko_index_from_pairs.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    L.ko_index_from_proteins.restype = C.c_void_p
    L.ko_index_from_proteins.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    L.ko_index_free.argtypes = [C.c_void_p]
    L.ko_index_n_pairs.restype = C.c_uint64
    L.ko_index_n_pairs.argtypes = [C.c_void_p]
    L.ko_index_pairs.restype = u64p
    L.ko_index_pairs.argtypes = [C.c_void_p]
    L.ko_fasta_ids.argtypes = [C.c_uint32, C.c_void_p]
    L.ko_index_get.restype = C.c_uint32
    L.ko_index_get.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
    L.ko_result_new.restype = C.c_void_p
    L.ko_result_free.argtypes = [C.c_void_p]
    L.ko_result_n.restype = C.c_size_t
    L.ko_result_n.argtypes = [C.c_void_p]
    L.ko_result_pid.restype = u32p
    L.ko_result_pid.argtypes = [C.c_void_p]
    L.ko_result_kmatch.restype = i64p
    L.ko_result_kmatch.argtypes = [C.c_void_p]
    L.ko_result_pos.restype = u8p
    L.ko_result_pos.argtypes = [C.c_void_p]
    for f in ("ko_result_n_lookup", "ko_result_n_found", "ko_result_n_post"):
        getattr(L, f).restype = C.c_uint64
        getattr(L, f).argtypes = [C.c_void_p]
    L.ko_search_query.argtypes = [C.c_void_p, C.c_char_p, C.c_long, C.c_int32, C.c_int, C.c_void_p]
    L.ko_filter_results.restype = C.c_int64
    L.ko_filter_results.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_double, C.c_int64, C.c_int64]
    L.ko_set_best_start_codon.restype = C.c_int32
    L.ko_set_best_start_codon.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p,
                                          C.c_int32, C.c_int32, C.c_char_p, C.c_long,
                                          C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    for f in ("ko_search_protein_batch", "ko_search_reads_batch"):
        getattr(L, f).restype = C.c_uint64
        getattr(L, f).argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    _lib = L
    return L


def _b(s):
    return s.encode("latin-1") if isinstance(s, str) else bytes(s)


def encode_kmer(kmer):
    k = _b(kmer)
    return int(lib().ko_encode_kmer(k, len(k)))


def decode_kmer(key):
    buf = C.create_string_buffer(8)
    lib().ko_decode_kmer(key, buf)
    return buf.value.decode()


def create_bytes_key(kmer):
    buf = C.create_string_buffer(4)
    lib().ko_create_bytes_key(_b(kmer), buf)
    return buf.raw


def gcode_bacteria(codon):
    v = lib().ko_gcode_bacteria_packed(_b(codon))
    aa = chr(v & 0xFF) if v & 0xFF else ""
    return aa, bool(v >> 8 & 1), bool(v >> 9 & 1)


def size_in_kmer(seq):
    s = _b(seq)
    return int(lib().ko_size_in_kmer(s, len(s)))


def get_orfs(dna):
    """-> list of dict(seq, start, end, plus, starts) in the reference's order."""
    L = lib()
    d = _b(dna)
    ol = L.ko_orf_list_new()
    try:
        L.ko_get_orfs(d, len(d), ol)
        n = L.ko_orf_list_count(ol)
        orfs, aa, sa = L.ko_orf_list_orfs(ol), L.ko_orf_list_aa(ol), L.ko_orf_list_sa(ol)
        out = []
        for i in range(n):
            o = orfs[i]
            out.append(dict(seq=bytes(aa[o.aa_off:o.aa_off + o.aa_len]).decode("latin-1"),
                            start=o.start_position, end=o.end_position, plus=bool(o.plus_strand),
                            starts=[sa[o.sa_off + j] for j in range(o.sa_len)]))
        return out
    finally:
        L.ko_orf_list_free(ol)


def pack(seqs):
    """list of bytes/str -> (u8 array, u64 offsets[n+1])"""
    bs = [_b(s) for s in seqs]
    offs = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        offs[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
    buf = np.frombuffer(b"".join(bs), dtype=np.uint8).copy() if bs else np.zeros(0, np.uint8)
    return buf, offs


def fasta_ids(n):
    ids = np.zeros(n, dtype=np.uint32)
    lib().ko_fasta_ids(n, ids.ctypes.data)
    return ids


class Index:
    """key -> set<protein id>; makedb emit loops + indexdb de-dup (see C file)."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def from_pairs(cls, keys, ids):
        keys = np.ascontiguousarray(keys, dtype=np.uint32)
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        return cls(lib().ko_index_from_pairs(keys.ctypes.data, ids.ctypes.data, len(keys)))

    @classmethod
    def from_proteins(cls, seqs, ids=None, packed=None):
        buf, offs = packed if packed is not None else pack(seqs)
        n = len(offs) - 1
        if ids is None:
            ids = np.arange(n, dtype=np.uint32)  # TSV rule: 0-based (inputTSV.go:141-142)
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        buf = np.ascontiguousarray(buf)
        return cls(lib().ko_index_from_proteins(buf.ctypes.data, offs.ctypes.data, n, ids.ctypes.data))

    def pairs(self):
        n = lib().ko_index_n_pairs(self._h)
        p = lib().ko_index_pairs(self._h)
        return np.ctypeslib.as_array(p, shape=(n,)).copy() if n else np.zeros(0, np.uint64)

    def get(self, key):
        n = lib().ko_index_get(self._h, key, None, 0)
        out = np.zeros(n, dtype=np.uint32)
        if n:
            lib().ko_index_get(self._h, key, out.ctypes.data, n)
        return out

    def search(self, seq, size=None, want_positions=False):
        """-> (pids u32[n], kmatch i64[n], pos bool[n,size] or None); hits ordered
        by (Kmatch desc, pid asc)."""
        L = lib()
        s = _b(seq)
        if size is None:
            size = int(L.ko_size_in_kmer(s, len(s)))
        r = L.ko_result_new()
        try:
            L.ko_search_query(self._h, s, len(s), size, int(want_positions), r)
            n = L.ko_result_n(r)
            pid = np.ctypeslib.as_array(L.ko_result_pid(r), shape=(n,)).copy() if n else np.zeros(0, np.uint32)
            km = np.ctypeslib.as_array(L.ko_result_kmatch(r), shape=(n,)).copy() if n else np.zeros(0, np.int64)
            pos = None
            if want_positions:
                pos = (np.ctypeslib.as_array(L.ko_result_pos(r), shape=(n, size)).copy().astype(bool)
                       if n and size > 0 else np.zeros((n, max(size, 0)), bool))
            return pid, km, pos
        finally:
            L.ko_result_free(r)

    def batch(self, packed, kind="protein", begin=0, end=None):
        """Timed CPU-baseline driver. -> dict(checksum, n_lookup, n_found, n_post)"""
        L = lib()
        buf, offs = packed
        if end is None:
            end = len(offs) - 1
        r = L.ko_result_new()
        try:
            f = L.ko_search_protein_batch if kind == "protein" else L.ko_search_reads_batch
            cs = f(self._h, buf.ctypes.data, offs.ctypes.data, begin, end, r)
            return dict(checksum=int(cs), n_lookup=int(L.ko_result_n_lookup(r)),
                        n_found=int(L.ko_result_n_found(r)), n_post=int(L.ko_result_n_post(r)))
        finally:
            L.ko_result_free(r)

    def close(self):
        if self._h:
            lib().ko_index_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def filter_results(kmatch_sorted_desc, size_in_kmer_, min_k_ratio=0.05, min_k_match=10, max_results=10):
    km = np.ascontiguousarray(kmatch_sorted_desc, dtype=np.int64)
    return int(lib().ko_filter_results(km.ctypes.data, len(km), size_in_kmer_, min_k_ratio,
                                       min_k_match, max_results))


def set_best_start_codon(kmatch, pos, size, starts_alt, plus, seq, start_position):
    """-> (best_start_trim, new_start_position, new_size_in_kmer)"""
    km = np.ascontiguousarray(kmatch, dtype=np.int64)
    p = np.ascontiguousarray(pos, dtype=np.uint8)
    sa = np.ascontiguousarray(starts_alt, dtype=np.int32)
    sp = C.c_int32(start_position)
    so = C.c_int32(0)
    s = _b(seq)
    t = lib().ko_set_best_start_codon(km.ctypes.data, len(km), p.ctypes.data, size, sa.ctypes.data,
                                      len(sa), int(plus), s, len(s), C.byref(sp), C.byref(so))
    return int(t), int(sp.value), int(so.value)


# ---- the alignment step (-aln): oracle/align_oracle.c ------------------------------------------------------
class _Alignment(C.Structure):
    _fields_ = [("identity", C.c_float), ("similarity", C.c_float), ("length", C.c_int32), ("mismatches", C.c_int32),
                ("gap_openings", C.c_int32), ("raw", C.c_int32), ("bitscore", C.c_double), ("evalue", C.c_double),
                ("q_start", C.c_int32), ("q_end", C.c_int32), ("s_start", C.c_int32), ("s_end", C.c_int32)]


def matrix_scores(sub_matrix, gap_open, gap_extend):
    """GetMatrixScores (matrixScores.go:107-115) -> (lambda, K) or None ("No matrix found")"""
    L = lib()
    L.ko_matrix_scores.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lam, k = C.c_double(), C.c_double()
    if L.ko_matrix_scores(sub_matrix.encode(), gap_open, gap_extend, C.byref(lam), C.byref(k)):
        return None
    return lam.value, k.value


def b62(a, b, gap_col=0):
    """matrix.BLOSUM62 by letters of "-ABCDEFGHIJKLMNPQRSTVWXYZ*" """
    L = lib()
    L.ko_b62.argtypes = [C.c_int, C.c_int, C.c_int]
    alpha = "-ABCDEFGHIJKLMNPQRSTVWXYZ*"
    return L.ko_b62(alpha.index(a), alpha.index(b), gap_col)


def align(query, subject, number_of_aa, sub_matrix="blosum62", gap_open=11, gap_extend=1, dp_gap_open=-11, gap_col=0):
    """align.Align (align.go:46-161) -> dict, or None for "No matrix found"; raises ValueError on a letter outside the
    alphabet"""
    L = lib()
    L.ko_align.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_uint64, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int,
                           C.POINTER(_Alignment), C.c_char_p, C.POINTER(C.c_int)]
    q = query if isinstance(query, bytes) else query.encode("latin-1")
    s = subject if isinstance(subject, bytes) else subject.encode("latin-1")
    out = _Alignment()
    buf = C.create_string_buffer(3 * (len(q) + len(s)) + 8)
    n = C.c_int()
    rc = L.ko_align(q, len(q), s, len(s), int(number_of_aa), sub_matrix.encode(), gap_open, gap_extend, dp_gap_open, gap_col,
                    C.byref(out), buf, C.byref(n))
    if rc == 1:
        return None
    if rc == 2:
        raise ValueError("letter outside the protein alphabet")
    if rc:
        raise MemoryError()
    ln = n.value
    raw = buf.raw
    d = {k: getattr(out, k) for k, _ in _Alignment._fields_}
    d["aln"] = (raw[:ln].decode("latin-1"), raw[ln:2 * ln].decode("latin-1"), raw[2 * ln:3 * ln].decode("latin-1"))
    return d
