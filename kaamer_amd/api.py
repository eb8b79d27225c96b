"""Python plumbing over the C ABI: builds images from numpy buffers, opens an
index on a device, and runs batches either from host buffers or from
torch-owned device tensors (device memory and streams are torch's job here,
nothing else).  All compute happens inside libkaamer_hip.so.
"""
import ctypes as C

import numpy as np

from . import abi


def pack_sequences(seqs):
    """list of bytes/str -> (uint8 array, uint64 offsets[n+1])"""
    bs = [s.encode("latin-1") if isinstance(s, str) else bytes(s) for s in seqs]
    offs = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        offs[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
    buf = np.frombuffer(b"".join(bs), dtype=np.uint8).copy() if bs else np.zeros(0, np.uint8)
    return buf, offs


class Image:
    """Host copy of one shard's table (the builder's output)."""

    def __init__(self, handle):
        self._h = C.c_void_p(handle)

    @classmethod
    def from_pairs(cls, keys, ids, shard=0, n_shards=1, load_factor=0.5):
        keys = np.ascontiguousarray(keys, dtype=np.uint32)
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        pairs = np.empty((len(keys), 2), dtype=np.uint32)
        pairs[:, 0] = keys
        pairs[:, 1] = ids
        h = C.c_void_p()
        abi.check(abi.lib().kaamer_image_build_pairs(pairs.ctypes.data, len(keys), shard, n_shards,
                                                     load_factor, C.byref(h)))
        return cls(h.value)

    @classmethod
    def from_proteins(cls, seqs=None, ids=None, packed=None, shard=0, n_shards=1, load_factor=0.5, device=None):
        """device=None: the host builder; device=d: the same image built on GPU d (byte-identical)."""
        buf, offs = packed if packed is not None else pack_sequences(seqs)
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        idp = None
        if ids is not None:
            ids = np.ascontiguousarray(ids, dtype=np.uint32)
            idp = ids.ctypes.data
        h = C.c_void_p()
        if device is None:
            abi.check(abi.lib().kaamer_image_build_proteins(buf.ctypes.data, offs.ctypes.data, idp,
                                                            len(offs) - 1, shard, n_shards, load_factor,
                                                            C.byref(h)))
        else:
            abi.check(abi.lib().kaamer_image_build_proteins_device(buf.ctypes.data, offs.ctypes.data, idp,
                                                                   len(offs) - 1, shard, n_shards, load_factor,
                                                                   int(device), C.byref(h)))
        return cls(h.value)

    @classmethod
    def load(cls, path):
        h = C.c_void_p()
        abi.check(abi.lib().kaamer_image_load(str(path).encode(), C.byref(h)))
        return cls(h.value)

    def save(self, path):
        abi.check(abi.lib().kaamer_image_save(self._h, str(path).encode()))

    def stats(self):
        s = abi.ImageStats()
        abi.check(abi.lib().kaamer_image_get_stats(self._h, C.byref(s)))
        return s.as_dict()

    def get(self, key):
        n = abi.lib().kaamer_image_get(self._h, int(key), None, 0)
        out = np.zeros(n, dtype=np.uint32)
        if n:
            abi.lib().kaamer_image_get(self._h, int(key), out.ctypes.data, n)
        return out

    def close(self):
        if self._h:
            abi.lib().kaamer_image_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Proteins:
    """What makedb accepts from a database file (ids, sequences, annotations): kaamer_makedb_fasta / _tsv
    (pkg/makedb/inputFASTA.go, inputTSV.go) and the table kaamer_fetch_hits reads (search.go:454-470)."""

    def __init__(self, handle):
        self._h = C.c_void_p(handle)

    @classmethod
    def _make(cls, fn, text):
        text = bytes(text)
        h = C.c_void_p()
        abi.check(fn(text, len(text), C.byref(h)))
        return cls(h.value)

    @classmethod
    def from_fasta(cls, text):
        return cls._make(abi.lib().kaamer_makedb_fasta, text)

    @classmethod
    def from_tsv(cls, text):
        return cls._make(abi.lib().kaamer_makedb_tsv, text)

    @classmethod
    def from_embl(cls, text):
        return cls._make(abi.lib().kaamer_makedb_embl, text)

    @classmethod
    def from_gbk(cls, text):
        return cls._make(abi.lib().kaamer_makedb_gbk, text)

    @classmethod
    def load(cls, path):
        h = C.c_void_p()
        abi.check(abi.lib().kaamer_proteins_load(str(path).encode(), C.byref(h)))
        return cls(h.value)

    def save(self, path):
        abi.check(abi.lib().kaamer_proteins_save(self._h, str(path).encode()))

    def __len__(self):
        return abi.lib().kaamer_proteins_count(self._h)

    @property
    def ids(self):
        n = len(self)
        return np.ctypeslib.as_array(abi.lib().kaamer_proteins_ids(self._h), shape=(n,)).copy() if n else np.zeros(0, np.uint32)

    @property
    def packed(self):
        n = len(self)
        offs = np.ctypeslib.as_array(abi.lib().kaamer_proteins_offsets(self._h), shape=(n + 1,)).copy()
        buf = np.ctypeslib.as_array(abi.lib().kaamer_proteins_seqs(self._h), shape=(int(offs[-1]),)).copy() \
            if offs[-1] else np.zeros(0, np.uint8)
        return buf, offs

    @property
    def feature_names(self):
        L = abi.lib()
        return [L.kaamer_proteins_feature_name(self._h, i) for i in range(L.kaamer_proteins_n_features(self._h))]

    def stats(self):
        out = (C.c_uint64 * 3)()
        abi.lib().kaamer_proteins_stats(self._h, out)
        return dict(NumberOfProteins=out[0], NumberOfAA=out[1], NumberOfKmers=out[2], Features=self.feature_names)

    def image(self, shard=0, n_shards=1, load_factor=0.5, device=None):
        h = C.c_void_p()
        if device is None:
            abi.check(abi.lib().kaamer_image_build_makedb(self._h, shard, n_shards, load_factor, C.byref(h)))
        else:
            abi.check(abi.lib().kaamer_image_build_makedb_device(self._h, shard, n_shards, load_factor, int(device), C.byref(h)))
        return Image(h.value)

    def fetch_hits(self, ids):
        """[None | dict(EntryId, Sequence, Length, Features)] per protein id (protein.proto)"""
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        ent = (abi.ProteinEntry * max(1, len(ids)))()
        abi.check(abi.lib().kaamer_fetch_hits(self._h, ids.ctypes.data, len(ids), ent))
        names = self.feature_names
        out = []
        for e in ent[:len(ids)]:
            if not e.found:
                out.append(None)
                continue
            fo = [e.feature_off[i] for i in range(e.n_features + 1)]
            blob = C.string_at(e.features + fo[0], fo[-1] - fo[0]) if fo[-1] > fo[0] else b""
            out.append(dict(EntryId=C.string_at(e.entry_id, e.entry_id_len), Sequence=C.string_at(e.sequence, e.sequence_len),
                            Length=e.length,
                            Features={names[i]: blob[fo[i] - fo[0]:fo[i + 1] - fo[0]] for i in range(e.n_features)}))
        return out

    def close(self):
        if self._h:
            abi.lib().kaamer_proteins_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def build_image_from_proteins(seqs, ids=None, **kw):
    return Image.from_proteins(seqs, ids=ids, **kw)


META_DTYPE = np.dtype([("src_seq", "<u4"), ("size_in_kmer", "<i4"), ("start_position", "<i4"), ("end_position", "<i4"),
                       ("plus_strand", "<i4"), ("aa_len", "<u4"), ("aa_off", "<u8"), ("sa_off", "<u4"), ("sa_len", "<u4")])


class TopResult:
    """Host view of one kaamer_batch_top (copied out, the C object is freed): the reported queries
    (`rep_query`, `meta`, `trim`, CSR `top_off` + `top_pid` / `top_kmatch` / `top_first_pos`, `orf_aa`),
    plus dense per-query views for convenience (`top_cnt[n_queries]`, `dense()`)."""

    def __init__(self, out):
        o = out.contents
        n, r, k = o.n_queries, o.n_reported, o.max_results
        self.n_queries, self.n_reported, self.max_results = n, r, k
        z = np.zeros(0, np.uint32)
        self.rep_query = np.ctypeslib.as_array(o.rep_query, shape=(r,)).copy() if r else z
        meta = np.ctypeslib.as_array(C.cast(o.q, C.POINTER(C.c_uint8)), shape=(r * C.sizeof(abi.QueryMeta),)).copy() \
            if r else np.zeros(0, np.uint8)
        self.meta = meta.view(META_DTYPE)
        self.trim = np.ctypeslib.as_array(o.trim, shape=(r,)).copy() if r else np.zeros(0, np.int32)
        self.top_off = np.ctypeslib.as_array(o.top_off, shape=(r + 1,)).copy()
        ne = int(self.top_off[r])
        self.top_pid = np.ctypeslib.as_array(o.top_pid, shape=(ne,)).copy() if ne else z
        self.top_kmatch = np.ctypeslib.as_array(o.top_kmatch, shape=(ne,)).copy() if ne else z
        self.top_first_pos = np.ctypeslib.as_array(o.top_first_pos, shape=(ne,)).copy() if ne else z
        aa_len = int((self.meta["aa_off"] + self.meta["aa_len"]).max()) if r and bool(o.orf_aa) else 0
        self.orf_aa = np.ctypeslib.as_array(o.orf_aa, shape=(aa_len,)).copy() if aa_len else np.zeros(0, np.uint8)
        self.counters = o.counters.as_dict()
        self.top_cnt = np.zeros(n, np.uint32)
        if r:
            self.top_cnt[self.rep_query] = np.diff(self.top_off.astype(np.int64)).astype(np.uint32)

    def dense(self):
        """(top_pid, top_kmatch) as [n_queries, max_results] arrays, zero beyond top_cnt"""
        pid = np.zeros((self.n_queries, self.max_results), np.uint32)
        km = np.zeros((self.n_queries, self.max_results), np.uint32)
        for i in range(self.n_reported):
            a, b = int(self.top_off[i]), int(self.top_off[i + 1])
            pid[self.rep_query[i], :b - a] = self.top_pid[a:b]
            km[self.rep_query[i], :b - a] = self.top_kmatch[a:b]
        return pid, km


class _BatchOwner:
    """frees one kaamer_batch_out when the last array that views its buffers is gone"""

    def __init__(self, out):
        self.out = out

    def __del__(self):
        try:
            if self.out is not None:
                abi.lib().kaamer_batch_free(self.out)
                self.out = None
        except Exception:
            pass


def _owned_view(owner, ptr, n, dtype):
    """numpy view of n items at `ptr` (a ctypes pointer) that keeps `owner` alive: the exporting ctypes array holds a
    reference to it, and every array derived from the view holds the exporter"""
    dtype = np.dtype(dtype)
    if n == 0:
        return np.zeros(0, dtype)
    raw = (C.c_uint8 * (n * dtype.itemsize)).from_address(C.cast(ptr, C.c_void_p).value)
    raw._owner = owner
    return np.frombuffer(raw, dtype=dtype)


class BatchResult:
    """Host view of one kaamer_batch_out: the arrays are views of the C object's buffers (no copies).  The buffers
    live as long as any of the arrays (or anything sliced from them) does: `ix.search(...).hit_pid` stays valid after
    the BatchResult itself is gone."""

    def __init__(self, out):
        o = out.contents
        own = _BatchOwner(out)
        n = o.n_queries
        self.n_queries = n
        self.meta = _owned_view(own, o.q, n, META_DTYPE)
        self.hit_off = _owned_view(own, o.hit_off, n + 1, np.uint64)
        self.hit_cnt = _owned_view(own, o.hit_cnt, n, np.uint32)
        nh = int(self.hit_off[n])
        self.hit_pid = _owned_view(own, o.hit_pid, nh, np.uint32)
        self.hit_kmatch = _owned_view(own, o.hit_kmatch, nh, np.uint32)
        self.hit_first_pos = _owned_view(own, o.hit_first_pos, nh, np.uint32)
        self.counters = o.counters.as_dict()
        self.pos_off = self.pos_bits = None
        if bool(o.pos_off) and bool(o.pos_bits):
            self.pos_off = _owned_view(own, o.pos_off, nh, np.uint64)
            nw = 0
            if nh:  # words in use = end of the last bitmap
                has = self.hit_cnt > 0
                last = (self.hit_off[:n][has] + self.hit_cnt[has].astype(np.uint64) - np.uint64(1)).astype(np.int64)
                words = (self.meta["size_in_kmer"][has].astype(np.int64) + 63) // 64
                nw = int((self.pos_off[last].astype(np.int64) + words).max())
            self.pos_bits = _owned_view(own, o.pos_bits, nw, np.uint64)
        aa_len = int((self.meta["aa_off"] + self.meta["aa_len"]).max()) if n and bool(o.orf_aa) else 0
        self.orf_aa = _owned_view(own, o.orf_aa, aa_len, np.uint8)
        sa_len = int((self.meta["sa_off"] + self.meta["sa_len"]).max()) if n and bool(o.starts_alt) else 0
        self.starts_alt = _owned_view(own, o.starts_alt, sa_len, np.int32)

    def close(self):
        """drops this object's views (the buffers go when no other array views them)"""
        for k in ("meta", "hit_off", "hit_cnt", "hit_pid", "hit_kmatch", "hit_first_pos", "pos_off", "pos_bits", "orf_aa", "starts_alt"):
            setattr(self, k, None)

    def span(self, q):
        """[first, last+1) of query q's hits in the hit arrays"""
        a = int(self.hit_off[q])
        return a, a + int(self.hit_cnt[q])

    def positions(self, q):
        """{protein id: bool[SizeInKmer]} of query q (PositionHits, search.go:442-452)."""
        a, b = self.span(q)
        size = int(self.meta["size_in_kmer"][q])
        out = {}
        words = (size + 63) // 64
        for i in range(a, b):
            w = self.pos_bits[int(self.pos_off[i]):int(self.pos_off[i]) + words]
            bits = np.unpackbits(w.view(np.uint8), bitorder="little")[:size].astype(bool)
            out[int(self.hit_pid[i])] = bits
        return out

    def hits(self, q):
        """{protein id: Kmatch} of query q (the parity object: SURVEY §2.1)."""
        a, b = self.span(q)
        return dict(zip(self.hit_pid[a:b].tolist(), self.hit_kmatch[a:b].tolist()))

    def first_pos(self, q):
        a, b = self.span(q)
        return dict(zip(self.hit_pid[a:b].tolist(), self.hit_first_pos[a:b].tolist()))


class Index:
    """The table resident in one device's HBM."""

    def __init__(self, handle, device):
        self._h = C.c_void_p(handle)
        self.device = device

    @classmethod
    def from_image(cls, image, device=0):
        h = C.c_void_p()
        abi.check(abi.lib().kaamer_index_open_image(image._h, device, C.byref(h)))
        return cls(h.value, device)

    @classmethod
    def from_proteins(cls, seqs=None, ids=None, packed=None, shard=0, n_shards=1, load_factor=0.5, device=0):
        """The table built on the device it is searched on (kaamer_index_build_proteins): no image in between."""
        buf, offs = packed if packed is not None else pack_sequences(seqs)
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        idp = None
        if ids is not None:
            ids = np.ascontiguousarray(ids, dtype=np.uint32)
            idp = ids.ctypes.data
        h = C.c_void_p()
        abi.check(abi.lib().kaamer_index_build_proteins(buf.ctypes.data, offs.ctypes.data, idp, len(offs) - 1, shard,
                                                        n_shards, load_factor, int(device), C.byref(h)))
        return cls(h.value, device)

    @classmethod
    def open(cls, path, device=0):
        h = C.c_void_p()
        abi.check(abi.lib().kaamer_index_open(str(path).encode(), device, C.byref(h)))
        return cls(h.value, device)

    def stats(self):
        s = abi.ImageStats()
        abi.check(abi.lib().kaamer_index_get_stats(self._h, C.byref(s)))
        return s.as_dict()

    def search(self, seqs=None, packed=None, seq_type=abi.PROTEIN, want_positions=False, flat=True):
        """Host-buffer form (kaamer_search_batch_flat: the entry point a cgo shim binds; flat=False: the struct form
        kaamer_search_batch)."""
        buf, offs = packed if packed is not None else pack_sequences(seqs)
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        out = C.POINTER(abi.BatchOut)()
        if flat:
            abi.check(abi.lib().kaamer_search_batch_flat(self._h, buf.ctypes.data if len(buf) else None, offs.ctypes.data,
                                                         len(offs) - 1, seq_type, int(want_positions), C.byref(out)))
        else:
            bi = abi.BatchIn(buf.ctypes.data if len(buf) else None, offs.ctypes.data, len(offs) - 1, seq_type,
                             int(want_positions))
            abi.check(abi.lib().kaamer_search_batch(self._h, C.byref(bi), C.byref(out)))
        return BatchResult(out)

    def submit(self, seqs=None, packed=None, seq_type=abi.PROTEIN, want_positions=False):
        """kaamer_submit_batch_flat -> a ticket whose wait() returns the BatchResult (full hit lists)"""
        buf, offs = packed if packed is not None else pack_sequences(seqs)
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        t = C.c_void_p()
        abi.check(abi.lib().kaamer_submit_batch_flat(self._h, buf.ctypes.data if len(buf) else None, offs.ctypes.data, len(offs) - 1,
                                                     seq_type, int(want_positions), C.byref(t)))
        return FullTicket(t)

    def search_top(self, seqs=None, packed=None, seq_type=abi.PROTEIN, min_k_ratio=0.05, min_k_match=10, max_results=10, flat=True):
        """Host-buffer form that returns the reported hits only (kaamer_search_batch_top_flat; flat=False: the struct
        form): sortMapByValue order, SetBestStartCodon for nucleotide/reads, FilterResults -- all on the device."""
        buf, offs = packed if packed is not None else pack_sequences(seqs)
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        out = C.POINTER(abi.BatchTop)()
        if flat:
            abi.check(abi.lib().kaamer_search_batch_top_flat(self._h, buf.ctypes.data if len(buf) else None, offs.ctypes.data,
                                                             len(offs) - 1, seq_type, min_k_ratio, min_k_match, max_results, C.byref(out)))
        else:
            bi = abi.BatchIn(buf.ctypes.data if len(buf) else None, offs.ctypes.data, len(offs) - 1, seq_type, 0)
            to = abi.TopnOpts(min_k_ratio, min_k_match, max_results, 0, None, None, 0, 0)
            abi.check(abi.lib().kaamer_search_batch_top(self._h, C.byref(bi), C.byref(to), C.byref(out)))
        try:
            return TopResult(out)
        finally:
            abi.lib().kaamer_batch_top_free(out)

    def submit_top(self, seqs=None, packed=None, seq_type=abi.PROTEIN, min_k_ratio=0.05, min_k_match=10, max_results=10, flat=True):
        """kaamer_submit_batch_top[_flat]: the batch is copied and enqueued on a free slot; -> a ticket whose wait() returns
        the TopResult.  Several tickets may be in flight; submit blocks while every slot is busy."""
        buf, offs = packed if packed is not None else pack_sequences(seqs)
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        t = C.c_void_p()
        if flat:
            abi.check(abi.lib().kaamer_submit_batch_top_flat(self._h, buf.ctypes.data if len(buf) else None, offs.ctypes.data,
                                                             len(offs) - 1, seq_type, min_k_ratio, min_k_match, max_results, C.byref(t)))
        else:
            bi = abi.BatchIn(buf.ctypes.data if len(buf) else None, offs.ctypes.data, len(offs) - 1, seq_type, 0)
            to = abi.TopnOpts(min_k_ratio, min_k_match, max_results, 0, None, None, 0, 0)
            abi.check(abi.lib().kaamer_submit_batch_top(self._h, C.byref(bi), C.byref(to), C.byref(t)))
        return TopTicket(t)

    def stream(self, seq_type=abi.READS, min_k_ratio=0.05, min_k_match=10, max_results=10):
        return TopStream(self, seq_type, min_k_ratio, min_k_match, max_results)

    def close(self):
        if self._h:
            abi.lib().kaamer_index_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ShardedIndex:
    """kaamer_index_open_sharded*: one handle over W hash-prefix shards, each resident on its own device, driven from
    one process (no communicator)."""

    def __init__(self, handle):
        self._h = C.c_void_p(handle)

    @classmethod
    def from_images(cls, images, devices):
        n = len(images)
        arr = (C.c_void_p * n)(*[im._h for im in images])
        dev = (C.c_int * n)(*devices)
        h = C.c_void_p()
        abi.check(abi.lib().kaamer_index_open_sharded_images(arr, dev, n, C.byref(h)))
        return cls(h.value)

    @classmethod
    def open(cls, paths, devices):
        n = len(paths)
        arr = (C.c_char_p * n)(*[str(p).encode() for p in paths])
        dev = (C.c_int * n)(*devices)
        h = C.c_void_p()
        abi.check(abi.lib().kaamer_index_open_sharded(arr, dev, n, C.byref(h)))
        return cls(h.value)

    def search_top(self, seqs=None, packed=None, seq_type=abi.PROTEIN, min_k_ratio=0.05, min_k_match=10, max_results=10, flat=True):
        buf, offs = packed if packed is not None else pack_sequences(seqs)
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        out = C.POINTER(abi.BatchTop)()
        if flat:
            abi.check(abi.lib().kaamer_sharded_search_batch_top_flat(self._h, buf.ctypes.data if len(buf) else None, offs.ctypes.data,
                                                                     len(offs) - 1, seq_type, min_k_ratio, min_k_match, max_results, C.byref(out)))
        else:
            bi = abi.BatchIn(buf.ctypes.data if len(buf) else None, offs.ctypes.data, len(offs) - 1, seq_type, 0)
            to = abi.TopnOpts(min_k_ratio, min_k_match, max_results, 0, None, None, 0, 0)
            abi.check(abi.lib().kaamer_sharded_search_batch_top(self._h, C.byref(bi), C.byref(to), C.byref(out)))
        try:
            return TopResult(out)
        finally:
            abi.lib().kaamer_batch_top_free(out)

    def submit_top(self, seqs=None, packed=None, seq_type=abi.PROTEIN, min_k_ratio=0.05, min_k_match=10, max_results=10):
        """kaamer_sharded_submit_batch_top_flat -> a ticket (wait() -> TopResult); up to three calls in flight per handle"""
        buf, offs = packed if packed is not None else pack_sequences(seqs)
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        t = C.c_void_p()
        abi.check(abi.lib().kaamer_sharded_submit_batch_top_flat(self._h, buf.ctypes.data if len(buf) else None, offs.ctypes.data,
                                                                 len(offs) - 1, seq_type, min_k_ratio, min_k_match, max_results, C.byref(t)))
        return TopTicket(t, sharded=True)

    def exchange_info(self):
        """-> dict(block_bytes, need_entries, queries, adaptive) of the last finished call on the handle's first set"""
        out = (C.c_uint64 * 4)()
        abi.check(abi.lib().kaamer_sharded_exchange_info(self._h, out))
        return {"block_bytes": int(out[0]), "need_entries": int(out[1]), "queries": int(out[2]), "adaptive": bool(out[3])}

    def close(self):
        if self._h:
            abi.lib().kaamer_sharded_index_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Replicas:
    """kaamer_index_open_replicas*: the same database resident on several devices of ONE process; chunks of a read set are
    dealt round-robin (kaamer_replica_stream_*), or a whole file is searched (kaamer_search_file)."""

    CHUNK_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(abi.BatchTop))

    def __init__(self, handle):
        self._h = C.c_void_p(handle)

    @classmethod
    def from_image(cls, image, devices):
        dev = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        abi.check(abi.lib().kaamer_index_open_replicas_image(image._h, dev, len(devices), C.byref(h)))
        return cls(h.value)

    @classmethod
    def open(cls, path, devices):
        dev = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        abi.check(abi.lib().kaamer_index_open_replicas(str(path).encode(), dev, len(devices), C.byref(h)))
        return cls(h.value)

    def __len__(self):
        return int(abi.lib().kaamer_replicas_count(self._h))

    def search_file(self, path, fmt="fastq", seq_type=abi.READS, min_k_ratio=0.05, min_k_match=10, max_results=10,
                    chunk_seqs=1 << 20, chunk_bytes=1 << 28, in_flight=0, strict=False, on_chunk=None):
        """kaamer_search_file.  on_chunk(first_seq, reads_handle, TopResult) per chunk, in input order (the reads handle is
        valid during the call: kaamer_reads_* accessors / api._reads_to_arrays).  -> summed counters"""
        err = []

        def cb(user, first, reads, top):
            try:
                if on_chunk is not None:
                    on_chunk(int(first), C.c_void_p(reads), TopResult(top))
                return 0
            except BaseException as e:  # noqa: BLE001  (must not propagate through the C frames)
                err.append(e)
                return 1
        c = abi.Counters()
        fn = self.CHUNK_CB(cb)
        rc = abi.lib().kaamer_search_file(self._h, str(path).encode(), 1 if fmt == "fastq" else 0, int(strict), seq_type, min_k_ratio,
                                          min_k_match, max_results, chunk_seqs, chunk_bytes, in_flight, C.cast(fn, C.c_void_p), None, C.byref(c))
        if err:
            raise err[0]
        abi.check(rc)
        return c.as_dict()

    def close(self):
        if self._h:
            abi.lib().kaamer_replicas_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FullTicket:
    """one full-hit-list batch in flight (kaamer_full_ticket); wait() exactly once, discarded when dropped"""

    def __init__(self, handle):
        self._h = handle

    def wait(self):
        out = C.POINTER(abi.BatchOut)()
        h, self._h = self._h, None
        abi.check(abi.lib().kaamer_wait_batch(h, C.byref(out)))
        return BatchResult(out)

    def discard(self):
        h, self._h = self._h, None
        if h:
            abi.lib().kaamer_full_ticket_discard(h)

    def __del__(self):
        try:
            self.discard()
        except Exception:
            pass


class TopTicket:
    """one batch in flight (kaamer_ticket / kaamer_sharded_ticket); wait() exactly once.  A ticket dropped unwaited is
    discarded (kaamer_ticket_discard): its slot goes back to the pool instead of staying busy for good."""

    def __init__(self, handle, sharded=False):
        self._h, self._sharded = handle, sharded

    def wait(self):
        out = C.POINTER(abi.BatchTop)()
        h, self._h = self._h, None
        L = abi.lib()
        abi.check((L.kaamer_sharded_wait_batch_top if self._sharded else L.kaamer_wait_batch_top)(h, C.byref(out)))
        try:
            return TopResult(out)
        finally:
            L.kaamer_batch_top_free(out)

    def discard(self):
        h, self._h = self._h, None
        if h:
            L = abi.lib()
            (L.kaamer_sharded_ticket_discard if self._sharded else L.kaamer_ticket_discard)(h)

    def __del__(self):
        try:
            self.discard()
        except Exception:
            pass


class TopStream:
    """kaamer_stream_*: a FIFO of batches with fixed options (push chunk i + 1 while chunk i is searched)"""

    def __init__(self, index, seq_type, min_k_ratio, min_k_match, max_results):
        to = abi.TopnOpts(min_k_ratio, min_k_match, max_results, 0, None, None, 0, 0)
        h = C.c_void_p()
        abi.check(abi.lib().kaamer_stream_open(index._h, seq_type, C.byref(to), C.byref(h)))
        self._h, self.index = h, index

    def push(self, buf, offs):
        """-> False when every slot is busy with this stream's own chunks (pop first), True when the chunk was taken"""
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        rc = abi.lib().kaamer_stream_push(self._h, buf.ctypes.data if len(buf) else None, offs.ctypes.data, len(offs) - 1)
        if rc == abi.E_BUSY:
            return False
        abi.check(rc)
        return True

    def pop(self):
        out = C.POINTER(abi.BatchTop)()
        abi.check(abi.lib().kaamer_stream_pop(self._h, C.byref(out)))
        try:
            return TopResult(out)
        finally:
            abi.lib().kaamer_batch_top_free(out)

    @property
    def pending(self):
        return int(abi.lib().kaamer_stream_pending(self._h))

    def close(self):
        if self._h:
            abi.lib().kaamer_stream_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Workspace:
    """Reusable device buffers for the device-resident call."""

    def __init__(self, index, max_seq_bytes, max_seqs, max_queries=0, max_hits=0, g_tier_slots=0, seq_type=abi.PROTEIN,
                 first_pos=0, want_positions=False, max_pos_words=0, compact=False, concurrent_batches=0):
        self.index = index
        self.seq_type = seq_type
        o = abi.WorkspaceOpts(max_seq_bytes, max_seqs, max_queries, max_hits,
                              g_tier_slots, seq_type, first_pos, int(want_positions), int(compact), max_pos_words,
                              int(concurrent_batches), 0)
        h = C.c_void_p()
        abi.check(abi.lib().kaamer_workspace_create(index._h, C.byref(o), C.byref(h)))
        self._h = h

    def search_device(self, d_seqs_ptr, d_offsets_ptr, n_seqs, seq_bytes, seq_type=None, stream=0):
        """Enqueue one batch; pointers are raw device addresses (e.g. tensor.data_ptr()),
        `stream` a raw hipStream_t (e.g. torch.cuda.current_stream().cuda_stream)."""
        if seq_type is None:
            seq_type = self.seq_type
        r = abi.DeviceResult()
        abi.check(abi.lib().kaamer_search_device(self.index._h, self._h, d_seqs_ptr, d_offsets_ptr, n_seqs,
                                                 seq_bytes, seq_type, C.c_void_p(stream), C.byref(r)))
        return r

    def merge_device(self, d_ent_off_ptr, d_pid_ptr, d_km_ptr, d_fp_ptr, n_queries, n_entries, stream=0):
        """Merge partial hit lists (sharded index); same result object as search_device."""
        r = abi.DeviceResult()
        abi.check(abi.lib().kaamer_merge_device(self._h, d_ent_off_ptr, d_pid_ptr, d_km_ptr, d_fp_ptr, n_queries,
                                                n_entries, C.c_void_p(stream), C.byref(r)))
        return r

    def topn_device(self, min_k_ratio=0.05, min_k_match=10, max_results=10, best_start_codon=False,
                    d_size_in_kmer_ptr=None, stream=0, orf_source=None, q_first=0, q_stride=1):
        """sortMapByValue + (SetBestStartCodon) + FilterResults of the last search/merge, on the device;
        orf_source: the workspace whose queries the (merged) results belong to (result i = its query q_first + i q_stride)"""
        o = abi.TopnOpts(min_k_ratio, min_k_match, max_results, int(best_start_codon), d_size_in_kmer_ptr,
                         orf_source._h if orf_source is not None else None, q_first, q_stride)
        r = abi.TopnResult()
        abi.check(abi.lib().kaamer_topn_device(self._h, C.byref(o), C.c_void_p(stream), C.byref(r)))
        return r

    def set_count_stream(self, stream):
        """kaamer_workspace_set_count_stream: the counting stage of this workspace's batches runs on `stream` (raw hipStream_t; 0 / None: off)"""
        abi.check(abi.lib().kaamer_workspace_set_count_stream(self._h, C.c_void_p(stream or 0)))

    @property
    def query_capacity(self):
        return int(abi.lib().kaamer_workspace_query_capacity(self._h))

    def exchange_pack(self, layout, d_send_ptr, stream=0):
        abi.check(abi.lib().kaamer_exchange_pack(self._h, C.byref(layout), d_send_ptr, C.c_void_p(stream)))

    def exchange_stats(self, back=0):
        """-> (merge sequence number, queries of the batch, entries the largest block any pair of ranks needed, overflow)
        of the merge `back` calls ago on this workspace; the same figures on every rank"""
        out = (C.c_uint64 * 4)()
        abi.check(abi.lib().kaamer_exchange_stats(self._h, back, out))
        return int(out[0]), int(out[1]), int(out[2]), int(out[3])

    def exchange_merge(self, layout, d_recv_ptr, stream=0):
        r = abi.DeviceResult()
        abi.check(abi.lib().kaamer_exchange_merge(self._h, C.byref(layout), d_recv_ptr, C.c_void_p(stream), C.byref(r)))
        return r

    def finish(self, stream=0):
        c = abi.Counters()
        abi.check(abi.lib().kaamer_workspace_finish(self._h, C.c_void_p(stream), C.byref(c)))
        return c.as_dict()

    def kernel_ms_sum(self):
        """dict(probe_ms, count_ms, total_ms, calls) summed since reset_timers()."""
        a, b, c, n = C.c_double(), C.c_double(), C.c_double(), C.c_uint32()
        abi.check(abi.lib().kaamer_workspace_kernel_ms_sum(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(n)))
        return dict(probe_ms=float(a.value), count_ms=float(b.value), total_ms=float(c.value), calls=int(n.value))

    def reset_timers(self):
        abi.lib().kaamer_workspace_reset_timers(self._h)

    def set_timing(self, every):
        """0: no kernel timers (default); k: HIP events around the kernels of every k-th call"""
        abi.lib().kaamer_workspace_set_timing(self._h, int(every))

    def close(self):
        if self._h:
            abi.lib().kaamer_workspace_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _reads_to_arrays(h):
    """a kaamer_reads handle -> (seqs u8, offsets u64, size_in_kmer, names bytes, name offsets, plus_strand), copied"""
    L = abi.lib()
    n = L.kaamer_reads_count(h)
    offs = np.ctypeslib.as_array(L.kaamer_reads_offsets(h), shape=(n + 1,)).copy()
    noff = np.ctypeslib.as_array(L.kaamer_reads_name_offsets(h), shape=(n + 1,)).copy()
    seqs = np.ctypeslib.as_array(L.kaamer_reads_seqs(h), shape=(int(offs[n]),)).copy() if offs[n] else np.zeros(0, np.uint8)
    names = bytes(np.ctypeslib.as_array(C.cast(L.kaamer_reads_names(h), C.POINTER(C.c_uint8)), shape=(int(noff[n]),))) if noff[n] else b""
    size = np.ctypeslib.as_array(L.kaamer_reads_size_in_kmer(h), shape=(n,)).copy() if n else np.zeros(0, np.int32)
    plus = np.ctypeslib.as_array(L.kaamer_reads_plus_strand(h), shape=(n,)).copy() if n else np.zeros(0, np.int32)
    return seqs, offs, size, names, noff, plus


class Reader:
    """kaamer_reader_*: GetQueriesFasta / GetQueriesFastq over a file in chunks (gzip inflated incrementally)"""

    def __init__(self, path=None, fmt="fastq", strict=False, fd=None):
        h = C.c_void_p()
        f = 1 if fmt == "fastq" else 0
        if fd is not None:
            abi.check(abi.lib().kaamer_reader_open_fd(fd, f, int(strict), C.byref(h)))
        else:
            abi.check(abi.lib().kaamer_reader_open(str(path).encode(), f, int(strict), C.byref(h)))
        self._h = h

    def next_handle(self, max_seqs, max_bytes):
        """-> a raw kaamer_reads handle (free with kaamer_reads_free), or None at the end"""
        r = C.c_void_p()
        abi.check(abi.lib().kaamer_reader_next(self._h, max_seqs, max_bytes, C.byref(r)))
        if abi.lib().kaamer_reads_count(r) == 0 and self.done:
            abi.lib().kaamer_reads_free(r)
            return None
        return r

    def next(self, max_seqs=1 << 20, max_bytes=1 << 28):
        """-> (seqs, offsets, size_in_kmer, names, name_offsets, plus_strand) of the next chunk, or None at the end"""
        r = self.next_handle(max_seqs, max_bytes)
        if r is None:
            return None
        try:
            return _reads_to_arrays(r)
        finally:
            abi.lib().kaamer_reads_free(r)

    @property
    def done(self):
        return bool(abi.lib().kaamer_reader_done(self._h))

    @property
    def records(self):
        return int(abi.lib().kaamer_reader_records(self._h))

    def records_list(self, max_seqs=1 << 20, max_bytes=1 << 28):
        """every record of the file as parse_reads returns them (tests)"""
        out = []
        while True:
            c = self.next(max_seqs, max_bytes)
            if c is None:
                return out
            seqs, offs, size, names, noff, plus = c
            sb = bytes(seqs)
            out += [dict(seq=sb[int(offs[i]):int(offs[i + 1])].decode("latin-1"), name=names[int(noff[i]):int(noff[i + 1])].decode("latin-1"),
                         size=int(size[i]), plus=bool(plus[i])) for i in range(len(size))]

    def close(self):
        if self._h:
            abi.lib().kaamer_reader_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def parse_reads(text, fmt="fasta"):
    """GetQueriesFasta / GetQueriesFastq on a text buffer -> list of dict(seq, name, size, plus)"""
    data = text.encode("latin-1") if isinstance(text, str) else bytes(text)
    h = C.c_void_p()
    f = abi.lib().kaamer_parse_fasta if fmt == "fasta" else abi.lib().kaamer_parse_fastq
    abi.check(f(data, len(data), C.byref(h)))
    try:
        L = abi.lib()
        n = L.kaamer_reads_count(h)
        offs = np.ctypeslib.as_array(L.kaamer_reads_offsets(h), shape=(n + 1,)).copy()
        noff = np.ctypeslib.as_array(L.kaamer_reads_name_offsets(h), shape=(n + 1,)).copy()
        seqs = bytes(np.ctypeslib.as_array(L.kaamer_reads_seqs(h), shape=(int(offs[n]),))) if offs[n] else b""
        names = bytes(np.ctypeslib.as_array(C.cast(L.kaamer_reads_names(h), C.POINTER(C.c_uint8)), shape=(int(noff[n]),))) if noff[n] else b""
        size = np.ctypeslib.as_array(L.kaamer_reads_size_in_kmer(h), shape=(n,)).copy() if n else np.zeros(0, np.int32)
        plus = np.ctypeslib.as_array(L.kaamer_reads_plus_strand(h), shape=(n,)).copy() if n else np.zeros(0, np.int32)
        return [dict(seq=seqs[int(offs[i]):int(offs[i + 1])].decode("latin-1"),
                     name=names[int(noff[i]):int(noff[i + 1])].decode("latin-1"), size=int(size[i]), plus=bool(plus[i]))
                for i in range(n)]
    finally:
        abi.lib().kaamer_reads_free(h)


def align_pairs(seqs=None, packed=None, pairs=(), number_of_aa=0, sub_matrix="blosum62", gap_open=11, gap_extend=1, device=0):
    """kaamer_align_pairs: align.Align (align.go:46-161) for every (query index, subject index) pair -> list of dicts
    (None where the reference keeps an empty AlignmentResult: "No matrix found"; a ValueError-like status is returned as
    {"status": n})"""
    buf, offs = packed if packed is not None else pack_sequences(seqs)
    buf = np.ascontiguousarray(buf, dtype=np.uint8)
    offs = np.ascontiguousarray(offs, dtype=np.uint64)
    pq = np.ascontiguousarray([p[0] for p in pairs], dtype=np.uint32)
    ps = np.ascontiguousarray([p[1] for p in pairs], dtype=np.uint32)
    h = C.c_void_p()
    L = abi.lib()
    abi.check(L.kaamer_align_pairs(device, buf.ctypes.data if len(buf) else None, offs.ctypes.data, len(offs) - 1,
                                   pq.ctypes.data if len(pq) else None, ps.ctypes.data if len(ps) else None, len(pq), int(number_of_aa),
                                   sub_matrix.encode(), gap_open, gap_extend, C.byref(h)))
    try:
        n = L.kaamer_alignments_count(h)
        items = L.kaamer_alignments_items(h)
        text = L.kaamer_alignments_text(h)
        out = []
        for i in range(n):
            a = items[i]
            if a.status == 1:
                out.append(None)
                continue
            d = {k: getattr(a, k) for k, _ in abi.Alignment._fields_ if k not in ("aln_off", "reserved")}
            ln, off = a.length, a.aln_off
            raw = C.string_at(C.addressof(text.contents) + off, 3 * ln) if ln else b""
            d["aln"] = (raw[:ln].decode("latin-1"), raw[ln:2 * ln].decode("latin-1"), raw[2 * ln:].decode("latin-1"))
            out.append(d)
        return out
    finally:
        L.kaamer_alignments_free(h)
