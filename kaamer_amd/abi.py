"""ctypes binding of libkaamer_hip.so — one Python declaration per symbol of
include/kaamer_hip.h.  Loading fails loudly if the library has not been built
(`python -m kaamer_amd.build`); there is no fallback implementation.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# KAAMER_LIB points at an alternative build of the same library (tuning experiments only)
LIB_PATH = os.environ.get("KAAMER_LIB") or os.path.join(_HERE, "libkaamer_hip.so")

OK, E_ARG, E_IO, E_NOMEM, E_HIP, E_CAPACITY, E_FORMAT, E_BUSY = 0, -1, -2, -3, -4, -5, -6, -7
NUCLEOTIDE, PROTEIN, READS = 0, 1, 2


class KaamerError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("kaamer_hip error %d: %s" % (code, msg))
        self.code = code


class Pair(C.Structure):
    _fields_ = [("key", C.c_uint32), ("protein_id", C.c_uint32)]


class ImageStats(C.Structure):
    _fields_ = [("n_pairs", C.c_uint64), ("n_keys", C.c_uint64), ("n_buckets", C.c_uint64),
                ("arena_words", C.c_uint64), ("n_inline", C.c_uint64), ("n_lists", C.c_uint64),
                ("max_list", C.c_uint64), ("n_displaced", C.c_uint64), ("shard", C.c_uint32),
                ("n_shards", C.c_uint32), ("max_protein_id", C.c_uint32), ("reserved", C.c_uint32)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_ if k != "reserved"}


class QueryMeta(C.Structure):
    _fields_ = [("src_seq", C.c_uint32), ("size_in_kmer", C.c_int32), ("start_position", C.c_int32),
                ("end_position", C.c_int32), ("plus_strand", C.c_int32), ("aa_len", C.c_uint32),
                ("aa_off", C.c_uint64), ("sa_off", C.c_uint32), ("sa_len", C.c_uint32)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("n_in", "n_queries", "n_lookup", "n_probe", "n_found",
                                          "n_post", "n_hits", "n_overflow", "n_lists", "n_list_ids")]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class BatchIn(C.Structure):
    _fields_ = [("seqs", C.c_void_p), ("offsets", C.c_void_p), ("n_seqs", C.c_uint32),
                ("seq_type", C.c_int32), ("want_positions", C.c_int32)]


class BatchOut(C.Structure):
    _fields_ = [("n_queries", C.c_uint32), ("q", C.POINTER(QueryMeta)),
                ("hit_off", C.POINTER(C.c_uint64)), ("hit_cnt", C.POINTER(C.c_uint32)), ("hit_pid", C.POINTER(C.c_uint32)),
                ("hit_kmatch", C.POINTER(C.c_uint32)), ("hit_first_pos", C.POINTER(C.c_uint32)),
                ("pos_off", C.POINTER(C.c_uint64)), ("pos_bits", C.POINTER(C.c_uint64)),
                ("orf_aa", C.POINTER(C.c_uint8)), ("starts_alt", C.POINTER(C.c_int32)),
                ("counters", Counters)]


class WorkspaceOpts(C.Structure):
    _fields_ = [("max_seq_bytes", C.c_uint64), ("max_seqs", C.c_uint32), ("max_queries", C.c_uint32),
                ("max_hits", C.c_uint64),
                ("g_tier_slots", C.c_uint64), ("seq_type", C.c_int32), ("first_pos", C.c_uint32), ("want_positions", C.c_uint32),
                ("compact", C.c_uint32), ("max_pos_words", C.c_uint64), ("concurrent_batches", C.c_uint32), ("reserved", C.c_uint32)]


class DeviceResult(C.Structure):
    _fields_ = [("n_queries_cap", C.c_uint32), ("d_n_queries", C.c_void_p), ("d_q", C.c_void_p),
                ("d_hit_off", C.c_void_p), ("d_hit_cnt", C.c_void_p), ("hit_capacity", C.c_uint64), ("d_hit_pid", C.c_void_p),
                ("d_hit_kmatch", C.c_void_p),
                ("d_hit_first_pos", C.c_void_p), ("d_orf_aa", C.c_void_p), ("d_starts_alt", C.c_void_p),
                ("d_counters", C.c_void_p), ("d_pos_off", C.c_void_p), ("d_pos_bits", C.c_void_p),
                ("d_pos_base", C.c_void_p)]


class TopnOpts(C.Structure):
    _fields_ = [("min_k_ratio", C.c_double), ("min_k_match", C.c_int64), ("max_results", C.c_uint32),
                ("best_start_codon", C.c_uint32), ("d_size_in_kmer", C.c_void_p),
                ("orf_source", C.c_void_p), ("q_first", C.c_uint32), ("q_stride", C.c_uint32)]


class ExchangeLayout(C.Structure):
    _fields_ = [("world", C.c_uint32), ("rank", C.c_uint32), ("q_cap", C.c_uint32), ("arrays", C.c_uint32),
                ("e_cap", C.c_uint64), ("block_words", C.c_uint64)]


class BatchTop(C.Structure):
    _fields_ = [("n_queries", C.c_uint32), ("n_reported", C.c_uint32), ("max_results", C.c_uint32),
                ("rep_query", C.POINTER(C.c_uint32)), ("q", C.POINTER(QueryMeta)), ("trim", C.POINTER(C.c_int32)),
                ("top_off", C.POINTER(C.c_uint64)), ("top_pid", C.POINTER(C.c_uint32)),
                ("top_kmatch", C.POINTER(C.c_uint32)), ("top_first_pos", C.POINTER(C.c_uint32)),
                ("orf_aa", C.POINTER(C.c_uint8)), ("counters", Counters)]


class ProteinEntry(C.Structure):
    _fields_ = [("found", C.c_uint32), ("length", C.c_uint32), ("entry_id", C.c_void_p), ("entry_id_len", C.c_uint32),
                ("n_features", C.c_uint32), ("sequence", C.c_void_p), ("features", C.c_void_p), ("feature_off", C.POINTER(C.c_uint64)),
                ("sequence_len", C.c_uint32), ("reserved", C.c_uint32)]


class Alignment(C.Structure):
    _fields_ = [("identity", C.c_float), ("similarity", C.c_float), ("length", C.c_int32), ("mismatches", C.c_int32),
                ("gap_openings", C.c_int32), ("raw", C.c_int32), ("bitscore", C.c_double), ("evalue", C.c_double),
                ("query_start", C.c_int32), ("query_end", C.c_int32), ("subject_start", C.c_int32), ("subject_end", C.c_int32),
                ("aln_off", C.c_uint64), ("status", C.c_int32), ("reserved", C.c_int32)]


class TopnResult(C.Structure):
    _fields_ = [("max_results", C.c_uint32), ("d_top_cnt", C.c_void_p), ("d_top_pid", C.c_void_p),
                ("d_top_kmatch", C.c_void_p), ("d_top_first_pos", C.c_void_p), ("d_trim", C.c_void_p),
                ("d_start_position", C.c_void_p), ("d_size_in_kmer", C.c_void_p)]


# every symbol include/kaamer_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "kaamer_last_error": (C.c_char_p, []),
    "kaamer_abi_version": (C.c_int, []),
    "kaamer_encode_kmer": (C.c_uint32, [C.c_char_p]),
    "kaamer_shard_of": (C.c_uint32, [C.c_uint32, C.c_uint32]),
    "kaamer_image_build_pairs": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_double,
                                           C.POINTER(C.c_void_p)]),
    "kaamer_image_build_proteins": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                              C.c_uint32, C.c_double, C.POINTER(C.c_void_p)]),
    "kaamer_image_build_proteins_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                                     C.c_uint32, C.c_double, C.c_int, C.POINTER(C.c_void_p)]),
    "kaamer_index_build_proteins": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                              C.c_uint32, C.c_double, C.c_int, C.POINTER(C.c_void_p)]),
    "kaamer_image_save": (C.c_int, [C.c_void_p, C.c_char_p]),
    "kaamer_image_load": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "kaamer_image_get_stats": (C.c_int, [C.c_void_p, C.POINTER(ImageStats)]),
    "kaamer_image_free": (None, [C.c_void_p]),
    "kaamer_image_get": (C.c_uint32, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]),
    "kaamer_makedb_fasta": (C.c_int, [C.c_char_p, C.c_uint64, C.POINTER(C.c_void_p)]),
    "kaamer_makedb_tsv": (C.c_int, [C.c_char_p, C.c_uint64, C.POINTER(C.c_void_p)]),
    "kaamer_makedb_embl": (C.c_int, [C.c_char_p, C.c_uint64, C.POINTER(C.c_void_p)]),
    "kaamer_makedb_gbk": (C.c_int, [C.c_char_p, C.c_uint64, C.POINTER(C.c_void_p)]),
    "kaamer_makedb_text": (C.c_int, [C.c_char_p, C.c_uint64, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "kaamer_proteins_count": (C.c_uint32, [C.c_void_p]),
    "kaamer_proteins_ids": (C.POINTER(C.c_uint32), [C.c_void_p]),
    "kaamer_proteins_seqs": (C.POINTER(C.c_uint8), [C.c_void_p]),
    "kaamer_proteins_offsets": (C.POINTER(C.c_uint64), [C.c_void_p]),
    "kaamer_proteins_n_features": (C.c_uint32, [C.c_void_p]),
    "kaamer_proteins_feature_name": (C.c_char_p, [C.c_void_p, C.c_uint32]),
    "kaamer_proteins_stats": (None, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "kaamer_proteins_save": (C.c_int, [C.c_void_p, C.c_char_p]),
    "kaamer_proteins_load": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "kaamer_proteins_free": (None, [C.c_void_p]),
    "kaamer_image_build_makedb": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_double, C.POINTER(C.c_void_p)]),
    "kaamer_image_build_makedb_device": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_double, C.c_int,
                                                   C.POINTER(C.c_void_p)]),
    "kaamer_fetch_hits": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(ProteinEntry)]),
    "kaamer_index_open_image": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "kaamer_index_open": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]),
    "kaamer_index_close": (None, [C.c_void_p]),
    "kaamer_index_get_stats": (C.c_int, [C.c_void_p, C.POINTER(ImageStats)]),
    "kaamer_search_batch": (C.c_int, [C.c_void_p, C.POINTER(BatchIn), C.POINTER(C.POINTER(BatchOut))]),
    "kaamer_batch_free": (None, [C.POINTER(BatchOut)]),
    "kaamer_submit_batch_flat": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "kaamer_wait_batch": (C.c_int, [C.c_void_p, C.POINTER(C.POINTER(BatchOut))]),
    "kaamer_full_ticket_discard": (None, [C.c_void_p]),
    "kaamer_workspace_create": (C.c_int, [C.c_void_p, C.POINTER(WorkspaceOpts), C.POINTER(C.c_void_p)]),
    "kaamer_workspace_free": (None, [C.c_void_p]),
    "kaamer_workspace_set_count_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "kaamer_search_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                       C.c_uint64, C.c_int32, C.c_void_p, C.POINTER(DeviceResult)]),
    "kaamer_merge_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64,
                                      C.c_void_p, C.POINTER(DeviceResult)]),
    "kaamer_exchange_layout_init": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.POINTER(ExchangeLayout)]),
    "kaamer_workspace_query_capacity": (C.c_uint32, [C.c_void_p]),
    "kaamer_exchange_layout_fit": (C.c_int, [C.POINTER(ExchangeLayout), C.c_uint32, C.c_uint64, C.c_int32, C.POINTER(ExchangeLayout)]),
    "kaamer_exchange_stats": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64)]),
    "kaamer_exchange_pack": (C.c_int, [C.c_void_p, C.POINTER(ExchangeLayout), C.c_void_p, C.c_void_p]),
    "kaamer_exchange_merge": (C.c_int, [C.c_void_p, C.POINTER(ExchangeLayout), C.c_void_p, C.c_void_p, C.POINTER(DeviceResult)]),
    "kaamer_rccl_alltoall": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p]),
    "kaamer_workspace_finish": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(Counters)]),
    "kaamer_workspace_kernel_ms_sum": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                                 C.POINTER(C.c_double), C.POINTER(C.c_uint32)]),
    "kaamer_topn_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "kaamer_search_batch_top": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "kaamer_batch_top_free": (None, [C.c_void_p]),
    "kaamer_index_open_sharded": (C.c_int, [C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.c_uint32, C.POINTER(C.c_void_p)]),
    "kaamer_index_open_sharded_images": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_uint32, C.POINTER(C.c_void_p)]),
    "kaamer_sharded_index_shards": (C.c_uint32, [C.c_void_p]),
    "kaamer_sharded_index_close": (None, [C.c_void_p]),
    "kaamer_sharded_search_batch_top": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "kaamer_submit_batch_top": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "kaamer_wait_batch_top": (C.c_int, [C.c_void_p, C.c_void_p]),
    "kaamer_stream_open": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(C.c_void_p)]),
    "kaamer_stream_push": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]),
    "kaamer_stream_pop": (C.c_int, [C.c_void_p, C.c_void_p]),
    "kaamer_stream_pending": (C.c_uint32, [C.c_void_p]),
    "kaamer_stream_close": (None, [C.c_void_p]),
    "kaamer_ticket_discard": (None, [C.c_void_p]),
    # cgo-safe forms: (handle, seqs, offsets, n_seqs, seq_type, min_k_ratio, min_k_match, max_results, out)
    "kaamer_search_batch_top_flat": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int32, C.c_double,
                                               C.c_int64, C.c_uint32, C.c_void_p]),
    "kaamer_submit_batch_top_flat": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int32, C.c_double,
                                               C.c_int64, C.c_uint32, C.POINTER(C.c_void_p)]),
    "kaamer_search_batch_flat": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int32, C.c_int32,
                                           C.POINTER(C.POINTER(BatchOut))]),
    "kaamer_stream_open_flat": (C.c_int, [C.c_void_p, C.c_int32, C.c_double, C.c_int64, C.c_uint32, C.POINTER(C.c_void_p)]),
    "kaamer_sharded_submit_batch_top": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "kaamer_sharded_wait_batch_top": (C.c_int, [C.c_void_p, C.c_void_p]),
    "kaamer_sharded_ticket_discard": (None, [C.c_void_p]),
    "kaamer_sharded_search_batch_top_flat": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int32, C.c_double,
                                                       C.c_int64, C.c_uint32, C.c_void_p]),
    "kaamer_sharded_submit_batch_top_flat": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int32, C.c_double,
                                                       C.c_int64, C.c_uint32, C.POINTER(C.c_void_p)]),
    "kaamer_sharded_exchange_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "kaamer_workspace_reset_timers": (None, [C.c_void_p]),
    "kaamer_workspace_set_timing": (None, [C.c_void_p, C.c_uint32]),
    "kaamer_filter_results": (C.c_int64, [C.c_void_p, C.c_int64, C.c_int32, C.c_double, C.c_int64, C.c_int64]),
    "kaamer_sort_hits": (None, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "kaamer_set_best_start_codon": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int32,
                                                C.c_char_p, C.c_uint32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "kaamer_parse_fasta": (C.c_int, [C.c_char_p, C.c_uint64, C.POINTER(C.c_void_p)]),
    "kaamer_parse_fastq": (C.c_int, [C.c_char_p, C.c_uint64, C.POINTER(C.c_void_p)]),
    "kaamer_reads_count": (C.c_uint32, [C.c_void_p]),
    "kaamer_reads_seqs": (C.POINTER(C.c_uint8), [C.c_void_p]),
    "kaamer_reads_offsets": (C.POINTER(C.c_uint64), [C.c_void_p]),
    "kaamer_reads_size_in_kmer": (C.POINTER(C.c_int32), [C.c_void_p]),
    "kaamer_reads_names": (C.POINTER(C.c_char), [C.c_void_p]),
    "kaamer_reads_name_offsets": (C.POINTER(C.c_uint64), [C.c_void_p]),
    "kaamer_reads_plus_strand": (C.POINTER(C.c_int32), [C.c_void_p]),
    "kaamer_reads_free": (None, [C.c_void_p]),
    "kaamer_index_open_replicas": (C.c_int, [C.c_char_p, C.POINTER(C.c_int), C.c_uint32, C.POINTER(C.c_void_p)]),
    "kaamer_index_open_replicas_image": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.c_uint32, C.POINTER(C.c_void_p)]),
    "kaamer_replicas_count": (C.c_uint32, [C.c_void_p]),
    "kaamer_replicas_index": (C.c_void_p, [C.c_void_p, C.c_uint32]),
    "kaamer_replicas_close": (None, [C.c_void_p]),
    "kaamer_replica_stream_open_flat": (C.c_int, [C.c_void_p, C.c_int32, C.c_double, C.c_int64, C.c_uint32, C.POINTER(C.c_void_p)]),
    "kaamer_replica_stream_push": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]),
    "kaamer_replica_stream_pop": (C.c_int, [C.c_void_p, C.c_void_p]),
    "kaamer_replica_stream_pending": (C.c_uint32, [C.c_void_p]),
    "kaamer_replica_stream_close": (None, [C.c_void_p]),
    "kaamer_search_file": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int32, C.c_double, C.c_int64, C.c_uint32, C.c_uint32,
                                     C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(Counters)]),
    "kaamer_align_pairs": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64,
                                     C.c_char_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "kaamer_alignments_count": (C.c_uint32, [C.c_void_p]),
    "kaamer_alignments_items": (C.POINTER(Alignment), [C.c_void_p]),
    "kaamer_alignments_text": (C.POINTER(C.c_char), [C.c_void_p]),
    "kaamer_alignments_free": (None, [C.c_void_p]),
    "kaamer_align_matrix_scores": (C.c_int, [C.c_char_p, C.c_int32, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "kaamer_align_matrix_entry": (C.c_int32, [C.c_int32, C.c_int32]),
    "kaamer_reader_open": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "kaamer_reader_open_fd": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "kaamer_reader_next": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint64, C.POINTER(C.c_void_p)]),
    "kaamer_reader_done": (C.c_int, [C.c_void_p]),
    "kaamer_reader_records": (C.c_uint64, [C.c_void_p]),
    "kaamer_reader_close": (None, [C.c_void_p]),
}

_lib = None


def lib():
    """Loads libkaamer_hip.so.  `import torch` first when torch is used in the same
    process, so both resolve to ONE HIP runtime (same libamdhip64 SONAME)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libkaamer_hip.so is not built: run `python -m kaamer_amd.build` "
                          "(expected at %s); there is no CPU fallback" % LIB_PATH)
    try:
        import torch  # noqa: F401  (loads torch's libamdhip64 first when torch is installed)
    except Exception:
        pass
    L = C.CDLL(LIB_PATH)
    host_only = bool(os.environ.get("KAAMER_HOST_ONLY"))  # the sanitized CPU build of the host sources (tools/asan)
    for name, (res, args) in SYMBOLS.items():
        if host_only and not hasattr(L, name):
            continue
        f = getattr(L, name)  # AttributeError if the library lacks a declared symbol
        f.restype = res
        f.argtypes = args
    _lib = L
    return L


def check(rc):
    if rc != OK:
        raise KaamerError(rc, (lib().kaamer_last_error() or b"").decode("utf-8", "replace"))
