/*
 * kaamer_hip.h — C ABI of libkaamer_hip.so, the MI355X (gfx950) k-mer search
 * path for kaamer.
 *
 * This is the drop-in boundary.  The reference (zorino/kaamer) has no FFI; the
 * seam is the Go function boundary between the search drivers and the hot
 * path.  Each entry point cites the reference code it replaces (paths under
 * the reference tree).  A cgo shim binds exactly these symbols — see
 * INTEGRATION.md.
 *
 * Conventions
 *   - every call returns KAAMER_OK (0) or a negative kaamer_status; nothing
 *     throws or aborts (the reference log.Fatal()s);
 *   - kaamer_last_error() returns a thread-local message for the last failure;
 *   - input pointers are borrowed for the duration of the call only (cgo rule);
 *   - outputs of the host-buffer calls are library-owned until the matching
 *     *_free call;
 *   - plain pointers and sizes only, no C++ or torch types;
 *   - cgo safety: NO STRUCT PASSED TO THE LIBRARY NEEDS TO CONTAIN A CALLER POINTER.  Every host-buffer call has a
 *     *_flat form that takes the caller's buffers as direct arguments (cgo pins a Go pointer passed as an argument
 *     for the duration of the call; a Go pointer stored inside a struct that is itself passed by pointer is a
 *     run-time panic, "cgo argument has Go pointer to unpinned Go pointer").  The struct forms (kaamer_batch_in) are
 *     for C / C++ / ctypes callers.  kaamer_topn_opts' pointer fields are device / library handles (NULL from a host
 *     caller), never caller memory.
 */
#ifndef KAAMER_HIP_H
#define KAAMER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KAAMER_ABI_VERSION 4
#define KAAMER_KMER_SIZE 7 /* pkg/search/search.go:45, pkg/makedb/makedb.go:30 */

typedef enum {
    KAAMER_OK = 0,
    KAAMER_E_ARG = -1,       /* bad argument                                   */
    KAAMER_E_IO = -2,        /* file open/read/write/format                    */
    KAAMER_E_NOMEM = -3,     /* host or device allocation failed               */
    KAAMER_E_HIP = -4,       /* a HIP runtime call failed (no device, ...)     */
    KAAMER_E_CAPACITY = -5,  /* a workspace bound was exceeded; enlarge, retry */
    KAAMER_E_FORMAT = -6,    /* index image/version mismatch                   */
    KAAMER_E_BUSY = -7       /* every host slot holds a batch: wait / pop first */
} kaamer_status;

/* search.go:41-44 */
enum { KAAMER_NUCLEOTIDE = 0, KAAMER_PROTEIN = 1, KAAMER_READS = 2 };

const char *kaamer_last_error(void);
int kaamer_abi_version(void);

typedef struct kaamer_reads kaamer_reads;   /* parsed query records: "Readers" at the end of this file */

/* ------------------------------------------------------------------------- */
/* K-mer codec (host) — replaces K_.EncodeKmer / CreateBytesKey,              */
/* pkg/kvstore/k_store.go:66-76,91-117.                                       */
/* ------------------------------------------------------------------------- */
uint32_t kaamer_encode_kmer(const uint8_t kmer[KAAMER_KMER_SIZE]);

/* ------------------------------------------------------------------------- */
/* Builder — the GPU-layout emitter beside pkg/makedb + pkg/indexdb.           */
/*   pairs  = what makedb's emit loops write into kmer_store                   */
/*            (inputFASTA.go:245-248, inputTSV.go:236-239, inputEMBL.go:309-312,*/
/*            inputGBK.go:296-299): one (key, proteinId) per 7-mer window;     */
/*   build  = indexdb.IndexStore/KeyToList (indexdb.go:68-132) +               */
/*            RemoveDuplicatesFromSlice (kv_store.go:284-305): per key the     */
/*            unique id set; identical sets are stored once (the KComb idea,   */
/*            kcomb_store.go:42-85).                                           */
/* An image is the host copy of one shard's table (buckets + postings arena).  */
/* ------------------------------------------------------------------------- */
typedef struct { uint32_t key; uint32_t protein_id; } kaamer_pair;
typedef struct kaamer_image kaamer_image;

typedef struct {
    uint64_t n_pairs;      /* unique (key,id) pairs in this shard              */
    uint64_t n_keys;       /* distinct keys in this shard                      */
    uint64_t n_buckets;    /* 64-byte buckets                                  */
    uint64_t arena_words;  /* u32 words of the postings arena                  */
    uint64_t n_inline;     /* keys whose single id is stored in the slot       */
    uint64_t n_lists;      /* distinct postings lists in the arena             */
    uint64_t max_list;     /* longest postings list                            */
    uint64_t n_displaced;  /* keys not in their home bucket                    */
    uint32_t shard, n_shards;
    uint32_t max_protein_id;
    uint32_t reserved;
} kaamer_image_stats;

/* shard s of n_shards holds the keys with kaamer_shard_of(key) == s */
uint32_t kaamer_shard_of(uint32_t key, uint32_t n_shards);

int kaamer_image_build_pairs(const kaamer_pair *pairs, uint64_t n, uint32_t shard,
                             uint32_t n_shards, double load_factor, kaamer_image **out);
/* Emit loop + build: every window seqs[off[p]+i .. +7), i in [0,len-7], of
 * every protein with len >= 7, under id ids[p] (ids == NULL: the TSV rule,
 * 0-based running index, inputTSV.go:141-142). */
int kaamer_image_build_proteins(const uint8_t *seqs, const uint64_t *offsets,
                                const uint32_t *ids, uint32_t n_proteins, uint32_t shard,
                                uint32_t n_shards, double load_factor, kaamer_image **out);
/* The same build on the device (builder_device.hip): emit, radix sort, unique,
 * set sharing and bucket placement run on `device`; the image that comes back
 * is byte-identical to kaamer_image_build_proteins' (pkg/indexdb/indexdb.go:68-132,
 * pkg/kvstore/kcomb_store.go:42-85 at DB-UR shard size).  Inputs are host buffers. */
int kaamer_image_build_proteins_device(const uint8_t *seqs, const uint64_t *offsets,
                                       const uint32_t *ids, uint32_t n_proteins, uint32_t shard,
                                       uint32_t n_shards, double load_factor, int device,
                                       kaamer_image **out);
int kaamer_image_save(const kaamer_image *img, const char *path);
int kaamer_image_load(const char *path, kaamer_image **out);
int kaamer_image_get_stats(const kaamer_image *img, kaamer_image_stats *out);
void kaamer_image_free(kaamer_image *img);
/* host-side point read of an image (builder self-check; not the search path):
 * returns the number of ids under `key`, copies up to cap of them */
uint32_t kaamer_image_get(const kaamer_image *img, uint32_t key, uint32_t *ids, uint32_t cap);

/* ------------------------------------------------------------------------- */
/* makedb — which proteins a database file contributes and under which ids      */
/* (pkg/makedb/inputFASTA.go:95-124,191-250; inputTSV.go:92-142), kept with     */
/* their annotations: the table FetchHitsInformation reads                      */
/* (pkg/search/search.go:454-470, pkg/kvstore/protein.proto) instead of one     */
/* ProteinStore point read per hit.  Text is taken already decompressed.        */
/*   FASTA: record k (1-based) gets id k+1, the last two records share id N     */
/*          (reference behaviour); names with ", partial" and sequences shorter */
/*          than 7 are dropped; sequences are upper-cased; feature: ProteinName */
/*   TSV:   header row with EntryID and Sequence columns (any case); accepted   */
/*          rows get 0-based ids; sequences are NOT upper-cased; every other    */
/*          column is a feature                                                 */
/* ------------------------------------------------------------------------- */
typedef struct kaamer_proteins kaamer_proteins;
int kaamer_makedb_fasta(const char *text, uint64_t len, kaamer_proteins **out);
int kaamer_makedb_tsv(const char *text, uint64_t len, kaamer_proteins **out);
/*   EMBL:  UniProt flat file (pkg/makedb/inputEMBL.go:46-314).  Records end at a line "//"; record k (1-based
 *          ordinal of its "//", empty records use up a number) gets id k.  EntryId = first field of the ID line,
 *          features ProteinName (DE RecName / SubName, " {...};" evidence removed), GeneName, EC, GO, KEGG_ID,
 *          BioCyc_ID, HAMAP, Organism, TaxId (the reference's [12:] drops the id's first digit: kept),
 *          FullTaxonomy.  "Flags: Fragment;" entries are dropped.  The length is the one DECLARED on the SQ line:
 *          entries declaring < 7 are dropped and the k-mers are those of Sequence[:Length].  Not upper-cased, no
 *          ", partial" filter.
 *   GBK:   GenPept flat file (inputGBK.go:45-301).  Same record / id rule.  DEFINITION -> ProteinName (a trailing
 *          " [organism]." removed), VERSION -> EntryId, ORGANISM -> Organism + FullTaxonomy, ORIGIN -> the sequence,
 *          upper-cased; names containing ", partial" and sequences shorter than 7 are dropped.
 *   Where the Go code would panic on a malformed entry (a tag line shorter than the column it slices at, an SQ
 *   length beyond the sequence) the entry is dropped; the split-build options -offset / -length are not reproduced. */
int kaamer_makedb_embl(const char *text, uint64_t len, kaamer_proteins **out);
int kaamer_makedb_gbk(const char *text, uint64_t len, kaamer_proteins **out);
/* The four readers behind one entry point (format: 0 FASTA, 1 TSV, 2 EMBL, 3 GBK), with the reference's scanner limit on
 * request: strict_scanner = 1 ends the input at a line of 1 048 576 bytes or more, as bufio.Scanner with a 1 MiB buffer does
 * (inputFASTA.go:88-89 and the same two lines in the other readers): what was read before the line is processed as if the
 * file ended there.  strict_scanner = 0 (what the four functions above do) reads lines of any length. */
int kaamer_makedb_text(const char *text, uint64_t len, int32_t format, int32_t strict_scanner, kaamer_proteins **out);
uint32_t kaamer_proteins_count(const kaamer_proteins *p);           /* accepted proteins       */
const uint32_t *kaamer_proteins_ids(const kaamer_proteins *p);      /* their protein ids       */
const uint8_t *kaamer_proteins_seqs(const kaamer_proteins *p);      /* packed sequences        */
const uint64_t *kaamer_proteins_offsets(const kaamer_proteins *p);  /* count + 1               */
uint32_t kaamer_proteins_n_features(const kaamer_proteins *p);      /* KStats.Features         */
const char *kaamer_proteins_feature_name(const kaamer_proteins *p, uint32_t i);
void kaamer_proteins_stats(const kaamer_proteins *p, uint64_t out[3]); /* KStats: NumberOfProteins, NumberOfAA, NumberOfKmers */
int kaamer_proteins_save(const kaamer_proteins *p, const char *path);
int kaamer_proteins_load(const char *path, kaamer_proteins **out);
void kaamer_proteins_free(kaamer_proteins *p);
/* emit loops + indexdb collapse over the accepted proteins under their ids -> one shard's image */
int kaamer_image_build_makedb(const kaamer_proteins *p, uint32_t shard, uint32_t n_shards, double load_factor,
                              kaamer_image **out);
/* the same on `device` (kaamer_image_build_proteins_device) */
int kaamer_image_build_makedb_device(const kaamer_proteins *p, uint32_t shard, uint32_t n_shards,
                                     double load_factor, int device, kaamer_image **out);
/* One Protein entry (protein.proto).  Pointers go into the table and stay valid until it is freed;
 * feature i is features[feature_off[i] .. feature_off[i+1]). */
typedef struct {
    uint32_t found;             /* 0: no protein under this id (entry zeroed) */
    uint32_t length;            /* Protein.Length (EMBL: the length the SQ line */
                                /* declares; what was indexed)                 */
    const char *entry_id;       /* Protein.EntryId, entry_id_len bytes         */
    uint32_t entry_id_len;
    uint32_t n_features;
    const uint8_t *sequence;    /* Protein.Sequence, sequence_len bytes        */
    const char *features;
    const uint64_t *feature_off; /* n_features + 1                             */
    uint32_t sequence_len;      /* = length, except for an EMBL entry that     */
                                /* holds more residues than its SQ line        */
                                /* declares: the reference indexes             */
                                /* Sequence[:Length] and stores all of it      */
                                /* (inputEMBL.go:293-312)                      */
    uint32_t reserved;
} kaamer_protein_entry;
/* FetchHitsInformation (search.go:454-470) for n protein ids at once */
int kaamer_fetch_hits(const kaamer_proteins *p, const uint32_t *ids, uint32_t n, kaamer_protein_entry *out);

/* ------------------------------------------------------------------------- */
/* Index handle — replaces the read side of kvstore.KVStoresNew               */
/* (kv_stores.go:46-104) and KVStore.GetValueFromBadger (kv_store.go:179-204): */
/* the table lives in the HBM of one device.                                   */
/* ------------------------------------------------------------------------- */
typedef struct kaamer_index kaamer_index;

int kaamer_index_open_image(const kaamer_image *img, int device, kaamer_index **out);
/* makedb -> serving without an image in between: kaamer_image_build_proteins_device's
 * table stays on `device` and becomes the index. */
int kaamer_index_build_proteins(const uint8_t *seqs, const uint64_t *offsets, const uint32_t *ids,
                                uint32_t n_proteins, uint32_t shard, uint32_t n_shards,
                                double load_factor, int device, kaamer_index **out);
int kaamer_index_open(const char *path, int device, kaamer_index **out);
void kaamer_index_close(kaamer_index *ix);
int kaamer_index_get_stats(const kaamer_index *ix, kaamer_image_stats *out);

/* ------------------------------------------------------------------------- */
/* Search — replaces, per query, the block keyChan / KmerSearch /              */
/* StoreMatchPositions / sortMapByValue (search_protein.go:78-105,             */
/* search_fastq.go:94-118, search_nucleotide.go:91-115; search.go:414-452) and,*/
/* for nucleotide input, the preceding GetORFs call (dna.go:65-181;            */
/* search_fastq.go:74, search_nucleotide.go:136).                              */
/* ------------------------------------------------------------------------- */

/* One query handed to KmerSearch: a protein record, or one ORF of a read.     */
typedef struct {
    uint32_t src_seq;        /* index of the input sequence it came from       */
    int32_t size_in_kmer;    /* Query.SizeInKmer (search.go:290-293)           */
    int32_t start_position;  /* Location (dna.go:35-40); proteins: 1           */
    int32_t end_position;    /*                          proteins: len         */
    int32_t plus_strand;
    uint32_t aa_len;         /* residues of Query.Sequence                     */
    uint64_t aa_off;         /* into orf_aa (reads) / the input seqs (protein) */
    uint32_t sa_off, sa_len; /* Location.StartsAlternative, into starts_alt    */
} kaamer_query_meta;

/* Exact work counters of one batch, counted by the kernels themselves.        */
typedef struct {
    uint64_t n_in;       /* input bytes read (residues / nucleotides)          */
    uint64_t n_queries;  /* queries searched (proteins, or ORFs)               */
    uint64_t n_lookup;   /* k-mer lookups = KeyPos handed to KmerSearch        */
    uint64_t n_probe;    /* 64-byte buckets inspected (>= n_lookup)            */
    uint64_t n_found;    /* lookups whose key is present                       */
    uint64_t n_post;     /* postings expanded = sum |index[key]| over found    */
    uint64_t n_hits;     /* (query, protein) result pairs                      */
    uint64_t n_overflow; /* queries counted by the G tier: a table beyond a   */
                         /* wave's LDS arena, or more distinct hits than it    */
    uint64_t n_lists;    /* found lookups resolved through an arena list       */
                         /* (the others carry their single id in the slot)     */
    uint64_t n_list_ids; /* protein ids read from arena lists (<= n_post)      */
} kaamer_counters;

typedef struct {
    const uint8_t *seqs;      /* packed sequences, caller-owned                */
    const uint64_t *offsets;  /* n_seqs + 1                                    */
    uint32_t n_seqs;
    int32_t seq_type;         /* KAAMER_NUCLEOTIDE / PROTEIN / READS           */
    int32_t want_positions;   /* SearchOptions.ExtractPositions (search.go:64) */
} kaamer_batch_in;

typedef struct {
    uint32_t n_queries;
    const kaamer_query_meta *q;
    const uint64_t *hit_off;     /* CSR: hits of query i are [hit_off[i],      */
                                 /* hit_off[i+1]); n_queries + 1 entries       */
    const uint32_t *hit_cnt;     /* hits of each query                         */
    const uint32_t *hit_pid;     /* Hit.Key      (search.go:111-115)           */
    const uint32_t *hit_kmatch;  /* Hit.Kmatch; unsorted within a query        */
    const uint32_t *hit_first_pos; /* lowest query position that matched the   */
                                 /* hit: all SetBestStartCodon needs           */
                                 /* (dna.go:225-237)                           */
    const uint64_t *pos_off;     /* per hit, word offset into pos_bits; NULL   */
    const uint64_t *pos_bits;    /* PositionHits bitmaps (search.go:442-452),  */
                                 /* size_in_kmer bits per hit; NULL unless     */
                                 /* want_positions                             */
    const uint8_t *orf_aa;       /* reads/nucleotide: ORF amino-acid strings   */
    const int32_t *starts_alt;
    kaamer_counters counters;
} kaamer_batch_out;

/* May be called from any number of threads on one index (the worker pool of
 * search_protein.go:58-118): a call takes one of four slots -- workspace, staging,
 * stream -- and callers beyond the slots wait for one. */
int kaamer_search_batch(kaamer_index *ix, const kaamer_batch_in *in, kaamer_batch_out **out);
/* the same call with the caller's buffers as direct arguments (the form a cgo shim binds, see the conventions above) */
int kaamer_search_batch_flat(kaamer_index *ix, const uint8_t *seqs, const uint64_t *offsets, uint32_t n_seqs,
                             int32_t seq_type, int32_t want_positions, kaamer_batch_out **out);
void kaamer_batch_free(kaamer_batch_out *out);
/* The call in two halves (the worker pool of search_protein.go:58-118 when the full hit lists are wanted -- -pos reads
 * PositionHits -- without a blocked thread per batch): submit copies the caller's buffers, takes one of the four slots
 * (blocks while all are busy) and enqueues; wait brings the hit lists to the host, repeating the batch from the copy when a
 * workspace bound was too small, and gives the slot back.  A ticket is waited for, or discarded, exactly once. */
typedef struct kaamer_full_ticket kaamer_full_ticket;
int kaamer_submit_batch_flat(kaamer_index *ix, const uint8_t *seqs, const uint64_t *offsets, uint32_t n_seqs,
                             int32_t seq_type, int32_t want_positions, kaamer_full_ticket **ticket);
int kaamer_wait_batch(kaamer_full_ticket *ticket, kaamer_batch_out **out);
void kaamer_full_ticket_discard(kaamer_full_ticket *ticket);

/* ------------------------------------------------------------------------- */
/* Device-resident form of the same call: inputs already in HBM, results left  */
/* in HBM, everything enqueued on the caller's HIP stream, no host sync.       */
/* (Used by bench.py and by callers that pipeline batches; `stream` is a       */
/* hipStream_t passed as void*.)                                               */
/* ------------------------------------------------------------------------- */
typedef struct kaamer_workspace kaamer_workspace;

typedef struct {
    uint64_t max_seq_bytes;  /* largest packed input batch                     */
    uint32_t max_seqs;       /* most input sequences per batch                 */
    uint32_t max_queries;    /* most queries (ORFs) per batch; 0 = derive      */
    uint64_t max_hits;       /* most (query,protein) pairs per batch; 0=derive */
                             /* (from the input size).  It also bounds how far */
                             /* the per-query counting tables may grow on a     */
                             /* database where a k-mer meets several proteins:  */
                             /* every batch leaves the next one's table scale   */
                             /* (hits per k-mer x 1.9, <= 8) on the device, cut */
                             /* to what these arrays hold -- provision ~4 hits  */
                             /* per k-mer of the batch on such a database       */
    uint64_t g_tier_slots;   /* HBM counting-table slots for queries whose     */
                             /* distinct hits exceed the on-chip tiers; 0 = def*/
    int32_t seq_type;        /* KAAMER_PROTEIN, or KAAMER_NUCLEOTIDE / READS   */
                             /* (adds the 6-frame translation buffers)         */
    uint32_t first_pos;      /* hit_first_pos: 0 = as the reference fills      */
                             /* PositionHits (nucleotide/reads only,           */
                             /* search.go:416), 1 = always, 2 = never (zeros)  */
    uint32_t want_positions; /* full PositionHits bitmaps (ExtractPositions)   */
    uint32_t compact;        /* 1: finish with the hit lists packed in query   */
                             /* order (hit_off is a CSR offset array, one more */
                             /* scan + copy pass); 0: leave every list where   */
                             /* the counting kernel wrote it (hit_off[q],      */
                             /* hit_cnt[q] address a sparse array)             */
    uint64_t max_pos_words;  /* u64 words of bitmap storage; 0 = 8 x max_hits  */
    uint32_t concurrent_batches; /* batches the caller keeps in flight on this  */
                             /* index at once (other workspaces on other        */
                             /* streams); 0 / 1: a batch has the device to      */
                             /* itself.  With 2 or more the counting kernel of   */
                             /* a protein batch takes ONE workgroup per CU       */
                             /* instead of three, which leaves registers and     */
                             /* wave slots for the neighbours' probe kernels:    */
                             /* -8 % per batch with three in flight, +30 % for a */
                             /* batch that runs alone                            */
    uint32_t reserved;
} kaamer_workspace_opts;

typedef struct {
    uint32_t n_queries_cap;
    const uint32_t *d_n_queries;        /* device scalar                       */
    const kaamer_query_meta *d_q;
    const uint64_t *d_hit_off;          /* first hit of each query; with       */
                                        /* opts.compact also [n] = total (CSR) */
    const uint32_t *d_hit_cnt;          /* n_queries: hits of each query       */
    uint64_t hit_capacity;              /* entries of the three hit arrays     */
    const uint32_t *d_hit_pid;
    const uint32_t *d_hit_kmatch;
    const uint32_t *d_hit_first_pos;
    const uint8_t *d_orf_aa;
    const int32_t *d_starts_alt;
    const kaamer_counters *d_counters;  /* device copy, valid after the stream */
    const uint64_t *d_pos_off;          /* per hit: first word of its bitmap   */
    const uint64_t *d_pos_bits;         /* NULL unless want_positions          */
    const uint64_t *d_pos_base;         /* per query: first word; [n] = total  */
} kaamer_device_result;

int kaamer_workspace_create(kaamer_index *ix, const kaamer_workspace_opts *opts,
                            kaamer_workspace **out);
void kaamer_workspace_free(kaamer_workspace *ws);
int kaamer_search_device(kaamer_index *ix, kaamer_workspace *ws, const uint8_t *d_seqs,
                         const uint64_t *d_offsets, uint32_t n_seqs, uint64_t seq_bytes,
                         int32_t seq_type, void *stream, kaamer_device_result *out);
/* Two-stream form for callers that keep batches in flight: after this call, kaamer_search_device enqueues prep and
 * probe on ITS `stream` argument and everything behind the probe kernel (counting tiers, finalize; kaamer_topn_device
 * follows) on `count_stream`, ordered by events inside the library.  With ONE probe stream shared by all batches and the
 * workspaces' counting stages on one or two count streams, the device always holds one probe kernel (bound by memory
 * requests) next to the counting kernels of earlier batches (bound by latency) instead of whatever mix independent
 * streams drift into.  kaamer_workspace_finish and the exchange calls wait for the counting stage wherever it ran.
 * stream = NULL: back to one stream. */
int kaamer_workspace_set_count_stream(kaamer_workspace *ws, void *count_stream);
/* Sharded index (the table split by hash prefix over several devices): every shard
 * searches the whole batch for the keys it owns, the partial hit lists travel to the
 * query's owner (all-to-all, done by the caller), and the owner merges them here:
 * entries [ent_off[q], ent_off[q+1]) are the partial (protein id, Kmatch, first position)
 * of query q from all shards; Kmatch adds up, first positions take the minimum.
 * Results (CSR) replace the workspace's last result; finish as after a search. */
int kaamer_merge_device(kaamer_workspace *ws, const uint64_t *d_ent_off, const uint32_t *d_pid,
                        const uint32_t *d_kmatch, const uint32_t *d_first_pos, uint32_t n_queries,
                        uint64_t n_entries, void *stream, kaamer_device_result *out);
/* Post-steps of the workspace's last search (or merge) on the device, one wave per
 * query: sortMapByValue (search.go:132-152; ties by ascending protein id),
 * for nucleotide/reads input SetBestStartCodon (dna.go:198-272) with its gate
 * hits[0].Kmatch >= MinKMatch (search_fastq.go:119-123), then FilterResults
 * (search.go:189-220).  Only the hits a caller returns leave the device:
 * max_results entries per query instead of the full hit lists. */
typedef struct {
    double min_k_ratio;        /* SearchOptions.MinKRatio  (default 0.05)      */
    int64_t min_k_match;       /* SearchOptions.MinKMatch  (default 10)        */
    uint32_t max_results;      /* SearchOptions.MaxResults (default 10)        */
    uint32_t best_start_codon; /* 1: SetBestStartCodon first (nucleotide/reads)*/
    const int32_t *d_size_in_kmer; /* device, per query; NULL = the search's   */
                               /* own queries; after a merge: this, or orf_source */
    /* merged results of a sharded index: result i belongs to query             */
    /* q_first + i * q_stride of ANOTHER workspace's last search (the owner's    */
    /* own translation of the batch: rank r owns queries r, r+W, ...), whose     */
    /* ORFs, StartsAlternative and SizeInKmer the post-steps then use            */
    /* (SetBestStartCodon included).  NULL: the workspace's own queries.         */
    const struct kaamer_workspace *orf_source;
    uint32_t q_first, q_stride;
} kaamer_topn_opts;

typedef struct {
    uint32_t max_results;
    const uint32_t *d_top_cnt;        /* per query: hits kept; 0 = not reported */
    const uint32_t *d_top_pid;        /* [q * max_results + r], r < d_top_cnt[q], */
    const uint32_t *d_top_kmatch;     /* in sortMapByValue order                */
    const uint32_t *d_top_first_pos;
    const int32_t *d_trim;            /* residues SetBestStartCodon removed     */
    const int32_t *d_start_position;  /* Location.StartPosition afterwards      */
    const int32_t *d_size_in_kmer;    /* Query.SizeInKmer afterwards            */
} kaamer_topn_result;

int kaamer_topn_device(kaamer_workspace *ws, const kaamer_topn_opts *opts, void *stream,
                       kaamer_topn_result *out);
/* Host-buffer call that returns what the reference's drivers report
 * (search_protein.go:105-112, search_fastq.go:118-126): the queries FilterResults left
 * with at least one hit, their hits in sortMapByValue order, after SetBestStartCodon
 * for nucleotide / reads input (best_start_codon is set from in->seq_type).  The
 * packing is done on the device: only the reported queries, their hits (CSR) and their
 * ORF residues cross PCIe.  q[i] carries Location.StartPosition, SizeInKmer and the
 * Sequence window (aa_off, aa_len) as they are after SetBestStartCodon; aa_off indexes
 * orf_aa below (nucleotide / reads) or the caller's own input (protein). */
typedef struct {
    uint32_t n_queries;             /* queries (proteins / ORFs) searched      */
    uint32_t n_reported;            /* those with a hit left after the filter  */
    uint32_t max_results;
    const uint32_t *rep_query;      /* [n_reported] index among the n_queries, */
                                    /* ascending (ORFs: the reference's order) */
    const kaamer_query_meta *q;     /* [n_reported]                            */
    const int32_t *trim;            /* residues removed from the ORF head      */
    const uint64_t *top_off;        /* [n_reported + 1] CSR into the arrays    */
    const uint32_t *top_pid;
    const uint32_t *top_kmatch;
    const uint32_t *top_first_pos;  /* relative to the untrimmed ORF; zeros for */
                                    /* protein input (search.go:416)            */
    const uint8_t *orf_aa;          /* reported ORFs' residues, concatenated   */
    kaamer_counters counters;
} kaamer_batch_top;

int kaamer_search_batch_top(kaamer_index *ix, const kaamer_batch_in *in, const kaamer_topn_opts *top,
                            kaamer_batch_top **out);
void kaamer_batch_top_free(kaamer_batch_top *out);

/* The same call in two halves, for callers that keep several batches in flight: the worker pool of
 * search_protein.go:58-118 (N goroutines against shared read-only stores) becomes N callers against one index, each
 * batch on a slot of its own (workspace, stream, pinned staging, result block; KAAMER_HOST_SLOTS of them, default 4).
 * kaamer_search_batch_top is submit + wait and may be called from any number of threads concurrently.
 *   submit  takes a free slot (blocks while all are busy), COPIES the caller's buffers (they are borrowed for the
 *           call only), enqueues host-to-device copy, search, post-steps and ONE device-to-host copy of the packed
 *           result, and returns;
 *   wait    blocks until that batch is done, returns its result (free it with kaamer_batch_top_free) and releases the
 *           slot and the ticket.  A ticket must be waited for exactly once, by any thread. */
typedef struct kaamer_ticket kaamer_ticket;
int kaamer_submit_batch_top(kaamer_index *ix, const kaamer_batch_in *in, const kaamer_topn_opts *top, kaamer_ticket **ticket);
int kaamer_wait_batch_top(kaamer_ticket *ticket, kaamer_batch_top **out);
/* A ticket nobody will wait for (an error between submit and wait on the caller's side): lets the batch run out, drops
 * its result and gives the slot back.  Without it the slot stays busy for good; kaamer_index_close waits for every
 * slot to be given back. */
void kaamer_ticket_discard(kaamer_ticket *ticket);
/* cgo-safe forms: the caller's buffers and SearchOptions.MinKRatio / MinKMatch / MaxResults (search.go:56-71) as direct
 * arguments; best_start_codon follows from seq_type as in the struct forms. */
int kaamer_search_batch_top_flat(kaamer_index *ix, const uint8_t *seqs, const uint64_t *offsets, uint32_t n_seqs,
                                 int32_t seq_type, double min_k_ratio, int64_t min_k_match, uint32_t max_results,
                                 kaamer_batch_top **out);
int kaamer_submit_batch_top_flat(kaamer_index *ix, const uint8_t *seqs, const uint64_t *offsets, uint32_t n_seqs,
                                 int32_t seq_type, double min_k_ratio, int64_t min_k_match, uint32_t max_results,
                                 kaamer_ticket **ticket);

/* Streaming (BASELINE configs[4]: reads streamed host -> GPU with double-buffered copies): a FIFO of batches with
 * fixed options.  push copies chunk i + 1 and starts it while chunk i is still being searched; pop returns the
 * oldest chunk's reported hits.  push returns KAAMER_E_BUSY instead of blocking when every slot holds one of this
 * stream's own chunks: pop first.  One thread per stream; several streams may share an index. */
typedef struct kaamer_stream kaamer_stream;
int kaamer_stream_open(kaamer_index *ix, int32_t seq_type, const kaamer_topn_opts *top, kaamer_stream **out);
int kaamer_stream_open_flat(kaamer_index *ix, int32_t seq_type, double min_k_ratio, int64_t min_k_match,
                            uint32_t max_results, kaamer_stream **out);
int kaamer_stream_push(kaamer_stream *st, const uint8_t *seqs, const uint64_t *offsets, uint32_t n_seqs);
int kaamer_stream_pop(kaamer_stream *st, kaamer_batch_top **out);
uint32_t kaamer_stream_pending(const kaamer_stream *st);
void kaamer_stream_close(kaamer_stream *st);

/* ------------------------------------------------------------------------- */
/* Sharded index, the exchange step as device code (no reference counterpart:  */
/* kaamer is one process; the result must equal the per-query block            */
/* search_fastq.go:94-118 / search_protein.go:78-105 run against the whole DB). */
/* One process per GPU; rank r holds shard r (kaamer_image_build_* (r, W)),     */
/* every rank searches the WHOLE batch (nucleotide input: every rank            */
/* translates, so all ranks number the ORFs alike), query q is owned by rank    */
/* q mod W.  Per batch, on the caller's stream and without host synchronisation: */
/*   kaamer_search_device(shard)           partial hit lists of all queries      */
/*   kaamer_exchange_pack                  -> W fixed-size blocks, one per owner  */
/*   all-to-all with equal splits          grouped ncclSend/ncclRecv (RCCL over   */
/*                                         xGMI), the caller's communicator, or   */
/*                                         kaamer_rccl_alltoall below             */
/*   kaamer_exchange_merge(owner ws)       received blocks -> merged hit lists of  */
/*                                         the owned queries (integer sums: bit-   */
/*                                         identical to the one-device result)     */
/*   kaamer_topn_device(owner ws, orf_source = the search workspace, q_first =     */
/*                      rank, q_stride = W)                                        */
/* Block capacities are bounds like every workspace bound: exceeding them is      */
/* reported as KAAMER_E_CAPACITY by kaamer_workspace_finish, never a partial       */
/* result.  First positions travel only when the search workspace computes them    */
/* (opts.first_pos = 1, or nucleotide / reads input); create the owner's merge      */
/* workspace with the same explicit setting (1 or 2) -- an owner that wants them    */
/* and receives blocks without them reports the same error.                         */
/* ------------------------------------------------------------------------- */
typedef struct {
    uint32_t world, rank;
    uint32_t q_cap;        /* owned queries a block can describe                 */
    uint32_t arrays;       /* entry arrays in a block: 3 = id, Kmatch, first     */
                           /* position; 2 = no first positions (0 reads as 3)    */
    uint64_t e_cap;        /* partial (id, Kmatch, first position) entries/block */
    uint64_t block_words;  /* u32 words per block; send / receive buffers hold   */
                           /* world blocks                                       */
} kaamer_exchange_layout;

/* max_queries: the search workspace's query capacity (kaamer_workspace_query_capacity).  This is the CAPACITY layout:
 * send / receive buffers hold world * block_words words of it. */
int kaamer_exchange_layout_init(uint32_t world, uint32_t rank, uint32_t max_queries,
                                uint64_t max_entries_per_peer, kaamer_exchange_layout *out);
/* The layout of ONE batch inside the same buffers: blocks for n_queries queries of the batch and entries_per_block
 * partial entries (both cut to the capacity layout), without room for first positions when the workspaces do not
 * compute them (with_first_pos = 0: first_pos = 2 on both workspaces).  What the all-to-all moves is world blocks of THIS layout's
 * block_words, contiguous from the start of the buffers -- payload, not capacity.  Every rank must use the same layout
 * for a batch: derive the two figures from kaamer_exchange_stats of an earlier batch (identical on all ranks) plus a
 * margin; a batch that does not fit fails on every rank with KAAMER_E_CAPACITY and is repeated with the capacity
 * layout. */
int kaamer_exchange_layout_fit(const kaamer_exchange_layout *capacity, uint32_t n_queries, uint64_t entries_per_block,
                               int32_t with_first_pos, kaamer_exchange_layout *out);
/* What the W block headers of an earlier kaamer_exchange_merge on `merge_ws` said, the same on every rank:
 * out = { merge sequence number, queries of the batch, entries the largest block between ANY pair of ranks needed,
 * non-zero if a block overflowed }.  back = 0: the last merge enqueued, 1: the one before.  Waits for that merge's
 * header scan only (an event), not for the stream. */
int kaamer_exchange_stats(kaamer_workspace *merge_ws, uint32_t back, uint64_t out[4]);
uint32_t kaamer_workspace_query_capacity(const kaamer_workspace *ws);
/* the last search of `search_ws` -> d_send[world * block_words] */
int kaamer_exchange_pack(kaamer_workspace *search_ws, const kaamer_exchange_layout *layout,
                         uint32_t *d_send, void *stream);
/* d_recv[world * block_words] (block s = what rank s sent to this rank) -> merged results in
 * `merge_ws` (max_queries >= layout->q_cap, max_hits >= world * layout->e_cap, first_pos = 1 or 2: the same explicit
 * setting as the search workspace).  Blocks whose headers do not describe ONE batch (ranks on different batches, a
 * stale buffer), blocks that overflowed, and blocks of a sender whose own search failed all make the merge an error
 * (KAAMER_E_CAPACITY at kaamer_workspace_finish): never a partial result. */
int kaamer_exchange_merge(kaamer_workspace *merge_ws, const kaamer_exchange_layout *layout,
                          const uint32_t *d_recv, void *stream, kaamer_device_result *out);
/* Grouped ncclSend / ncclRecv of world equal blocks on `stream` with the caller's ncclComm_t.
 * The library does not link RCCL: the symbols are taken from the process (or librccl.so.1). */
int kaamer_rccl_alltoall(void *nccl_comm, const void *d_send, void *d_recv, uint64_t bytes_per_peer,
                         uint32_t world, void *stream);

/* ------------------------------------------------------------------------- */
/* One process, one handle, several devices: the reference server   */
/* is ONE process that opens its stores once and fans out goroutines            */
/* (api/server.go:47-65, search_fastq.go:60-66).  image i / paths[i] is shard i  */
/* of n_shards (kaamer_image_build_* (i, n_shards)) and is made resident on      */
/* devices[i] (the same device may appear more than once).  A call drives all    */
/* shards from the calling thread: every shard searches the whole batch for the  */
/* keys it owns, every owner (query q: shard q mod n_shards) pulls its blocks     */
/* with peer copies over xGMI, merges, runs the post-steps; the reported queries  */
/* come back in batch order, exactly what kaamer_search_batch_top returns on an   */
/* unsharded index.  No communicator, no second process.  Any number of threads   */
/* may call on one handle: a call takes one of three sets of per-shard            */
/* workspaces, buffers and streams; callers beyond the sets wait for one.         */
/* ------------------------------------------------------------------------- */
typedef struct kaamer_sharded_index kaamer_sharded_index;
int kaamer_index_open_sharded(const char *const *paths, const int *devices, uint32_t n_shards, kaamer_sharded_index **out);
int kaamer_index_open_sharded_images(const kaamer_image *const *images, const int *devices, uint32_t n_shards,
                                     kaamer_sharded_index **out);
uint32_t kaamer_sharded_index_shards(const kaamer_sharded_index *sx);
void kaamer_sharded_index_close(kaamer_sharded_index *sx);
int kaamer_sharded_search_batch_top(kaamer_sharded_index *sx, const kaamer_batch_in *in, const kaamer_topn_opts *top,
                                    kaamer_batch_top **out);
/* The call in two halves (the worker pool of search_protein.go:58-118 against one sharded handle without a blocked
 * thread per batch): submit takes a free set, copies the caller's buffers, enqueues the batch on every device and
 * returns; wait collects the result (repeating the batch from the staging copy when a bound was too small) and gives
 * the set back.  A ticket is waited for, or discarded, exactly once. */
typedef struct kaamer_sharded_ticket kaamer_sharded_ticket;
int kaamer_sharded_submit_batch_top(kaamer_sharded_index *sx, const kaamer_batch_in *in, const kaamer_topn_opts *top,
                                    kaamer_sharded_ticket **ticket);
int kaamer_sharded_wait_batch_top(kaamer_sharded_ticket *ticket, kaamer_batch_top **out);
void kaamer_sharded_ticket_discard(kaamer_sharded_ticket *ticket);
/* cgo-safe forms (see the conventions at the top) */
int kaamer_sharded_search_batch_top_flat(kaamer_sharded_index *sx, const uint8_t *seqs, const uint64_t *offsets,
                                         uint32_t n_seqs, int32_t seq_type, double min_k_ratio, int64_t min_k_match,
                                         uint32_t max_results, kaamer_batch_top **out);
int kaamer_sharded_submit_batch_top_flat(kaamer_sharded_index *sx, const uint8_t *seqs, const uint64_t *offsets,
                                         uint32_t n_seqs, int32_t seq_type, double min_k_ratio, int64_t min_k_match,
                                         uint32_t max_results, kaamer_sharded_ticket **ticket);
/* The exchange of the last finished call on the handle's first set: out = { bytes of one (shard -> owner) block as it
 * travelled, entries the largest block needed, queries of the batch, 1 if the blocks were sized from the previous
 * call's need (payload) rather than the buffers' capacity }. */
int kaamer_sharded_exchange_info(kaamer_sharded_index *sx, uint64_t out[4]);

/* ------------------------------------------------------------------------- */
/* One process, the same database on several devices (SURVEY 8e "replicas only": */
/* a database that fits one device is not sharded, the query stream is split;     */
/* BASELINE configs[4]).  The reference is ONE server process (api/server.go:47-65) */
/* whose FastqSearch takes a file (search_fastq.go:60-136).                        */
/* ------------------------------------------------------------------------- */
typedef struct kaamer_replicas kaamer_replicas;
/* the image is read once and made resident on every devices[i] */
int kaamer_index_open_replicas(const char *path, const int *devices, uint32_t n, kaamer_replicas **out);
int kaamer_index_open_replicas_image(const kaamer_image *img, const int *devices, uint32_t n, kaamer_replicas **out);
uint32_t kaamer_replicas_count(const kaamer_replicas *r);
kaamer_index *kaamer_replicas_index(const kaamer_replicas *r, uint32_t i);   /* replica i: every single-index call works on it */
void kaamer_replicas_close(kaamer_replicas *r);
/* A FIFO of chunks over the set: chunk k goes to replica k mod n (its own slots, stream and staging), pop returns the
 * chunks in push order.  push returns KAAMER_E_BUSY when every slot of the replica whose turn it is holds one of this
 * stream's chunks: pop first.  One thread per stream. */
typedef struct kaamer_replica_stream kaamer_replica_stream;
int kaamer_replica_stream_open_flat(kaamer_replicas *r, int32_t seq_type, double min_k_ratio, int64_t min_k_match,
                                    uint32_t max_results, kaamer_replica_stream **out);
int kaamer_replica_stream_push(kaamer_replica_stream *rs, const uint8_t *seqs, const uint64_t *offsets, uint32_t n_seqs);
int kaamer_replica_stream_pop(kaamer_replica_stream *rs, kaamer_batch_top **out);
uint32_t kaamer_replica_stream_pending(const kaamer_replica_stream *rs);
void kaamer_replica_stream_close(kaamer_replica_stream *rs);
/* FastqSearch / ProteinSearch over a FILE of any size (search_fastq.go:60-136, search_protein.go:40-118: reader ->
 * queryChan -> workers -> result handler): kaamer_reader_* (incremental, gzip; strict_scanner as there) feeds chunks of
 * at most chunk_seqs records / about chunk_bytes of sequence round-robin to the replicas, up to in_flight chunks per
 * replica in flight (0: three); `cb` is called once per chunk, in input order, with the chunk's records and its
 * reported hits (both valid during the call only); first_seq = index of the chunk's first record in the file;
 * rep_query / q[].src_seq of `top` count from the chunk's first record.  A non-zero return of cb stops the run.
 * `total` (may be NULL) receives the summed work counters.  Memory: in_flight x replicas chunks, whatever the file size. */
typedef int (*kaamer_chunk_cb)(void *user, uint64_t first_seq, const kaamer_reads *chunk, const kaamer_batch_top *top);
int kaamer_search_file(kaamer_replicas *r, const char *path, int format, int strict_scanner, int32_t seq_type,
                       double min_k_ratio, int64_t min_k_match, uint32_t max_results, uint32_t chunk_seqs,
                       uint64_t chunk_bytes, uint32_t in_flight, kaamer_chunk_cb cb, void *user, kaamer_counters *total);

/* Waits for `stream`, copies the counters to the host and reports a deferred
 * KAAMER_E_CAPACITY if a device-side bound was exceeded during the batch. */
int kaamer_workspace_finish(kaamer_workspace *ws, void *stream, kaamer_counters *out);
/* Kernel timers.  kaamer_workspace_set_timing(ws, k): k = 0 (default) records no
 * events; k >= 1 brackets the kernels of every k-th kaamer_search_device call with
 * HIP events on the caller's stream (each record idles the stream for a few
 * microseconds, hence the sampling).  kaamer_workspace_kernel_ms_sum returns the
 * times (ms) summed over the n_calls sampled calls since the previous
 * kaamer_workspace_reset_timers (at most 1024 are kept):
 *   probe_ms  the probe kernel (the dominant kernel: one 64-B bucket per lookup)
 *   count_ms  the counting tiers (postings expansion + per-protein counting)
 *   total_ms  the whole batch, prep to CSR
 * The stream must have been synchronised (kaamer_workspace_finish does). */
int kaamer_workspace_kernel_ms_sum(kaamer_workspace *ws, double *probe_ms, double *count_ms,
                                   double *total_ms, uint32_t *n_calls);
void kaamer_workspace_reset_timers(kaamer_workspace *ws);
void kaamer_workspace_set_timing(kaamer_workspace *ws, uint32_t every);

/* ------------------------------------------------------------------------- */
/* Host post-steps kept bit-compatible with the reference (they stay on the    */
/* host in the Go integration; exported so non-Go callers get the same         */
/* results).                                                                   */
/* ------------------------------------------------------------------------- */
/* QueryResult.FilterResults, search.go:189-220: hits sorted by Kmatch desc;
 * returns how many (a prefix) survive MinKRatio / MinKMatch / MaxResults.     */
int64_t kaamer_filter_results(const uint32_t *kmatch_sorted, int64_t n_hits, int32_t size_in_kmer,
                              double min_k_ratio, int64_t min_k_match, int64_t max_results);
/* sortMapByValue, search.go:132-152: order hit indices by Kmatch descending;
 * ties (nondeterministic in the reference) are broken by ascending protein id. */
void kaamer_sort_hits(const uint32_t *pid, const uint32_t *kmatch, int64_t n_hits, uint32_t *order);

/* SetBestStartCodon, dna.go:198-272: hits in sortMapByValue order with their first
 * matching positions; trims the ORF to the start codon preceding the first best-hit
 * position.  Returns the residues trimmed (0 = unchanged); start_position and
 * size_in_kmer are updated like dna.go:252-267. */
int32_t kaamer_set_best_start_codon(const uint32_t *kmatch_sorted, const uint32_t *first_pos_sorted,
                                    int64_t n_hits, const int32_t *starts_alt, int32_t n_starts,
                                    int32_t plus_strand, const uint8_t *orf_aa, uint32_t aa_len,
                                    int32_t *start_position, int32_t *size_in_kmer);

/* ------------------------------------------------------------------------- */
/* Alignment of the reported hits (`-aln`) — replaces, for all (query, hit) pairs */
/* of a batch at once, the loop of QueryResultHandler (search.go:483-494) over     */
/* align.Align (pkg/align/align.go:46-161): local alignment with affine gaps on    */
/* the device (one lane per pair), then the reference's own arithmetic on the host: */
/* Identity / Similarity (float32), Length, Mismatches, GapOpenings, Raw, BitScore  */
/* = (lambda * Raw - ln K) / ln 2, EValue = len(query) * NumberOfAA / 2^BitScore,    */
/* the 1-based inclusive coordinates and AlnString's three rows.                    */
/* The aligner itself is github.com/biogo/biogo v1.0.1 (align.SWAffine on          */
/* matrix.BLOSUM62 with GapOpen -11, align.go:62-67 -- fixed, whatever the options   */
/* are), which is not part of the reference tree: the recurrence is restated from    */
/* its published algorithm and its tie-breaking is not pinned by anything in the     */
/* reference.                                                  */
/* ------------------------------------------------------------------------- */
typedef struct {
    float identity, similarity;   /* AlignmentResult.Identity / .Similarity (percent, float32)     */
    int32_t length;               /* columns of the alignment = len of each row of AlnString        */
    int32_t mismatches, gap_openings, raw;
    double bitscore, evalue;
    int32_t query_start, query_end, subject_start, subject_end;   /* 1-based, inclusive             */
    uint64_t aln_off;             /* into kaamer_alignments_text: query row, match row, subject row, */
                                  /* `length` bytes each (AlnString joins them with '\n')            */
    int32_t status;               /* 0 aligned; 1 "No matrix found" (GetMatrixScores: the reference  */
                                  /* keeps an empty AlignmentResult; also any matrix but BLOSUM62,    */
                                  /* whose data the similarity marks need); 2 a letter outside the    */
                                  /* protein alphabet "-ABCDEFGHIJKLMNPQRSTVWXYZ*" (SWAffine fails);   */
                                  /* 3 a sequence too long                                            */
    int32_t reserved;
} kaamer_alignment;
typedef struct kaamer_alignments kaamer_alignments;
/* pair i aligns sequence pair_query[i] (the Query.Sequence) with sequence pair_subject[i] (the hit's Protein.Sequence) of
 * the packed buffer (seqs, offsets[n_seqs + 1]); number_of_aa = KStats.NumberOfAA (kaamer_proteins_stats); sub_matrix /
 * gap_open / gap_extend = SearchOptions.SubMatrix / GapOpen / GapExtend (defaults "blosum62", 11, 1: api/server.go:149-151).
 * All caller buffers are direct arguments (cgo-safe).  Results in pair order. */
int kaamer_align_pairs(int device, const uint8_t *seqs, const uint64_t *offsets, uint32_t n_seqs,
                       const uint32_t *pair_query, const uint32_t *pair_subject, uint32_t n_pairs,
                       uint64_t number_of_aa, const char *sub_matrix, int32_t gap_open, int32_t gap_extend,
                       kaamer_alignments **out);
uint32_t kaamer_alignments_count(const kaamer_alignments *a);
const kaamer_alignment *kaamer_alignments_items(const kaamer_alignments *a);
const char *kaamer_alignments_text(const kaamer_alignments *a);
void kaamer_alignments_free(kaamer_alignments *a);
/* GetMatrixScores (matrixScores.go:107-115): KAAMER_E_ARG ("No matrix found") when the table has no such row */
int kaamer_align_matrix_scores(const char *sub_matrix, int32_t gap_open, int32_t gap_extend, double *lambda, double *k);
/* GetAlnScoreAA on BLOSUM62 (matrixScores.go:125-127): letters of AAPosInMatrix, anything else reads as '-' */
int32_t kaamer_align_matrix_entry(int32_t a, int32_t b);

/* ------------------------------------------------------------------------- */
/* Readers — GetQueriesFasta / GetQueriesFastq (search.go:222-412) on a text   */
/* buffer (already decompressed): packed sequences + SizeInKmer + names, with  */
/* the reference's quirks (every FASTA record but the last is upper-cased; '*' */
/* rule; FASTQ sequence lines must match ^[ATGCNatgcn]+$).                     */
/* ------------------------------------------------------------------------- */
/* `text` is the bytes of the file.  Bytes that start with the gzip signature (what the reference's
 * http.DetectContentType calls application/x-gzip, search.go:255-263, 361-366) are inflated first: every member of the
 * stream (Go's gzip.Reader is multistream); a stream that breaks off or is damaged reads as what inflated before that
 * (the reference's scanner stops at the read error and processes what it got); a first header that is no gzip header
 * gives an EMPTY read set (gzip.NewReader fails, GetQueriesFasta prints the error and returns without a query,
 * search.go:259-263).  kaamer_makedb_embl / _gbk inflate the same way (inputEMBL.go:76-84, inputGBK.go:75-83) but report
 * a bad first header as KAAMER_E_FORMAT (the reference log.Fatal()s there); the FASTA and TSV makedb readers have no gzip
 * branch in the reference. */
int kaamer_parse_fasta(const char *text, uint64_t len, kaamer_reads **out);
int kaamer_parse_fastq(const char *text, uint64_t len, kaamer_reads **out);
uint32_t kaamer_reads_count(const kaamer_reads *r);
const uint8_t *kaamer_reads_seqs(const kaamer_reads *r);
const uint64_t *kaamer_reads_offsets(const kaamer_reads *r);       /* count + 1 */
const int32_t *kaamer_reads_size_in_kmer(const kaamer_reads *r);
const char *kaamer_reads_names(const kaamer_reads *r);
const uint64_t *kaamer_reads_name_offsets(const kaamer_reads *r);  /* count + 1 */
/* Location.PlusStrand as the reference's reader leaves it: 1 for the first record, 0 (Go's zero value) for every
 * following one (search.go:297,399 rebuild the Query without a Location); protein results report it unchanged.
 * Not reproduced: bufio.Scanner's 1 MiB line limit (search.go:273-274: a longer line silently ends the
 * reference's scan) -- lines of any length are read. */
const int32_t *kaamer_reads_plus_strand(const kaamer_reads *r);
void kaamer_reads_free(kaamer_reads *r);

/* The same readers over a FILE, in chunks (search.go:240-283: open, sniff 32 bytes, gzip.NewReader when they carry the
 * gzip signature, bufio.Scanner line by line): a read set of any size goes through bounded memory -- BASELINE
 * configs[4].  format: 0 = GetQueriesFasta, 1 = GetQueriesFastq.  kaamer_reader_next hands out the next records -- up
 * to max_seqs of them, stopping once about max_bytes of sequence are in the chunk -- as a kaamer_reads (accessors above;
 * free it with kaamer_reads_free), ready for kaamer_stream_push / kaamer_replica_stream_push; a chunk with zero records
 * and kaamer_reader_done() = 1 is the end.  The rules that span records hold across chunks: FASTA upper-cases every
 * record but the LAST of the file, PlusStrand is 1 on the file's first record only.  gzip: every member, a stream that
 * breaks off or is damaged reads as what inflated before; a first header that is no gzip header: no records.
 * strict_scanner = 1 additionally reproduces what the reference's scanner setup does to unusual input: a line of
 * 1 048 576 bytes or more ends the input there (bufio.Scanner with a 1 MiB buffer returns false, search.go:273-274; the
 * record being read is emitted as the last one), and a file whose first 32 bytes are not what http.DetectContentType
 * calls "text/plain; charset=utf-8" (a UTF-16 byte-order mark, a control byte) yields no records (search.go:266-270).
 * strict_scanner = 0 reads lines of any length. */
typedef struct kaamer_reader kaamer_reader;
int kaamer_reader_open(const char *path, int format, int strict_scanner, kaamer_reader **out);
int kaamer_reader_open_fd(int fd, int format, int strict_scanner, kaamer_reader **out); /* fd stays the caller's */
int kaamer_reader_next(kaamer_reader *rd, uint32_t max_seqs, uint64_t max_bytes, kaamer_reads **out);
int kaamer_reader_done(const kaamer_reader *rd);
uint64_t kaamer_reader_records(const kaamer_reader *rd);   /* records handed out so far */
void kaamer_reader_close(kaamer_reader *rd);

#ifdef __cplusplus
}
#endif
#endif /* KAAMER_HIP_H */
