/*
 * kaamer_oracle.c — CPU restatement of the kaamer k-mer search hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped product (kaamer_amd/,
 * include/, the C-ABI library) may link, import or call this file.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and
 * only as the checker / the reported CPU baseline.
 *
 * PARITY UNPINNED: the reference (zorino/kaamer, Go) ships no tests, no golden
 * vectors and cannot be built in this pipeline (no Go toolchain, un-vendored
 * badger/counters/xxhash/protobuf).  This restatement is pinned only by
 *   - the worked example in docs/client.md:120-156 (270-aa query, SizeInKmer
 *     264, self hit Kmatch 264, positions 1-264),
 *   - the codon table parsed as text from pkg/search/gcode.go:36-101
 *     (tests/golden/gcode_bacteria.json),
 *   - hand-derived known-answer vectors, and
 *   - an independent pure-Python restatement in tests/pyref.py.
 *
 * Every function cites the reference file:line (paths under /root/reference)
 * that it follows.  The style is deliberately literal: strings are built the
 * way the Go code builds them, maps are replaced by tables filled by the same
 * loops, Badger is replaced by a sorted (key,id) array with binary search
 * (exact-match point-read semantics, kv_store.go:179-204).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define KMER_SIZE 7      /* pkg/search/search.go:45, pkg/makedb/makedb.go:30 */
#define MIN_LEN_CDS 21   /* pkg/search/dna.go:26 */

/* ------------------------------------------------------------------------ */
/* K-mer codec: pkg/kvstore/k_store.go:39-117                                */
/* ------------------------------------------------------------------------ */

static uint32_t g_single[256];       /* aaTable[{a,'.'}]  (k_store.go:50-52) */
static uint32_t g_pair[256][256];    /* aaTable[{a,b}]    (k_store.go:53-59) */
static uint8_t g_single_set[256];
static int g_tables_ready = 0;

/* NewAATable, k_store.go:39-64: same alphabet order, same counter. */
static void ko_build_tables(void)
{
    static const char aa[] = "ACDEFGHIKLMNPQRSTUVWY"; /* k_store.go:41 */
    if (g_tables_ready) return;
    memset(g_single, 0, sizeof g_single);
    memset(g_pair, 0, sizeof g_pair);
    memset(g_single_set, 0, sizeof g_single_set);
    uint32_t i = 22;                                   /* k_store.go:46 */
    for (int j = 0; j < 21; j++) {
        unsigned a = (unsigned char)aa[j];
        g_single[a] = (uint32_t)j;                     /* k_store.go:49-51 */
        g_single_set[a] = 1;
        /* aaTable[{a,'.'}] = j is an ordinary map entry: the PAIR lookup of
         * EncodeKmer (k_store.go:102-103) also finds it when the second letter
         * of a pair is '.', e.g. "C.AAAAA" -> 0x008582C0. */
        g_pair[a][(unsigned char)'.'] = (uint32_t)j;
        for (int b = 0; b < 21; b++) {
            g_pair[a][(unsigned char)aa[b]] = i;       /* k_store.go:54-56 */
            i++;                                       /* k_store.go:58 */
        }
    }
    g_tables_ready = 1;
}

/* EncodeKmer, k_store.go:91-117.  A Go map miss yields the zero value, so any
 * byte outside the alphabet makes its pair (or the last single) contribute 0. */
uint32_t ko_encode_kmer(const uint8_t *kmer, int len)
{
    ko_build_tables();
    uint32_t kmer_int = 0;
    int i = 0;
    uint8_t shift_index = 1;
    while (i + 2 < len) {                              /* k_store.go:100 */
        uint32_t v = g_pair[kmer[i]][kmer[i + 1]];
        kmer_int |= v << (32 - (uint8_t)(shift_index * 9)); /* k_store.go:103 */
        shift_index++;
        i += 2;
    }
    kmer_int |= g_single[kmer[len - 1]];               /* k_store.go:109-110 */
    return kmer_int;
}

/* CreateBytesKey, k_store.go:66-76: big-endian 4 bytes. */
void ko_create_bytes_key(const uint8_t *kmer, uint8_t out[4])
{
    uint32_t k = ko_encode_kmer(kmer, KMER_SIZE);
    out[0] = (uint8_t)(k >> 24); out[1] = (uint8_t)(k >> 16);
    out[2] = (uint8_t)(k >> 8);  out[3] = (uint8_t)k;
}

/* DecodeKmer, k_store.go:120-145 (inverse, for valid keys only). */
void ko_decode_kmer(uint32_t key, char out[8])
{
    static const char aa[] = "ACDEFGHIKLMNPQRSTUVWY";
    uint32_t f[3] = { (key >> 23) & 0x1FF, (key >> 14) & 0x1FF, (key >> 5) & 0x1FF };
    for (int s = 0; s < 3; s++) {
        if (f[s] >= 22 && f[s] < 22 + 441) {
            out[2 * s] = aa[(f[s] - 22) / 21];
            out[2 * s + 1] = aa[(f[s] - 22) % 21];
        } else {
            out[2 * s] = '?'; out[2 * s + 1] = '?';
        }
    }
    uint32_t d = key & 0x1F;
    out[6] = d < 21 ? aa[d] : '?';
    out[7] = 0;
}

/* ------------------------------------------------------------------------ */
/* Translation: pkg/search/gcode.go:36-101 (gcodeBacteria == NCBI table 11)   */
/* ------------------------------------------------------------------------ */

typedef struct { char aa; uint8_t start, stop; } ko_amino; /* dna.go:29-33; aa==0 <=> AA:"" */

static int ko_nt_index(uint8_t c)
{
    switch (c) { case 't': return 0; case 'c': return 1; case 'a': return 2; case 'g': return 3; }
    return -1;
}

/* gcodeBacteria[codon]; a map miss (any byte other than lower-case t/c/a/g)
 * yields the zero AminoAcid: AA "", Start false, Stop false (dna.go:106). */
ko_amino ko_gcode_bacteria(const uint8_t *codon)
{
    static const char AAS[]    = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
    static const char STARTS[] = "---M---------------M------------MMMM---------------M------------";
    ko_amino z = { 0, 0, 0 };
    int a = ko_nt_index(codon[0]), b = ko_nt_index(codon[1]), c = ko_nt_index(codon[2]);
    if (a < 0 || b < 0 || c < 0) return z;
    int idx = a * 16 + b * 4 + c;
    ko_amino r;
    r.aa = AAS[idx];
    r.start = STARTS[idx] == 'M';
    r.stop = AAS[idx] == '*';
    return r;
}

/* for tests: codon given as 3 chars -> packed (aa | start<<8 | stop<<9) */
uint32_t ko_gcode_bacteria_packed(const uint8_t *codon)
{
    ko_amino r = ko_gcode_bacteria(codon);
    return (uint32_t)(uint8_t)r.aa | ((uint32_t)r.start << 8) | ((uint32_t)r.stop << 9);
}

/* strings.ToLower on ASCII (dna.go:59,68).  Bytes >= 0x80 are out of scope. */
static void ko_to_lower(uint8_t *s, size_t n)
{
    for (size_t i = 0; i < n; i++) if (s[i] >= 'A' && s[i] <= 'Z') s[i] = (uint8_t)(s[i] + 32);
}

/* ReverseComplement, dna.go:55-63: reverse, then a<->t g<->c, others unchanged. */
static void ko_reverse_complement(const uint8_t *dna, size_t n, uint8_t *out)
{
    for (size_t i = 0; i < n; i++) {
        uint8_t c = dna[n - 1 - i];
        if (c >= 'A' && c <= 'Z') c = (uint8_t)(c + 32);
        switch (c) { case 'a': c = 't'; break; case 't': c = 'a'; break;
                     case 'g': c = 'c'; break; case 'c': c = 'g'; break; }
        out[i] = c;
    }
}

/* GetFrame, dna.go:183-196.  Returns pointer into buf (dna or its revcomp) and length.
 * The reference panics (slice out of range) when len(dna) < startPos; we return
 * an empty frame there (documented divergence: reads shorter than 3 nt). */
static const uint8_t *ko_get_frame(int frame_number, const uint8_t *dna, const uint8_t *rc,
                                   long n, long *frame_len)
{
    const uint8_t *src = dna;
    if (frame_number < 0) { src = rc; frame_number = -frame_number; }
    long start_pos = frame_number - 1;
    long len_frame = n - start_pos;
    if (len_frame < 0) { *frame_len = 0; return src; }
    long end_pos = n - (len_frame % 3);
    *frame_len = end_pos - start_pos;
    return src + start_pos;
}

/* ORF record.  Location fields as dna.go:35-45. */
typedef struct {
    int32_t start_position;
    int32_t end_position;
    int32_t plus_strand;
    uint32_t aa_off, aa_len;   /* into the aa arena   */
    uint32_t sa_off, sa_len;   /* into the starts_alternative arena */
} ko_orf;

typedef struct {
    ko_orf *orfs; size_t n_orfs, cap_orfs;
    uint8_t *aa;  size_t n_aa, cap_aa;
    int32_t *sa;  size_t n_sa, cap_sa;
} ko_orf_list;

static void *ko_grow(void *p, size_t *cap, size_t need, size_t elt)
{
    if (need <= *cap) return p;
    size_t nc = *cap ? *cap * 2 : 64;
    while (nc < need) nc *= 2;
    p = realloc(p, nc * elt);
    *cap = nc;
    return p;
}

ko_orf_list *ko_orf_list_new(void) { return (ko_orf_list *)calloc(1, sizeof(ko_orf_list)); }
void ko_orf_list_free(ko_orf_list *l) { if (!l) return; free(l->orfs); free(l->aa); free(l->sa); free(l); }
void ko_orf_list_clear(ko_orf_list *l) { l->n_orfs = l->n_aa = l->n_sa = 0; }
size_t ko_orf_list_count(const ko_orf_list *l) { return l->n_orfs; }
const ko_orf *ko_orf_list_orfs(const ko_orf_list *l) { return l->orfs; }
const uint8_t *ko_orf_list_aa(const ko_orf_list *l) { return l->aa; }
const int32_t *ko_orf_list_sa(const ko_orf_list *l) { return l->sa; }
size_t ko_orf_list_aa_len(const ko_orf_list *l) { return l->n_aa; }
size_t ko_orf_list_sa_len(const ko_orf_list *l) { return l->n_sa; }

/* GetORFs, dna.go:65-181.  The geneticCode argument is ignored by the
 * reference (dna.go:106 always uses gcodeBacteria), so it is not taken here.
 * ORFs of this read are APPENDED to `out`, ordered as the reference's final
 * sort.Slice does, with ties kept in frame/emission order (sort.Slice is not
 * stable; for <= 6 ORFs Go's implementation is an insertion sort, i.e. this). */
void ko_get_orfs(const uint8_t *dna_in, long n, ko_orf_list *out)
{
    static const int frame_start_position[6] = { 0, 1, 2, 0, 1, 2 }; /* dna.go:25 */
    static const int frame_numbers[6] = { 1, 2, 3, -1, -2, -3 };     /* dna.go:69-76 */
    uint8_t *dna = (uint8_t *)malloc((size_t)n + 1);
    uint8_t *rc = (uint8_t *)malloc((size_t)n + 1);
    memcpy(dna, dna_in, (size_t)n);
    ko_to_lower(dna, (size_t)n);                                     /* dna.go:68 */
    ko_reverse_complement(dna, (size_t)n, rc);
    size_t first_orf = out->n_orfs;

    /* scratch cds / starts for the ORF being built */
    size_t cds_cap = (size_t)n / 3 + 8;
    uint8_t *cds = (uint8_t *)malloc(cds_cap);
    int32_t *starts = (int32_t *)malloc(cds_cap * sizeof(int32_t));

    for (int frame_pos = 0; frame_pos < 6; frame_pos++) {            /* dna.go:78 */
        long frame_len;
        const uint8_t *frame_seq = ko_get_frame(frame_numbers[frame_pos], dna, rc, n, &frame_len);
        int start_pos = frame_start_position[frame_pos];             /* dna.go:80 */
        int plus_strand = frame_pos <= 2;                            /* dna.go:81 */
        long abs_pos = frame_pos;                                    /* dna.go:82 */
        if (!plus_strand) abs_pos = n - start_pos - 1;               /* dna.go:84 */
        long current_pos = 0;
        long orf_start_position = abs_pos + 1;                       /* dna.go:89 */
        size_t n_cds = 0, n_starts = 0;
        int inside_orf = 1;                                          /* dna.go:98 */
        long current_aa_pos = 0;                                     /* dna.go:102 */

        for (long i = 0; i < frame_len - (frame_len % 3); i += 3) {  /* dna.go:104 */
            current_pos = i;
            ko_amino current_aa = ko_gcode_bacteria(frame_seq + i);  /* dna.go:106 */
            if (current_aa.start) {
                if (!inside_orf) {                                   /* dna.go:109-116 */
                    inside_orf = 1;
                    current_aa_pos = 0;
                    orf_start_position = frame_pos + i + 1;
                    if (!plus_strand) orf_start_position = n - (frame_pos + i) + 3;
                    starts[n_starts++] = (int32_t)current_aa_pos;
                } else {
                    starts[n_starts++] = (int32_t)current_aa_pos;    /* dna.go:119 */
                }
            }
            if (inside_orf && current_aa.aa) cds[n_cds++] = (uint8_t)current_aa.aa; /* dna.go:123-125 */
            if (current_aa.stop) {                                   /* dna.go:127 */
                if (inside_orf && n_cds >= MIN_LEN_CDS) {
                    long end_pos = i + 3 + frame_pos;                /* dna.go:129 */
                    if (!plus_strand) end_pos = orf_start_position - ((long)n_cds * 3) + 1; /* :131 */
                    out->orfs = (ko_orf *)ko_grow(out->orfs, &out->cap_orfs, out->n_orfs + 1, sizeof(ko_orf));
                    out->aa = (uint8_t *)ko_grow(out->aa, &out->cap_aa, out->n_aa + n_cds, 1);
                    out->sa = (int32_t *)ko_grow(out->sa, &out->cap_sa, out->n_sa + n_starts + 1, sizeof(int32_t));
                    ko_orf *o = &out->orfs[out->n_orfs++];
                    o->start_position = (int32_t)orf_start_position;
                    o->end_position = (int32_t)end_pos;
                    o->plus_strand = plus_strand;
                    o->aa_off = (uint32_t)out->n_aa; o->aa_len = (uint32_t)n_cds;
                    memcpy(out->aa + out->n_aa, cds, n_cds); out->n_aa += n_cds;
                    o->sa_off = (uint32_t)out->n_sa; o->sa_len = (uint32_t)n_starts;
                    memcpy(out->sa + out->n_sa, starts, n_starts * sizeof(int32_t)); out->n_sa += n_starts;
                }
                orf_start_position = 0;                              /* dna.go:138-147 */
                n_starts = 0;
                n_cds = 0;                                           /* dna.go:148 */
                inside_orf = 0;                                      /* dna.go:149 */
            }
            current_aa_pos += 1;                                     /* dna.go:152 */
        }

        if (inside_orf && n_cds >= MIN_LEN_CDS) {                    /* dna.go:155-163 */
            long end_pos = current_pos + 3 + frame_pos;
            if (!plus_strand) end_pos = orf_start_position - ((long)n_cds * 3) + 1;
            out->orfs = (ko_orf *)ko_grow(out->orfs, &out->cap_orfs, out->n_orfs + 1, sizeof(ko_orf));
            out->aa = (uint8_t *)ko_grow(out->aa, &out->cap_aa, out->n_aa + n_cds, 1);
            out->sa = (int32_t *)ko_grow(out->sa, &out->cap_sa, out->n_sa + n_starts + 1, sizeof(int32_t));
            ko_orf *o = &out->orfs[out->n_orfs++];
            o->start_position = (int32_t)orf_start_position;
            o->end_position = (int32_t)end_pos;
            o->plus_strand = plus_strand;
            o->aa_off = (uint32_t)out->n_aa; o->aa_len = (uint32_t)n_cds;
            memcpy(out->aa + out->n_aa, cds, n_cds); out->n_aa += n_cds;
            o->sa_off = (uint32_t)out->n_sa; o->sa_len = (uint32_t)n_starts;
            memcpy(out->sa + out->n_sa, starts, n_starts * sizeof(int32_t)); out->n_sa += n_starts;
        }
    }

    /* sort.Slice, dna.go:167-177: key = EndPosition (plus) / StartPosition (minus).
     * Stable insertion sort over this read's ORFs. */
    for (size_t a = first_orf + 1; a < out->n_orfs; a++) {
        ko_orf t = out->orfs[a];
        long tk = t.plus_strand ? t.end_position : t.start_position;
        size_t b = a;
        while (b > first_orf) {
            ko_orf *p = &out->orfs[b - 1];
            long pk = p->plus_strand ? p->end_position : p->start_position;
            if (tk < pk) { out->orfs[b] = *p; b--; } else break;
        }
        out->orfs[b] = t;
    }
    free(cds); free(starts); free(dna); free(rc);
}

/* SizeInKmer of a query/ORF: search.go:290-293,314-317; search_fastq.go:81-92 */
int32_t ko_size_in_kmer(const uint8_t *seq, long len)
{
    int32_t s = (int32_t)(len - KMER_SIZE + 1);
    if (len > 0 && seq[len - 1] == '*') s--;
    return s;
}

/* ------------------------------------------------------------------------ */
/* Index: makedb emit loops + indexdb set de-dup                              */
/*   pkg/makedb/inputFASTA.go:245-248, inputTSV.go:236-239 (emit)            */
/*   pkg/indexdb/indexdb.go:92-132 + kv_store.go:284-305 (unique ids per key)*/
/* Logical result: key -> set<proteinId>.  Stored as sorted unique (key,id). */
/* ------------------------------------------------------------------------ */

typedef struct {
    uint64_t *pairs;   /* key<<32 | id, sorted ascending, unique */
    uint64_t n;
} ko_index;

static int ko_cmp_u64(const void *a, const void *b)
{
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : x > y;
}

ko_index *ko_index_from_pairs(const uint32_t *keys, const uint32_t *ids, uint64_t n)
{
    ko_index *ix = (ko_index *)calloc(1, sizeof *ix);
    ix->pairs = (uint64_t *)malloc((n ? n : 1) * sizeof(uint64_t));
    for (uint64_t i = 0; i < n; i++) ix->pairs[i] = ((uint64_t)keys[i] << 32) | ids[i];
    qsort(ix->pairs, n, sizeof(uint64_t), ko_cmp_u64);
    uint64_t m = 0;
    for (uint64_t i = 0; i < n; i++)
        if (m == 0 || ix->pairs[m - 1] != ix->pairs[i]) ix->pairs[m++] = ix->pairs[i];
    ix->n = m;
    return ix;
}

/* Emit loop over proteins: every window seq[i:i+7], i in [0,len-7] of every
 * protein with len >= 7 (inputFASTA.go:228-230,245-248; inputTSV.go:139,236-239).
 * `ids[p]` is the protein id the parser assigned (parser-specific rules are
 * applied by the caller: see ko_fasta_ids / TSV = running index). */
ko_index *ko_index_from_proteins(const uint8_t *seqs, const uint64_t *offsets, uint32_t n_prot,
                                 const uint32_t *ids)
{
    uint64_t total = 0;
    for (uint32_t p = 0; p < n_prot; p++) {
        uint64_t len = offsets[p + 1] - offsets[p];
        if (len >= KMER_SIZE) total += len - KMER_SIZE + 1;
    }
    uint32_t *keys = (uint32_t *)malloc((total ? total : 1) * sizeof(uint32_t));
    uint32_t *vals = (uint32_t *)malloc((total ? total : 1) * sizeof(uint32_t));
    uint64_t k = 0;
    for (uint32_t p = 0; p < n_prot; p++) {
        const uint8_t *s = seqs + offsets[p];
        long len = (long)(offsets[p + 1] - offsets[p]);
        if (len < KMER_SIZE) continue;                       /* inputFASTA.go:228 */
        for (long i = 0; i < len - KMER_SIZE + 1; i++) {     /* inputFASTA.go:245 */
            keys[k] = ko_encode_kmer(s + i, KMER_SIZE);
            vals[k] = ids[p];
            k++;
        }
    }
    ko_index *ix = ko_index_from_pairs(keys, vals, k);
    free(keys); free(vals);
    return ix;
}

void ko_index_free(ko_index *ix) { if (!ix) return; free(ix->pairs); free(ix); }
uint64_t ko_index_n_pairs(const ko_index *ix) { return ix->n; }
const uint64_t *ko_index_pairs(const ko_index *ix) { return ix->pairs; }

/* FASTA parser id rule, inputFASTA.go:95-124 (offset=0, length=MaxUint):
 * proteinNb is incremented at every header and the PREVIOUS entry is queued
 * with the new proteinNb, so record k (1-based) gets id k+1; the last record
 * is queued at EOF with proteinNb unchanged, i.e. id N.  Records N-1 and N
 * share id N (reference behaviour, reproduced). */
void ko_fasta_ids(uint32_t n_records, uint32_t *ids)
{
    for (uint32_t k = 1; k <= n_records; k++) ids[k - 1] = (k < n_records) ? k + 1 : n_records;
}

/* exact-match point read (kv_store.go:179-204): first pair with this key */
static uint64_t ko_lower_bound(const ko_index *ix, uint32_t key)
{
    uint64_t lo = 0, hi = ix->n, target = (uint64_t)key << 32;
    while (lo < hi) {
        uint64_t mid = lo + (hi - lo) / 2;
        if (ix->pairs[mid] < target) lo = mid + 1; else hi = mid;
    }
    return lo;
}

/* number of ids under key, ids copied to out (if non-NULL, up to cap) */
uint32_t ko_index_get(const ko_index *ix, uint32_t key, uint32_t *out, uint32_t cap)
{
    uint64_t i = ko_lower_bound(ix, key);
    uint32_t c = 0;
    while (i < ix->n && (uint32_t)(ix->pairs[i] >> 32) == key) {
        if (out && c < cap) out[c] = (uint32_t)ix->pairs[i];
        c++; i++;
    }
    return c;
}

/* ------------------------------------------------------------------------ */
/* Lookup + count: pkg/search/search.go:414-452, search_protein.go:94-98     */
/* ------------------------------------------------------------------------ */

typedef struct {
    uint32_t *pid; int64_t *kmatch; size_t n, cap, km_cap;
    uint8_t *pos;        /* n * size_in_kmer bools (PositionHits), optional */
    size_t pos_cap;
    uint64_t n_lookup, n_found, n_post;   /* work counters for the bench   */
    /* scratch */
    uint64_t *tmp; size_t tmp_cap;
} ko_result;

ko_result *ko_result_new(void) { return (ko_result *)calloc(1, sizeof(ko_result)); }
void ko_result_free(ko_result *r) { if (!r) return; free(r->pid); free(r->kmatch); free(r->pos); free(r->tmp); free(r); }
size_t ko_result_n(const ko_result *r) { return r->n; }
const uint32_t *ko_result_pid(const ko_result *r) { return r->pid; }
const int64_t *ko_result_kmatch(const ko_result *r) { return r->kmatch; }
const uint8_t *ko_result_pos(const ko_result *r) { return r->pos; }
uint64_t ko_result_n_lookup(const ko_result *r) { return r->n_lookup; }
uint64_t ko_result_n_found(const ko_result *r) { return r->n_found; }
uint64_t ko_result_n_post(const ko_result *r) { return r->n_post; }

/* One query: keys for positions k in [0,SizeInKmer) (search_protein.go:95-98),
 * each looked up (search.go:421-429), every id of the set incremented once per
 * position (search.go:431-432), PositionHits[id][pos]=true (search.go:442-452).
 * Hits are returned ordered by (Kmatch desc, pid asc) — the reference's order
 * among ties is nondeterministic (sync.Map range + unstable sort.Sort,
 * search.go:132-152); this is one valid permutation. */
void ko_search_query(const ko_index *ix, const uint8_t *seq, long len, int32_t size_in_kmer,
                     int want_positions, ko_result *r)
{
    (void)len;
    r->n = 0;
    size_t nt = 0;
    for (int32_t k = 0; k < size_in_kmer; k++) {
        uint32_t key = ko_encode_kmer(seq + k, KMER_SIZE);
        r->n_lookup++;
        uint64_t i = ko_lower_bound(ix, key);
        int found = 0;
        while (i < ix->n && (uint32_t)(ix->pairs[i] >> 32) == key) {
            r->tmp = (uint64_t *)ko_grow(r->tmp, &r->tmp_cap, nt + 1, sizeof(uint64_t));
            r->tmp[nt++] = ((uint64_t)(uint32_t)ix->pairs[i] << 32) | (uint32_t)k;  /* (id,pos) */
            found = 1; r->n_post++; i++;
        }
        r->n_found += (uint64_t)found;
    }
    qsort(r->tmp, nt, sizeof(uint64_t), ko_cmp_u64);
    /* run-length over ids -> Counter values */
    size_t nh = 0;
    for (size_t a = 0; a < nt;) {
        size_t b = a;
        while (b < nt && (r->tmp[b] >> 32) == (r->tmp[a] >> 32)) b++;
        r->pid = (uint32_t *)ko_grow(r->pid, &r->cap, nh + 1, sizeof(uint32_t));
        r->kmatch = (int64_t *)ko_grow(r->kmatch, &r->km_cap, nh + 1, sizeof(int64_t));
        r->pid[nh] = (uint32_t)(r->tmp[a] >> 32);
        r->kmatch[nh] = (int64_t)(b - a);
        nh++;
        a = b;
    }
    /* order: Kmatch desc, pid asc (insertion into index permutation) */
    uint32_t *perm = (uint32_t *)malloc((nh ? nh : 1) * sizeof(uint32_t));
    for (size_t a = 0; a < nh; a++) perm[a] = (uint32_t)a;
    /* simple merge-free stable sort by kmatch desc: pids are already ascending */
    for (size_t a = 1; a < nh; a++) {
        uint32_t t = perm[a]; size_t b = a;
        while (b > 0 && r->kmatch[perm[b - 1]] < r->kmatch[t]) { perm[b] = perm[b - 1]; b--; }
        perm[b] = t;
    }
    uint32_t *pid2 = (uint32_t *)malloc((nh ? nh : 1) * sizeof(uint32_t));
    int64_t *km2 = (int64_t *)malloc((nh ? nh : 1) * sizeof(int64_t));
    for (size_t a = 0; a < nh; a++) { pid2[a] = r->pid[perm[a]]; km2[a] = r->kmatch[perm[a]]; }
    if (want_positions && size_in_kmer > 0) {
        size_t need = nh * (size_t)size_in_kmer;
        r->pos = (uint8_t *)ko_grow(r->pos, &r->pos_cap, need ? need : 1, 1);
        memset(r->pos, 0, need);
        /* rank of each original hit in the output order */
        uint32_t *rank = (uint32_t *)malloc((nh ? nh : 1) * sizeof(uint32_t));
        for (size_t a = 0; a < nh; a++) rank[perm[a]] = (uint32_t)a;
        size_t h = 0;
        for (size_t a = 0; a < nt;) {
            size_t b = a;
            while (b < nt && (r->tmp[b] >> 32) == (r->tmp[a] >> 32)) {
                r->pos[(size_t)rank[h] * (size_t)size_in_kmer + (uint32_t)r->tmp[b]] = 1;
                b++;
            }
            h++; a = b;
        }
        free(rank);
    }
    memcpy(r->pid, pid2, nh * sizeof(uint32_t));
    memcpy(r->kmatch, km2, nh * sizeof(int64_t));
    free(perm); free(pid2); free(km2);
    r->n = nh;
}

/* FilterResults, search.go:189-220.  Hits must be sorted by Kmatch desc.
 * Returns the number of hits kept (a prefix). */
int64_t ko_filter_results(const int64_t *kmatch, int64_t n_hits, int32_t size_in_kmer,
                          double min_k_ratio, int64_t min_k_match, int64_t max_results)
{
    int64_t last_good = n_hits - 1;                                  /* search.go:192 */
    for (int64_t i = 0; i < n_hits; i++) {
        if (((double)kmatch[i] / (double)size_in_kmer) < min_k_ratio || kmatch[i] < min_k_match) {
            if (last_good == n_hits - 1) last_good = i - 1;          /* search.go:196-198 */
        }
    }
    if (last_good >= max_results) last_good = max_results - 1;       /* search.go:203-204 */
    if (last_good < 0) return 0;
    return last_good + 1;
}

/* SetBestStartCodon, dna.go:198-272.  Inputs: hits sorted (Kmatch desc),
 * position bools [n_hits][size_in_kmer], StartsAlternative.  Returns bestStart
 * (number of residues trimmed from the ORF head; 0 = unchanged) and updates
 * start_position / size_in_kmer like dna.go:252-267.  seq/len describe the ORF
 * amino-acid string (for the trailing '*' rule). */
int32_t ko_set_best_start_codon(const int64_t *kmatch, int64_t n_hits, const uint8_t *pos,
                                int32_t size_in_kmer_in, const int32_t *starts_alt, int32_t n_starts,
                                int32_t plus_strand, const uint8_t *seq, long len,
                                int32_t *start_position, int32_t *size_in_kmer_out)
{
    *size_in_kmer_out = size_in_kmer_in;
    if (n_starts < 1) return 0;                                      /* dna.go:210-212 */
    /* bestHits: dna.go:203-208 */
    int64_t best_hit_score = 0;
    int64_t n_best = 0;
    int64_t *best = (int64_t *)malloc((n_hits ? n_hits : 1) * sizeof(int64_t));
    for (int64_t h = 0; h < n_hits; h++)
        if (kmatch[h] >= best_hit_score) { best_hit_score = kmatch[h]; best[n_best++] = h; }
    int32_t best_start = starts_alt[0];
    int32_t first_start = starts_alt[0];
    int64_t first_best_hit_pos = 999999999;                          /* dna.go:219 */
    int exit_ = 0;
    for (int64_t b = 0; b < n_best; b++) {                           /* dna.go:225-237 */
        const uint8_t *p = pos + (size_t)best[b] * (size_t)size_in_kmer_in;
        for (int32_t i = 0; i < size_in_kmer_in; i++) {
            if (p[i]) {
                if (i < first_best_hit_pos) first_best_hit_pos = i;
                exit_ = 1;
            }
            if (exit_) break;
        }
    }
    for (int32_t s = 0; s < n_starts; s++) {                         /* dna.go:240-249 */
        if (starts_alt[s] <= first_best_hit_pos) best_start = starts_alt[s];
        else break;
    }
    free(best);
    if (best_start != first_start) {                                 /* dna.go:252-267 */
        if (plus_strand) *start_position = *start_position + 3 * best_start;
        else *start_position = *start_position - 3 * best_start;
        long new_len = len - best_start;
        int32_t s = (int32_t)(new_len - KMER_SIZE + 1);
        if (new_len > 0 && seq[len - 1] == '*') s--;
        *size_in_kmer_out = s;
        return best_start;
    }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* Batch drivers used as the timed CPU baseline (bench.py cpu_baseline leg)   */
/* ------------------------------------------------------------------------ */

/* Protein batch: search_protein.go:70-114 per query; returns Σ hits (a
 * checksum so the work cannot be elided) and accumulates counters in r. */
uint64_t ko_search_protein_batch(const ko_index *ix, const uint8_t *seqs, const uint64_t *offsets,
                                 uint32_t q_begin, uint32_t q_end, ko_result *r)
{
    uint64_t checksum = 0;
    for (uint32_t q = q_begin; q < q_end; q++) {
        const uint8_t *s = seqs + offsets[q];
        long len = (long)(offsets[q + 1] - offsets[q]);
        int32_t sz = ko_size_in_kmer(s, len);
        if (sz < 7) continue;                                        /* search_protein.go:74-76 */
        ko_search_query(ix, s, len, sz, 0, r);
        for (size_t h = 0; h < r->n; h++) checksum += (uint64_t)r->pid[h] * 31u + (uint64_t)r->kmatch[h];
    }
    return checksum;
}

/* Reads batch: search_fastq.go:72-118 per read (GetORFs then per-ORF search). */
uint64_t ko_search_reads_batch(const ko_index *ix, const uint8_t *seqs, const uint64_t *offsets,
                               uint32_t q_begin, uint32_t q_end, ko_result *r)
{
    uint64_t checksum = 0;
    ko_orf_list *ol = ko_orf_list_new();
    for (uint32_t q = q_begin; q < q_end; q++) {
        ko_orf_list_clear(ol);
        ko_get_orfs(seqs + offsets[q], (long)(offsets[q + 1] - offsets[q]), ol);
        for (size_t o = 0; o < ol->n_orfs; o++) {
            const uint8_t *s = ol->aa + ol->orfs[o].aa_off;
            long len = (long)ol->orfs[o].aa_len;
            int32_t sz = ko_size_in_kmer(s, len);
            ko_search_query(ix, s, len, sz, 1, r);
            for (size_t h = 0; h < r->n; h++) checksum += (uint64_t)r->pid[h] * 31u + (uint64_t)r->kmatch[h];
        }
    }
    ko_orf_list_free(ol);
    return checksum;
}
