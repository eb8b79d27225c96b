// kaamer_layout.h — HBM layout of the k-mer -> protein-id table, shared by the
// host builder and the gfx950 kernels.
//
// The table replaces the reference's two Badger stores on the read path
// (kmer_store: key -> kcombId, kcomb_store: kcombId -> KComb{ProteinKeys};
// pkg/kvstore/kv_stores.go:46-104, kcomb.proto) by
//
//   buckets : n_buckets x 128 B, each 8 CELLS of 16 B.  One random 128-byte request
//             costs the memory system the same as a 64-byte one (tools/
//             random_read_bench.hip: ~53e9 requests/s at either size), so a bucket
//             carries its keys' SHORT POSTINGS LISTS inline: 8 lanes x 16 B read one
//             bucket, the lane whose cell matches already holds the ids.
//   arena   : u32 words; only lists too long for the bucket live here, as bare
//             ids padded to 16 B and addressed in 16-B units; identical sets are
//             stored once (the KComb sharing of kcomb_store.go:42-85).
//
// cell = {key, w1, w2, w3}:
//   key = 0xFFFFFFFF             empty cell (cells of a bucket fill from 0 up, no deletions)
//   key = 0xFFFFFFFE             CONTINUATION of the cell before it: w1..w3 = ids 3..5
//   w1 bit31 = 0                 inline list: w1 = id0, w2 = id1, w3 = id2 (KH_NO_ID = unused);
//                                w3 bit31 set: the next cell continues the list
//   w1 bit31 = 1                 arena list: w1 & 0x7FFFFFFF = offset in 16-B units (never 0),
//                                w2 = number of ids (>= 4), w3 = id0; the arena holds id1 ...
// ids ascend inside a list; protein ids are < 0x7FFFFFFF.  The largest valid key is
// 0xE773B9D4 ("YYYYYYY"), so both markers are outside the key space.
//
// Open addressing on buckets: a key lives in the first bucket along home, home+1, ...
// (mod n_buckets) that had a free cell when it was inserted; a lookup stops at the
// first bucket whose last cell is empty.  A list of 4..6 ids takes two cells when
// that bucket has two free, otherwise one cell and the arena.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define KH_HD __host__ __device__ inline
#else
#define KH_HD inline
#endif

#define KH_EMPTY_KEY 0xFFFFFFFFu
#define KH_CONT_KEY 0xFFFFFFFEu
#define KH_EMPTY_PID 0xFFFFFFFFu   /* "no id" of the counting tables and result arrays */
#define KH_NO_ID 0x7FFFFFFFu       /* unused id word of a cell */
#define KH_MAX_PID 0x7FFFFFFEu
#ifndef KH_CELLS_PER_BUCKET
#define KH_CELLS_PER_BUCKET 4
#endif
#define KH_BUCKET_BYTES (16 * KH_CELLS_PER_BUCKET)
#define KH_ARENA_BIT 0x80000000u   /* in w1 */
#define KH_CONT_BIT 0x80000000u    /* in w3 of an inline head cell */
#define KH_INLINE_IDS 3            /* ids in a head cell */
#define KH_CELL_IDS_MAX 6          /* head + one continuation cell */
#define KH_IMAGE_MAGIC 0x32544B4852454D41ull /* "AMERHKT2" little-endian tag */
#define KH_IMAGE_VERSION (0x200 + KH_CELLS_PER_BUCKET)  /* layout 2, cells per bucket */

struct alignas(16) kh_cell { uint32_t key, w1, w2, w3; };
struct alignas(KH_BUCKET_BYTES) kh_bucket { kh_cell c[KH_CELLS_PER_BUCKET]; };

// murmur3 finalizer
KH_HD uint32_t kh_mix32(uint32_t h)
{
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}

// shard = hash prefix; home bucket = the rest of the hash scaled to n_buckets
KH_HD uint32_t kh_shard_of(uint32_t key, uint32_t n_shards)
{
    return (uint32_t)(((uint64_t)kh_mix32(key) * n_shards) >> 32);
}
KH_HD uint64_t kh_home_bucket(uint32_t key, uint32_t n_shards, uint64_t n_buckets)
{
    uint32_t rest = (uint32_t)((uint64_t)kh_mix32(key) * n_shards);  // low 32 bits
    return ((uint64_t)rest * n_buckets) >> 32;
}

// ---- k-mer codec (pkg/kvstore/k_store.go:39-117) in closed form -------------
// residue code: index in "ACDEFGHIKLMNPQRSTUVWY" (k_store.go:41), 30 = '.', 31 = any
// other byte.  pair(a,b) = 22 + 21*a + b (the counter of k_store.go:46-59); the table
// also holds {a,'.'} -> index of a (k_store.go:48-52), which the pair lookup of
// EncodeKmer (k_store.go:102-103) finds when the SECOND letter of a pair is '.';
// every other combination is a Go map miss and yields 0.
#define KH_CODE_UNKNOWN 31u
#define KH_CODE_DOT 30u
KH_HD uint32_t kh_residue_code(uint8_t c)
{
    switch (c) {
    case 'A': return 0;  case 'C': return 1;  case 'D': return 2;  case 'E': return 3;
    case 'F': return 4;  case 'G': return 5;  case 'H': return 6;  case 'I': return 7;
    case 'K': return 8;  case 'L': return 9;  case 'M': return 10; case 'N': return 11;
    case 'P': return 12; case 'Q': return 13; case 'R': return 14; case 'S': return 15;
    case 'T': return 16; case 'U': return 17; case 'V': return 18; case 'W': return 19;
    case 'Y': return 20;
    case '.': return KH_CODE_DOT;
    default: return KH_CODE_UNKNOWN;
    }
}
// (written as selects: the kernels evaluate this for every residue position, and exec-mask branches cost more than the arithmetic)
KH_HD uint32_t kh_pair(uint32_t a, uint32_t b)
{
    const uint32_t both = 22u + 21u * a + b;
    const uint32_t r = b < 21u ? both : (b == KH_CODE_DOT ? a : 0u);
    return a < 21u ? r : 0u;
}
KH_HD uint32_t kh_key_from_codes(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t c4,
                                 uint32_t c5, uint32_t c6)
{
    return (kh_pair(c0, c1) << 23) | (kh_pair(c2, c3) << 14) | (kh_pair(c4, c5) << 5) |
           (c6 < 21u ? c6 : 0u);
}

// ---- image header (file and in-memory) --------------------------------------
struct kh_image_header {
    uint64_t magic;
    uint32_t version;
    uint32_t kmer_size;
    uint32_t shard, n_shards;
    uint64_t n_buckets;
    uint64_t arena_words;
    uint64_t n_pairs, n_keys, n_inline, n_lists, max_list, n_displaced;
    uint32_t max_protein_id;
    uint32_t pad0;
    double load_factor;
    uint64_t n_cont;        /* keys whose list takes a continuation cell */
    uint64_t n_arena_keys;  /* keys whose list lives in the arena */
    uint8_t reserved[4096 - 120];
};
static_assert(sizeof(kh_image_header) == 4096, "header is one page");
