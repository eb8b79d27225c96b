// kaamer_layout.h — HBM layout of the k-mer -> protein-id table, shared by the
// host builder and the gfx950 kernels.
//
// The table replaces the reference's two Badger stores on the read path
// (kmer_store: key -> kcombId, kcomb_store: kcombId -> KComb{ProteinKeys};
// pkg/kvstore/kv_stores.go:46-104, kcomb.proto) by
//
//   buckets : n_buckets x 64 B, each 8 slots of {u32 key, u32 val}
//             (one HBM/fabric sector per probe; 4 lanes x 16 B read one bucket)
//   arena   : u32 words; a postings list is {count, id0, id1, ...} padded to
//             16 B and addressed in 16-B units; identical sets are stored once
//             (the KComb sharing of kcomb_store.go:42-85)
//
// slot.val : bit31 = 1  -> the key has ONE protein id, stored inline (low 31 bits)
//            bit31 = 0  -> offset of the postings list, in 16-B units (never 0)
// slot.key : 0xFFFFFFFF = empty (the largest valid key is 0xE773B9D4, "YYYYYYY")
//
// Open addressing: a key lives in the first bucket with a free slot along
// home, home+1, ... (mod n_buckets); a lookup stops at the first bucket that
// has an empty slot.  No deletions.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define KH_HD __host__ __device__ inline
#else
#define KH_HD inline
#endif

#define KH_EMPTY_KEY 0xFFFFFFFFu
#define KH_EMPTY_PID 0xFFFFFFFFu
#define KH_SLOTS_PER_BUCKET 8
#define KH_BUCKET_BYTES 64
#define KH_INLINE_BIT 0x80000000u
#define KH_IMAGE_MAGIC 0x31544B4852454D41ull /* "AMERHKT1" little-endian tag */
#define KH_IMAGE_VERSION 1

struct kh_slot { uint32_t key, val; };
struct alignas(64) kh_bucket { kh_slot s[KH_SLOTS_PER_BUCKET]; };

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
// (written as selects: the kernels evaluate this for every residue position, and exec-mask
// branches cost more than the arithmetic)
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
    uint8_t reserved[4096 - 104];
};
static_assert(sizeof(kh_image_header) == 4096, "header is one page");
