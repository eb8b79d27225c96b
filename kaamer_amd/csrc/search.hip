// search.hip — gfx950 kernels and the C-ABI entry points of the k-mer search path.
//
// What runs on the device, per batch (all on the caller's stream, no host sync):
//
//   prep_protein_kernel   Query.SizeInKmer etc. for protein records
//                         (search.go:290-293; search_protein.go:70-76)
//   kmer_search_kernel    THE hot loop: sliding 7-mer encode (k_store.go:91-117,
//                         search_protein.go:94-98), bucket probe (replaces the
//                         two Badger point reads of search.go:421-429), postings
//                         expansion and per-protein counting (search.go:431-436,
//                         442-452) in an LDS hash table, ballot/prefix-sum
//                         compaction of the hit list
//   scan / gather kernels hit lists -> CSR in query order
//
// One wavefront (64 lanes) owns one query at a time; a workgroup is exactly one
// wave, so __syncthreads() is a wave-local fence and waves never wait for each
// other.  This is integer hashing/indexing: no MFMA; the bound is HBM (random
// 64-byte bucket reads + postings).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "kaamer_internal.h"

// ------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

int kaamer_fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIPCHK(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return kaamer_fail(KAAMER_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// ------------------------------------------------------------------------------------
// device-side structures
// ------------------------------------------------------------------------------------
struct kaamer_index {
    int device;
    kh_image_header hdr;
    kh_bucket *d_buckets;
    uint32_t *d_arena;
};

enum { ST_POOL_FULL = 1u, ST_OVF_LIST_FULL = 2u, ST_QUERY_CAP = 4u, ST_AA_CAP = 8u, ST_OVERFLOW_UNSERVED = 16u };
enum { CTR_IN = 0, CTR_QUERIES, CTR_LOOKUP, CTR_PROBE, CTR_FOUND, CTR_POST, CTR_HITS, CTR_OVERFLOW, CTR_LISTS, CTR_LIST_IDS, CTR_N };
static_assert(sizeof(kaamer_counters) == CTR_N * 8, "counter layout");
#define CTR_REPLICAS 64

struct SearchParams {
    const uint4 *table;  // buckets viewed as 4 x uint4 each
    uint64_t n_buckets;
    uint32_t n_shards, shard;
    const uint32_t *arena;
    const uint8_t *residues;  // input seqs (protein) or ORF amino acids (reads)
    const kaamer_query_meta *q;
    const uint32_t *d_nq;
    int32_t min_size;  // search_protein.go:74: protein queries with SizeInKmer < 7 are dropped
    // per-query result location in the pool
    uint64_t *q_start;
    uint32_t *q_cnt;
    uint32_t *pool_pid, *pool_km, *pool_fp;
    uint64_t pool_cap;
    unsigned long long *pool_cursor;
    uint32_t *ovf_list;
    uint32_t *ovf_count;
    uint32_t ovf_cap;
    unsigned long long *counters;  // [CTR_REPLICAS][CTR_N]
    uint32_t *status;
};

#define POOL_CHUNK 1024u
#define MAX_TIMED_CALLS 1024u
#define COOP_LIST_THRESHOLD 48u

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// LDS counting table: open addressing keyed by protein id.
template <int LOG2CAP>
__device__ __forceinline__ bool table_add(volatile uint32_t *keys, uint32_t *cnt, uint32_t *minpos,
                                          uint32_t *nd, uint32_t pid, uint32_t pos)
{
    constexpr uint32_t MASK = (1u << LOG2CAP) - 1u;
    uint32_t h = (pid * 0x9E3779B1u) >> (32 - LOG2CAP);
    for (uint32_t t = 0; t <= MASK; t++) {
        uint32_t k = keys[h];
        if (k == KH_EMPTY_PID) {
            uint32_t old = atomicCAS((uint32_t *)&keys[h], KH_EMPTY_PID, pid);
            if (old == KH_EMPTY_PID) { atomicAdd(nd, 1u); k = pid; }
            else k = old;
        }
        if (k == pid) {
            atomicAdd(&cnt[h], 1u);
            atomicMin(&minpos[h], pos);
            return true;
        }
        h = (h + 1u) & MASK;
    }
    return false;
}

template <int LOG2CAP>
__global__ __launch_bounds__(64) void kmer_search_kernel(SearchParams p)
{
    constexpr int CAP = 1 << LOG2CAP;
    constexpr uint32_t CAP_LIMIT = (uint32_t)(CAP - CAP / 4);  // leave 25 % free
    __shared__ uint32_t t_keys[CAP];
    __shared__ uint32_t t_cnt[CAP];
    __shared__ uint32_t t_min[CAP];
    __shared__ uint8_t s_lut[256];
    __shared__ uint8_t s_stage[128];
    __shared__ uint32_t s_nd, s_ovf;

    const uint32_t lane = lane_id();
    const uint32_t wave = blockIdx.x;
    const uint32_t n_waves = gridDim.x;
    const uint32_t nq = *p.d_nq;

    for (uint32_t i = lane; i < 256; i += 64) s_lut[i] = (uint8_t)kh_residue_code((uint8_t)i);

    unsigned long long c_in = 0, c_q = 0, c_lookup = 0, c_probe = 0, c_found = 0, c_post = 0, c_hits = 0, c_ovf = 0, c_lists = 0, c_lids = 0;
    uint64_t chunk_base = 0;
    uint32_t chunk_left = 0;

    for (uint32_t q = wave; q < nq; q += n_waves) {
        const kaamer_query_meta qm = p.q[q];
        const int32_t size = qm.size_in_kmer;
        if (size < p.min_size || size <= 0) {
            if (lane == 0) { p.q_cnt[q] = 0; p.q_start[q] = 0; }
            continue;
        }
        const uint8_t *res = p.residues + qm.aa_off;
        for (uint32_t i = lane; i < (uint32_t)CAP; i += 64) { t_keys[i] = KH_EMPTY_PID; t_cnt[i] = 0; t_min[i] = 0xFFFFFFFFu; }
        if (lane == 0) { s_nd = 0; s_ovf = 0; }
        __syncthreads();
        if (lane == 0) { c_q++; c_in += (unsigned long long)size + 6; }

        bool overflow = false;
        for (int32_t c0 = 0; c0 < size && !overflow; c0 += 64) {
            const int32_t n_here = min(64, size - c0);
            // stage residue codes of this window: n_here + 6 residues
            if ((int32_t)lane < n_here + 6) s_stage[lane] = s_lut[res[c0 + lane]];
            if ((int32_t)lane + 64 < n_here + 6) s_stage[lane + 64] = s_lut[res[c0 + 64 + lane]];
            __syncthreads();
            uint32_t key = KH_EMPTY_KEY;
            if ((int32_t)lane < n_here)
                key = kh_key_from_codes(s_stage[lane], s_stage[lane + 1], s_stage[lane + 2], s_stage[lane + 3],
                                        s_stage[lane + 4], s_stage[lane + 5], s_stage[lane + 6]);
            // sharded index: this device probes only the keys it owns
            if (p.n_shards > 1 && key != KH_EMPTY_KEY && kh_shard_of(key, p.n_shards) != p.shard) key = KH_EMPTY_KEY;
            uint32_t bucket = (key != KH_EMPTY_KEY) ? (uint32_t)kh_home_bucket(key, p.n_shards, p.n_buckets) : 0u;

            // ---- probe: 4 lanes read one 64-B bucket (16 B each); round j serves the
            // k-mers of lanes 16j..16j+15; lane 4g+j ends up owning k-mer 16j+g
            uint4 ld[4];
            uint32_t rkey[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int src = 16 * j + (int)(lane >> 2);
                const uint32_t kb = __shfl(bucket, src, 64);
                rkey[j] = __shfl(key, src, 64);
                ld[j] = make_uint4(KH_EMPTY_KEY, 0, KH_EMPTY_KEY, 0);
                if (rkey[j] != KH_EMPTY_KEY) ld[j] = p.table[(uint64_t)kb * 4 + (lane & 3u)];
            }
            uint32_t okey = KH_EMPTY_KEY, oval = 0, obucket = 0;
            bool oempty = true;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t kk = rkey[j];
                uint32_t r = (ld[j].x == kk) ? ld[j].y : ((ld[j].z == kk) ? ld[j].w : 0u);
                uint32_t e = (ld[j].x == KH_EMPTY_KEY || ld[j].z == KH_EMPTY_KEY) ? 1u : 0u;
                r |= __shfl_xor(r, 1, 64); e |= __shfl_xor(e, 1, 64);
                r |= __shfl_xor(r, 2, 64); e |= __shfl_xor(e, 2, 64);
                const int src = 16 * j + (int)(lane >> 2);
                const uint32_t kb = __shfl(bucket, src, 64);
                if ((lane & 3u) == (uint32_t)j) { okey = kk; oval = r; oempty = (e != 0u); obucket = kb; }
            }
            const bool valid = okey != KH_EMPTY_KEY;
            const uint32_t opos = (uint32_t)c0 + 16u * (lane & 3u) + (lane >> 2);
            if (valid) { c_lookup++; c_probe++; }
            // rare: home bucket full and key not in it -> walk following buckets alone
            if (valid && oval == 0u && !oempty) {
                for (uint64_t tries = 1; tries < p.n_buckets; tries++) {
                    obucket = (obucket + 1u == (uint32_t)p.n_buckets) ? 0u : obucket + 1u;
                    c_probe++;
                    bool e = false;
#pragma unroll
                    for (int s = 0; s < 4; s++) {
                        const uint4 v = p.table[(uint64_t)obucket * 4 + s];
                        if (v.x == okey) oval = v.y;
                        if (v.z == okey) oval = v.w;
                        e = e || v.x == KH_EMPTY_KEY || v.z == KH_EMPTY_KEY;
                    }
                    if (oval != 0u || e) break;
                }
            }
            // ---- postings -> counting table
            const bool found = valid && oval != 0u;
            bool coop = false;
            uint32_t lcnt = 0;
            bool ok = true;
            if (found) {
                c_found++;
                if (oval & KH_INLINE_BIT) {
                    c_post++;
                    ok = table_add<LOG2CAP>(t_keys, t_cnt, t_min, &s_nd, oval & ~KH_INLINE_BIT, opos);
                } else {
                    const uint4 *a = reinterpret_cast<const uint4 *>(p.arena) + oval;
                    const uint4 w = a[0];
                    lcnt = w.x;
                    c_post += lcnt;
                    c_lists++;
                    c_lids += lcnt;
                    if (lcnt > COOP_LIST_THRESHOLD) {
                        coop = true;
                    } else {
                        ok = table_add<LOG2CAP>(t_keys, t_cnt, t_min, &s_nd, w.y, opos);
                        if (ok && lcnt > 1) ok = table_add<LOG2CAP>(t_keys, t_cnt, t_min, &s_nd, w.z, opos);
                        if (ok && lcnt > 2) ok = table_add<LOG2CAP>(t_keys, t_cnt, t_min, &s_nd, w.w, opos);
                        for (uint32_t i = 3; ok && i < lcnt; i += 4) {
                            const uint4 x = a[1 + (i - 3) / 4];
                            ok = table_add<LOG2CAP>(t_keys, t_cnt, t_min, &s_nd, x.x, opos);
                            if (ok && i + 1 < lcnt) ok = table_add<LOG2CAP>(t_keys, t_cnt, t_min, &s_nd, x.y, opos);
                            if (ok && i + 2 < lcnt) ok = table_add<LOG2CAP>(t_keys, t_cnt, t_min, &s_nd, x.z, opos);
                            if (ok && i + 3 < lcnt) ok = table_add<LOG2CAP>(t_keys, t_cnt, t_min, &s_nd, x.w, opos);
                        }
                    }
                }
            }
            if (!ok) s_ovf = 1;
            // long lists: the whole wave walks them, coalesced
            unsigned long long m = __ballot(coop);
            while (m) {
                const int src = __ffsll((long long)m) - 1;
                m &= m - 1;
                const uint32_t off = __shfl(oval, src, 64);
                const uint32_t n = __shfl(lcnt, src, 64);
                const uint32_t pos = __shfl(opos, src, 64);
                const uint32_t *ids = p.arena + (uint64_t)off * 4 + 1;
                bool ok2 = true;
                for (uint32_t i = lane; ok2 && i < n; i += 64) ok2 = table_add<LOG2CAP>(t_keys, t_cnt, t_min, &s_nd, ids[i], pos);
                if (!ok2) s_ovf = 1;
                if (*(volatile uint32_t *)&s_nd > CAP_LIMIT) break;
            }
            __syncthreads();
            overflow = (*(volatile uint32_t *)&s_ovf != 0u) || (*(volatile uint32_t *)&s_nd > CAP_LIMIT);
        }

        if (overflow) {
            // hand the query to the global-memory tier
            if (lane == 0) {
                c_ovf++;
                uint32_t slot = atomicAdd(p.ovf_count, 1u);
                if (slot < p.ovf_cap) p.ovf_list[slot] = q;
                else atomicOr(p.status, (uint32_t)ST_OVF_LIST_FULL);
                p.q_cnt[q] = 0;
                p.q_start[q] = 0;
            }
            __syncthreads();
            continue;
        }

        // ---- compaction: ballot + prefix popcount -> dense hit list in the pool
        const uint32_t total = *(volatile uint32_t *)&s_nd;
        uint64_t base = 0;
        bool have = true;
        if (total > 0) {
            if (total > chunk_left) {
                const uint32_t need = total > POOL_CHUNK ? total : POOL_CHUNK;
                unsigned long long b = 0;
                if (lane == 0) b = atomicAdd(p.pool_cursor, (unsigned long long)need);
                b = __shfl(b, 0, 64);
                if (b + need > p.pool_cap) {
                    if (lane == 0) atomicOr(p.status, (uint32_t)ST_POOL_FULL);
                    have = false;
                } else {
                    chunk_base = b;
                    chunk_left = need;
                }
            }
            if (have) {
                base = chunk_base;
                chunk_base += total;
                chunk_left -= total;
                uint32_t running = 0;
                for (uint32_t i0 = 0; i0 < (uint32_t)CAP; i0 += 64) {
                    const uint32_t k = t_keys[i0 + lane];
                    const bool has = k != KH_EMPTY_PID;
                    const unsigned long long bm = __ballot(has);
                    if (has) {
                        const uint32_t idx = running + (uint32_t)__popcll(bm & ((1ull << lane) - 1ull));
                        p.pool_pid[base + idx] = k;
                        p.pool_km[base + idx] = t_cnt[i0 + lane];
                        p.pool_fp[base + idx] = t_min[i0 + lane];
                    }
                    running += (uint32_t)__popcll(bm);
                }
                if (lane == 0) c_hits += total;
            }
        }
        if (lane == 0) { p.q_cnt[q] = have ? total : 0u; p.q_start[q] = base; }
        __syncthreads();
    }

    // ---- counters: one replica line per 64 waves
    c_lookup = wave_sum(c_lookup); c_probe = wave_sum(c_probe);
    c_found = wave_sum(c_found);   c_post = wave_sum(c_post);
    c_lists = wave_sum(c_lists);   c_lids = wave_sum(c_lids);
    if (lane == 0) {
        unsigned long long *c = p.counters + (size_t)(wave % CTR_REPLICAS) * CTR_N;
        if (c_in) atomicAdd(&c[CTR_IN], c_in);
        if (c_q) atomicAdd(&c[CTR_QUERIES], c_q);
        if (c_lookup) atomicAdd(&c[CTR_LOOKUP], c_lookup);
        if (c_probe) atomicAdd(&c[CTR_PROBE], c_probe);
        if (c_found) atomicAdd(&c[CTR_FOUND], c_found);
        if (c_post) atomicAdd(&c[CTR_POST], c_post);
        if (c_hits) atomicAdd(&c[CTR_HITS], c_hits);
        if (c_ovf) atomicAdd(&c[CTR_OVERFLOW], c_ovf);
        if (c_lists) atomicAdd(&c[CTR_LISTS], c_lists);
        if (c_lids) atomicAdd(&c[CTR_LIST_IDS], c_lids);
    }
}

// ------------------------------------------------------------------------------------
// prep: protein records -> query meta (search.go:286-296; search_protein.go:70-76)
// ------------------------------------------------------------------------------------
__global__ void prep_protein_kernel(const uint8_t *seqs, const uint64_t *offsets, uint32_t n_seqs,
                                    kaamer_query_meta *q, uint32_t *d_nq)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) *d_nq = n_seqs;
    if (i >= n_seqs) return;
    const uint64_t b = offsets[i], e = offsets[i + 1];
    const int64_t len = (int64_t)(e - b);
    int32_t size = (int32_t)(len - KAAMER_KMER_SIZE + 1);       // search.go:290
    if (len > 0 && seqs[e - 1] == '*') size--;                  // search.go:291-293
    kaamer_query_meta m;
    m.src_seq = i;
    m.size_in_kmer = size;
    m.start_position = 1;                                       // search.go:225,303
    m.end_position = (int32_t)len;                              // search.go:294
    m.plus_strand = 1;
    m.aa_len = (uint32_t)len;
    m.aa_off = b;
    m.sa_off = 0;
    m.sa_len = 0;
    q[i] = m;
}

// ------------------------------------------------------------------------------------
// exclusive scan of q_cnt[0..nq) -> hit_off[0..nq], three launches
// ------------------------------------------------------------------------------------
#define SCAN_BLOCK 256
#define SCAN_ITEMS 8
#define SCAN_TILE (SCAN_BLOCK * SCAN_ITEMS)

__device__ __forceinline__ uint64_t block_exclusive_scan(uint64_t v, uint64_t *total)
{
    __shared__ uint64_t s_w[SCAN_BLOCK / 64];
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint64_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint64_t t = __shfl_up(inc, o, 64);
        if ((int)lane >= o) inc += t;
    }
    __syncthreads();
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    uint64_t woff = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < SCAN_BLOCK / 64; i++) {
        if (i < (int)w) woff += s_w[i];
        tot += s_w[i];
    }
    *total = tot;
    return woff + inc - v;
}

__global__ __launch_bounds__(SCAN_BLOCK) void scan_block_sums_kernel(const uint32_t *cnt, const uint32_t *d_nq,
                                                                    uint64_t *bsum)
{
    const uint32_t nq = *d_nq;
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE;
    if (base > nq) { if (threadIdx.x == 0) bsum[blockIdx.x] = 0; return; }
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        const uint64_t idx = base + (uint64_t)threadIdx.x * SCAN_ITEMS + i;
        if (idx < nq) s += cnt[idx];
    }
    uint64_t tot;
    block_exclusive_scan(s, &tot);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

__global__ __launch_bounds__(SCAN_BLOCK) void scan_top_kernel(uint64_t *bsum, uint32_t n_blocks)
{
    uint64_t carry = 0;
    for (uint32_t b0 = 0; b0 < n_blocks; b0 += SCAN_BLOCK) {
        const uint32_t i = b0 + threadIdx.x;
        const uint64_t v = i < n_blocks ? bsum[i] : 0;
        uint64_t tot;
        const uint64_t ex = block_exclusive_scan(v, &tot);
        if (i < n_blocks) bsum[i] = carry + ex;
        carry += tot;
        __syncthreads();
    }
}

__global__ __launch_bounds__(SCAN_BLOCK) void scan_apply_kernel(const uint32_t *cnt, const uint32_t *d_nq,
                                                               const uint64_t *bsum, uint64_t *hit_off)
{
    const uint32_t nq = *d_nq;
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE;
    if (base > nq) return;
    uint32_t v[SCAN_ITEMS];
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        const uint64_t idx = base + (uint64_t)threadIdx.x * SCAN_ITEMS + i;
        v[i] = idx < nq ? cnt[idx] : 0u;
        s += v[i];
    }
    uint64_t tot;
    uint64_t ex = block_exclusive_scan(s, &tot) + bsum[blockIdx.x];
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        const uint64_t idx = base + (uint64_t)threadIdx.x * SCAN_ITEMS + i;
        if (idx <= nq) hit_off[idx] = ex;  // index nq receives the grand total
        ex += v[i];
    }
}

// pool -> CSR: one thread per output hit, owner query found by binary search
__global__ void gather_hits_kernel(const uint32_t *d_nq, const uint64_t *hit_off, const uint64_t *q_start,
                                   const uint32_t *pool_pid, const uint32_t *pool_km, const uint32_t *pool_fp,
                                   uint32_t *out_pid, uint32_t *out_km, uint32_t *out_fp, uint64_t out_cap,
                                   uint32_t *status)
{
    const uint32_t nq = *d_nq;
    const uint64_t n_hits = hit_off[nq];
    if (n_hits > out_cap) { if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(status, (uint32_t)ST_POOL_FULL); return; }
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_hits; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t lo = 0, hi = nq;  // last q with hit_off[q] <= i
        while (hi - lo > 1) {
            const uint32_t mid = lo + (hi - lo) / 2;
            if (hit_off[mid] <= i) lo = mid; else hi = mid;
        }
        const uint64_t src = q_start[lo] + (i - hit_off[lo]);
        out_pid[i] = pool_pid[src];
        out_km[i] = pool_km[src];
        out_fp[i] = pool_fp[src];
    }
}

__global__ void finalize_counters_kernel(const unsigned long long *replicas, kaamer_counters *out,
                                         const uint32_t *ovf_count, uint32_t *status)
{
    if (threadIdx.x < CTR_N) {
        unsigned long long s = 0;
        for (int r = 0; r < CTR_REPLICAS; r++) s += replicas[(size_t)r * CTR_N + threadIdx.x];
        ((unsigned long long *)out)[threadIdx.x] = s;
    }
    if (threadIdx.x == 0 && *ovf_count != 0) atomicOr(status, (uint32_t)ST_OVERFLOW_UNSERVED);
}

// ------------------------------------------------------------------------------------
// workspace
// ------------------------------------------------------------------------------------
struct kaamer_workspace {
    int device;
    kaamer_workspace_opts opts;
    uint32_t q_cap;
    uint64_t hit_cap, pool_cap;
    uint32_t lds_log2;
    int search_grid;
    // device buffers
    kaamer_query_meta *d_q;
    uint32_t *d_nq;
    uint64_t *d_q_start;
    uint32_t *d_q_cnt;
    uint32_t *d_pool_pid, *d_pool_km, *d_pool_fp;
    unsigned long long *d_pool_cursor;
    uint32_t *d_ovf_list, *d_ovf_count;
    uint32_t ovf_cap;
    unsigned long long *d_counter_replicas;
    kaamer_counters *d_counters;
    uint32_t *d_status;
    uint64_t *d_bsum;
    uint32_t n_scan_blocks;
    uint64_t *d_hit_off;
    uint32_t *d_hit_pid, *d_hit_km, *d_hit_fp;
    std::vector<hipEvent_t> *ev;  // 4 events per timed call: total0, search0, search1, total1
    uint32_t n_timed;
    bool timed;
};

template <class T> static int dev_alloc(T **p, size_t n)
{
    *p = nullptr;
    hipError_t e = hipMalloc((void **)p, (n ? n : 1) * sizeof(T));
    if (e != hipSuccess) return kaamer_fail(KAAMER_E_NOMEM, "hipMalloc(%zu bytes): %s", n * sizeof(T), hipGetErrorString(e));
    return KAAMER_OK;
}

template <int L> static void launch_search(const SearchParams &p, int grid, hipStream_t s)
{
    hipLaunchKernelGGL(kmer_search_kernel<L>, dim3(grid), dim3(64), 0, s, p);
}

template <int L> static int search_occupancy(int *blocks_per_cu)
{
    HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, kmer_search_kernel<L>, 64, 0));
    return KAAMER_OK;
}

extern "C" {

const char *kaamer_last_error(void) { return g_err; }
int kaamer_abi_version(void) { return KAAMER_ABI_VERSION; }

int kaamer_index_open_image(const kaamer_image *img, int device, kaamer_index **out)
{
    if (!img || !out) return kaamer_fail(KAAMER_E_ARG, "index_open_image: bad argument");
    *out = nullptr;
    HIPCHK(hipSetDevice(device));
    kaamer_index *ix = new (std::nothrow) kaamer_index();
    if (!ix) return kaamer_fail(KAAMER_E_NOMEM, "index alloc");
    ix->device = device;
    ix->hdr = img->hdr;
    ix->d_buckets = nullptr;
    ix->d_arena = nullptr;
    int rc = dev_alloc(&ix->d_buckets, (size_t)img->hdr.n_buckets);
    if (!rc) rc = dev_alloc(&ix->d_arena, (size_t)(img->hdr.arena_words < 4 ? 4 : img->hdr.arena_words));
    if (rc) { kaamer_index_close(ix); return rc; }
    hipError_t e = hipMemcpy(ix->d_buckets, img->buckets, (size_t)img->hdr.n_buckets * sizeof(kh_bucket), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(ix->d_arena, img->arena, (size_t)img->hdr.arena_words * 4, hipMemcpyHostToDevice);
    if (e != hipSuccess) { kaamer_index_close(ix); return kaamer_fail(KAAMER_E_HIP, "index upload: %s", hipGetErrorString(e)); }
    *out = ix;
    return KAAMER_OK;
}

int kaamer_index_open(const char *path, int device, kaamer_index **out)
{
    kaamer_image *img = nullptr;
    int rc = kaamer_image_load(path, &img);
    if (rc) return rc;
    rc = kaamer_index_open_image(img, device, out);
    kaamer_image_free(img);
    return rc;
}

void kaamer_index_close(kaamer_index *ix)
{
    if (!ix) return;
    (void)hipSetDevice(ix->device);
    if (ix->d_buckets) (void)hipFree(ix->d_buckets);
    if (ix->d_arena) (void)hipFree(ix->d_arena);
    delete ix;
}

int kaamer_index_get_stats(const kaamer_index *ix, kaamer_image_stats *out)
{
    if (!ix || !out) return kaamer_fail(KAAMER_E_ARG, "index_get_stats: bad argument");
    kaamer_stats_from_header(&ix->hdr, out);
    return KAAMER_OK;
}

void kaamer_workspace_free(kaamer_workspace *ws)
{
    if (!ws) return;
    (void)hipSetDevice(ws->device);
    void *bufs[] = { ws->d_q, ws->d_nq, ws->d_q_start, ws->d_q_cnt, ws->d_pool_pid, ws->d_pool_km, ws->d_pool_fp,
                     ws->d_pool_cursor, ws->d_ovf_list, ws->d_ovf_count, ws->d_counter_replicas, ws->d_counters,
                     ws->d_status, ws->d_bsum, ws->d_hit_off, ws->d_hit_pid, ws->d_hit_km, ws->d_hit_fp };
    for (void *b : bufs) if (b) (void)hipFree(b);
    if (ws->ev) {
        for (hipEvent_t e : *ws->ev) (void)hipEventDestroy(e);
        delete ws->ev;
    }
    delete ws;
}

int kaamer_workspace_create(kaamer_index *ix, const kaamer_workspace_opts *opts, kaamer_workspace **out)
{
    if (!ix || !opts || !out) return kaamer_fail(KAAMER_E_ARG, "workspace_create: bad argument");
    *out = nullptr;
    HIPCHK(hipSetDevice(ix->device));
    kaamer_workspace *ws = new (std::nothrow) kaamer_workspace();
    if (!ws) return kaamer_fail(KAAMER_E_NOMEM, "workspace alloc");
    memset(ws, 0, sizeof *ws);
    ws->device = ix->device;
    ws->opts = *opts;
    ws->q_cap = opts->max_queries ? opts->max_queries : opts->max_seqs;
    if (ws->q_cap < 1) ws->q_cap = 1;
    ws->hit_cap = opts->max_hits ? opts->max_hits : (uint64_t)ws->q_cap * 64 + (1u << 20);
    uint32_t slots = opts->lds_slots ? opts->lds_slots : 512;
    uint32_t l2 = 6;
    while ((1u << l2) < slots && l2 < 12) l2++;
    if ((1u << l2) != slots) { delete ws; return kaamer_fail(KAAMER_E_ARG, "lds_slots must be a power of two in [64,4096]"); }
    ws->lds_log2 = l2;
    int per_cu = 0, rc = KAAMER_OK;
    switch (l2) {
    case 6: rc = search_occupancy<6>(&per_cu); break;
    case 7: rc = search_occupancy<7>(&per_cu); break;
    case 8: rc = search_occupancy<8>(&per_cu); break;
    case 9: rc = search_occupancy<9>(&per_cu); break;
    case 10: rc = search_occupancy<10>(&per_cu); break;
    case 11: rc = search_occupancy<11>(&per_cu); break;
    default: rc = search_occupancy<12>(&per_cu); break;
    }
    if (rc) { delete ws; return rc; }
    hipDeviceProp_t prop;
    hipError_t pe = hipGetDeviceProperties(&prop, ix->device);
    if (pe != hipSuccess) { delete ws; return kaamer_fail(KAAMER_E_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(pe)); }
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 32) per_cu = 32;
    ws->search_grid = prop.multiProcessorCount * per_cu;
    // every resident wave may hold one partly used chunk
    ws->pool_cap = ws->hit_cap + (uint64_t)ws->search_grid * POOL_CHUNK;
    ws->ovf_cap = ws->q_cap;
    ws->n_scan_blocks = (uint32_t)(((uint64_t)ws->q_cap + 1 + SCAN_TILE - 1) / SCAN_TILE);
    rc = dev_alloc(&ws->d_q, ws->q_cap);
    if (!rc) rc = dev_alloc(&ws->d_nq, 1);
    if (!rc) rc = dev_alloc(&ws->d_q_start, ws->q_cap);
    if (!rc) rc = dev_alloc(&ws->d_q_cnt, ws->q_cap);
    if (!rc) rc = dev_alloc(&ws->d_pool_pid, ws->pool_cap);
    if (!rc) rc = dev_alloc(&ws->d_pool_km, ws->pool_cap);
    if (!rc) rc = dev_alloc(&ws->d_pool_fp, ws->pool_cap);
    if (!rc) rc = dev_alloc(&ws->d_pool_cursor, 1);
    if (!rc) rc = dev_alloc(&ws->d_ovf_list, ws->ovf_cap);
    if (!rc) rc = dev_alloc(&ws->d_ovf_count, 1);
    if (!rc) rc = dev_alloc(&ws->d_counter_replicas, (size_t)CTR_REPLICAS * CTR_N);
    if (!rc) rc = dev_alloc(&ws->d_counters, 1);
    if (!rc) rc = dev_alloc(&ws->d_status, 1);
    if (!rc) rc = dev_alloc(&ws->d_bsum, ws->n_scan_blocks);
    if (!rc) rc = dev_alloc(&ws->d_hit_off, (size_t)ws->q_cap + 1);
    if (!rc) rc = dev_alloc(&ws->d_hit_pid, ws->hit_cap);
    if (!rc) rc = dev_alloc(&ws->d_hit_km, ws->hit_cap);
    if (!rc) rc = dev_alloc(&ws->d_hit_fp, ws->hit_cap);
    if (rc) { kaamer_workspace_free(ws); return rc; }
    ws->ev = new (std::nothrow) std::vector<hipEvent_t>();
    if (!ws->ev) { kaamer_workspace_free(ws); return kaamer_fail(KAAMER_E_NOMEM, "event ring"); }
    *out = ws;
    return KAAMER_OK;
}

int kaamer_search_device(kaamer_index *ix, kaamer_workspace *ws, const uint8_t *d_seqs, const uint64_t *d_offsets,
                         uint32_t n_seqs, uint64_t seq_bytes, int32_t seq_type, void *stream,
                         kaamer_device_result *out)
{
    if (!ix || !ws || !out || (n_seqs && (!d_seqs || !d_offsets))) return kaamer_fail(KAAMER_E_ARG, "search_device: bad argument");
    if (ix->device != ws->device) return kaamer_fail(KAAMER_E_ARG, "workspace belongs to another device");
    if (seq_type != KAAMER_PROTEIN) return kaamer_fail(KAAMER_E_ARG, "search_device: sequence type %d not supported yet", seq_type);
    if (n_seqs > ws->q_cap) return kaamer_fail(KAAMER_E_CAPACITY, "batch of %u sequences exceeds workspace max_seqs %u", n_seqs, ws->q_cap);
    (void)seq_bytes;
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipSetDevice(ix->device));
    if (ws->n_timed >= MAX_TIMED_CALLS) ws->n_timed = 0;
    while (ws->ev->size() < (size_t)(ws->n_timed + 1) * 4) {
        hipEvent_t e;
        HIPCHK(hipEventCreate(&e));
        ws->ev->push_back(e);
    }
    hipEvent_t *ev = ws->ev->data() + (size_t)ws->n_timed * 4;
    HIPCHK(hipEventRecord(ev[0], s));
    HIPCHK(hipMemsetAsync(ws->d_pool_cursor, 0, sizeof(unsigned long long), s));
    HIPCHK(hipMemsetAsync(ws->d_ovf_count, 0, sizeof(uint32_t), s));
    HIPCHK(hipMemsetAsync(ws->d_counter_replicas, 0, sizeof(unsigned long long) * CTR_REPLICAS * CTR_N, s));
    HIPCHK(hipMemsetAsync(ws->d_status, 0, sizeof(uint32_t), s));

    const int pb = 256;
    hipLaunchKernelGGL(prep_protein_kernel, dim3((n_seqs + pb - 1) / pb > 0 ? (n_seqs + pb - 1) / pb : 1), dim3(pb), 0, s,
                       d_seqs, d_offsets, n_seqs, ws->d_q, ws->d_nq);

    SearchParams p;
    p.table = reinterpret_cast<const uint4 *>(ix->d_buckets);
    p.n_buckets = ix->hdr.n_buckets;
    p.n_shards = ix->hdr.n_shards;
    p.shard = ix->hdr.shard;
    p.arena = ix->d_arena;
    p.residues = d_seqs;
    p.q = ws->d_q;
    p.d_nq = ws->d_nq;
    p.min_size = 7;  // search_protein.go:74-76
    p.q_start = ws->d_q_start;
    p.q_cnt = ws->d_q_cnt;
    p.pool_pid = ws->d_pool_pid;
    p.pool_km = ws->d_pool_km;
    p.pool_fp = ws->d_pool_fp;
    p.pool_cap = ws->pool_cap;
    p.pool_cursor = ws->d_pool_cursor;
    p.ovf_list = ws->d_ovf_list;
    p.ovf_count = ws->d_ovf_count;
    p.ovf_cap = ws->ovf_cap;
    p.counters = ws->d_counter_replicas;
    p.status = ws->d_status;

    int grid = ws->search_grid;
    if ((uint32_t)grid > n_seqs) grid = n_seqs > 0 ? (int)n_seqs : 1;
    HIPCHK(hipEventRecord(ev[1], s));
    switch (ws->lds_log2) {
    case 6: launch_search<6>(p, grid, s); break;
    case 7: launch_search<7>(p, grid, s); break;
    case 8: launch_search<8>(p, grid, s); break;
    case 9: launch_search<9>(p, grid, s); break;
    case 10: launch_search<10>(p, grid, s); break;
    case 11: launch_search<11>(p, grid, s); break;
    default: launch_search<12>(p, grid, s); break;
    }
    HIPCHK(hipEventRecord(ev[2], s));

    const uint32_t nsb = (uint32_t)(((uint64_t)n_seqs + 1 + SCAN_TILE - 1) / SCAN_TILE);
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3(nsb), dim3(SCAN_BLOCK), 0, s, ws->d_q_cnt, ws->d_nq, ws->d_bsum);
    hipLaunchKernelGGL(scan_top_kernel, dim3(1), dim3(SCAN_BLOCK), 0, s, ws->d_bsum, nsb);
    hipLaunchKernelGGL(scan_apply_kernel, dim3(nsb), dim3(SCAN_BLOCK), 0, s, ws->d_q_cnt, ws->d_nq, ws->d_bsum, ws->d_hit_off);
    hipLaunchKernelGGL(gather_hits_kernel, dim3(1024), dim3(256), 0, s, ws->d_nq, ws->d_hit_off, ws->d_q_start,
                       ws->d_pool_pid, ws->d_pool_km, ws->d_pool_fp, ws->d_hit_pid, ws->d_hit_km, ws->d_hit_fp,
                       ws->hit_cap, ws->d_status);
    hipLaunchKernelGGL(finalize_counters_kernel, dim3(1), dim3(64), 0, s, ws->d_counter_replicas, ws->d_counters,
                       ws->d_ovf_count, ws->d_status);
    HIPCHK(hipEventRecord(ev[3], s));
    HIPCHK(hipGetLastError());
    ws->n_timed++;
    ws->timed = true;

    out->n_queries_cap = ws->q_cap;
    out->d_n_queries = ws->d_nq;
    out->d_q = ws->d_q;
    out->d_hit_off = ws->d_hit_off;
    out->d_hit_pid = ws->d_hit_pid;
    out->d_hit_kmatch = ws->d_hit_km;
    out->d_hit_first_pos = ws->d_hit_fp;
    out->d_orf_aa = nullptr;
    out->d_starts_alt = nullptr;
    out->d_counters = ws->d_counters;
    return KAAMER_OK;
}

int kaamer_workspace_finish(kaamer_workspace *ws, void *stream, kaamer_counters *out)
{
    if (!ws) return kaamer_fail(KAAMER_E_ARG, "workspace_finish: bad argument");
    HIPCHK(hipSetDevice(ws->device));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    uint32_t status = 0;
    HIPCHK(hipMemcpy(&status, ws->d_status, sizeof status, hipMemcpyDeviceToHost));
    kaamer_counters c;
    HIPCHK(hipMemcpy(&c, ws->d_counters, sizeof c, hipMemcpyDeviceToHost));
    if (out) *out = c;
    if (status & ST_POOL_FULL) return kaamer_fail(KAAMER_E_CAPACITY, "hit pool exhausted: raise workspace max_hits (now %llu)", (unsigned long long)ws->hit_cap);
    if (status & ST_OVF_LIST_FULL) return kaamer_fail(KAAMER_E_CAPACITY, "overflow list exhausted");
    if (status & ST_OVERFLOW_UNSERVED)
        return kaamer_fail(KAAMER_E_CAPACITY, "%llu queries exceeded the on-chip counting table (%u slots): raise workspace lds_slots",
                           (unsigned long long)c.n_overflow, 1u << ws->lds_log2);
    if (status) return kaamer_fail(KAAMER_E_CAPACITY, "device status 0x%x", status);
    return KAAMER_OK;
}

int kaamer_workspace_last_kernel_ms(kaamer_workspace *ws, float *search_ms, float *total_ms)
{
    if (!ws || !ws->timed || ws->n_timed == 0) return kaamer_fail(KAAMER_E_ARG, "no timed batch on this workspace");
    HIPCHK(hipSetDevice(ws->device));
    hipEvent_t *ev = ws->ev->data() + (size_t)(ws->n_timed - 1) * 4;
    if (search_ms) HIPCHK(hipEventElapsedTime(search_ms, ev[1], ev[2]));
    if (total_ms) HIPCHK(hipEventElapsedTime(total_ms, ev[0], ev[3]));
    return KAAMER_OK;
}

int kaamer_workspace_kernel_ms_sum(kaamer_workspace *ws, double *search_ms, double *total_ms, uint32_t *n_calls)
{
    if (!ws) return kaamer_fail(KAAMER_E_ARG, "kernel_ms_sum: bad argument");
    HIPCHK(hipSetDevice(ws->device));
    double a = 0, b = 0;
    for (uint32_t i = 0; i < ws->n_timed; i++) {
        hipEvent_t *ev = ws->ev->data() + (size_t)i * 4;
        float x = 0, y = 0;
        HIPCHK(hipEventElapsedTime(&x, ev[1], ev[2]));
        HIPCHK(hipEventElapsedTime(&y, ev[0], ev[3]));
        a += x;
        b += y;
    }
    if (search_ms) *search_ms = a;
    if (total_ms) *total_ms = b;
    if (n_calls) *n_calls = ws->n_timed;
    return KAAMER_OK;
}

void kaamer_workspace_reset_timers(kaamer_workspace *ws)
{
    if (ws) ws->n_timed = 0;
}

// ------------------------------------------------------------------------------------
// host-buffer form
// ------------------------------------------------------------------------------------
struct batch_out_owner {
    kaamer_batch_out pub;
    std::vector<kaamer_query_meta> q;
    std::vector<uint64_t> hit_off;
    std::vector<uint32_t> pid, km, fp;
};

int kaamer_search_batch(kaamer_index *ix, const kaamer_batch_in *in, kaamer_batch_out **out)
{
    if (!ix || !in || !out || !in->offsets || (in->n_seqs && !in->seqs)) return kaamer_fail(KAAMER_E_ARG, "search_batch: bad argument");
    *out = nullptr;
    if (in->want_positions) return kaamer_fail(KAAMER_E_ARG, "want_positions not supported yet");
    HIPCHK(hipSetDevice(ix->device));
    const uint64_t seq_bytes = in->offsets[in->n_seqs];
    kaamer_workspace_opts o;
    memset(&o, 0, sizeof o);
    o.max_seq_bytes = seq_bytes;
    o.max_seqs = in->n_seqs ? in->n_seqs : 1;
    // distinct hits per query <= min(lookups, proteins)
    uint64_t bound = 0;
    for (uint32_t i = 0; i < in->n_seqs; i++) bound += in->offsets[i + 1] - in->offsets[i];
    o.max_hits = bound * 8 + 4096;
    kaamer_workspace *ws = nullptr;
    int rc = kaamer_workspace_create(ix, &o, &ws);
    if (rc) return rc;
    uint8_t *d_seqs = nullptr;
    uint64_t *d_off = nullptr;
    batch_out_owner *bo = nullptr;
    hipStream_t s = nullptr;
    kaamer_device_result dr;
    kaamer_counters c;
    uint32_t nq = 0;
    uint64_t n_hits = 0;
    hipError_t e;
    rc = dev_alloc(&d_seqs, (size_t)seq_bytes + 16);
    if (!rc) rc = dev_alloc(&d_off, (size_t)in->n_seqs + 1);
    if (rc) goto done;
    e = hipMemcpyAsync(d_seqs, in->seqs, (size_t)seq_bytes, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_off, in->offsets, ((size_t)in->n_seqs + 1) * 8, hipMemcpyHostToDevice, s);
    if (e != hipSuccess) { rc = kaamer_fail(KAAMER_E_HIP, "H2D: %s", hipGetErrorString(e)); goto done; }
    rc = kaamer_search_device(ix, ws, d_seqs, d_off, in->n_seqs, seq_bytes, in->seq_type, s, &dr);
    if (rc) goto done;
    rc = kaamer_workspace_finish(ws, s, &c);
    if (rc) goto done;
    bo = new (std::nothrow) batch_out_owner();
    if (!bo) { rc = kaamer_fail(KAAMER_E_NOMEM, "batch_out"); goto done; }
    e = hipMemcpy(&nq, dr.d_n_queries, 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) { bo->q.resize(nq); bo->hit_off.resize((size_t)nq + 1); }
    if (e == hipSuccess && nq) e = hipMemcpy(bo->q.data(), dr.d_q, (size_t)nq * sizeof(kaamer_query_meta), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(bo->hit_off.data(), dr.d_hit_off, ((size_t)nq + 1) * 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess) {
        n_hits = bo->hit_off[nq];
        bo->pid.resize(n_hits); bo->km.resize(n_hits); bo->fp.resize(n_hits);
        if (n_hits) {
            e = hipMemcpy(bo->pid.data(), dr.d_hit_pid, n_hits * 4, hipMemcpyDeviceToHost);
            if (e == hipSuccess) e = hipMemcpy(bo->km.data(), dr.d_hit_kmatch, n_hits * 4, hipMemcpyDeviceToHost);
            if (e == hipSuccess) e = hipMemcpy(bo->fp.data(), dr.d_hit_first_pos, n_hits * 4, hipMemcpyDeviceToHost);
        }
    }
    if (e != hipSuccess) { rc = kaamer_fail(KAAMER_E_HIP, "D2H: %s", hipGetErrorString(e)); goto done; }
    memset(&bo->pub, 0, sizeof bo->pub);
    bo->pub.n_queries = nq;
    bo->pub.q = bo->q.data();
    bo->pub.hit_off = bo->hit_off.data();
    bo->pub.hit_pid = bo->pid.data();
    bo->pub.hit_kmatch = bo->km.data();
    bo->pub.hit_first_pos = bo->fp.data();
    bo->pub.counters = c;
    *out = &bo->pub;
    bo = nullptr;
done:
    delete bo;
    if (d_seqs) (void)hipFree(d_seqs);
    if (d_off) (void)hipFree(d_off);
    kaamer_workspace_free(ws);
    return rc;
}

void kaamer_batch_free(kaamer_batch_out *out)
{
    if (!out) return;
    delete reinterpret_cast<batch_out_owner *>(out);  // pub is the first member
}

int64_t kaamer_filter_results(const uint32_t *kmatch_sorted, int64_t n_hits, int32_t size_in_kmer, double min_k_ratio,
                              int64_t min_k_match, int64_t max_results)
{
    // search.go:189-220
    int64_t last_good = n_hits - 1;
    for (int64_t i = 0; i < n_hits; i++) {
        const int64_t km = (int64_t)kmatch_sorted[i];
        if (((double)km / (double)size_in_kmer) < min_k_ratio || km < min_k_match) {
            if (last_good == n_hits - 1) last_good = i - 1;
        }
    }
    if (last_good >= max_results) last_good = max_results - 1;
    return last_good < 0 ? 0 : last_good + 1;
}

void kaamer_sort_hits(const uint32_t *pid, const uint32_t *kmatch, int64_t n_hits, uint32_t *order)
{
    std::vector<uint32_t> idx((size_t)n_hits);
    for (int64_t i = 0; i < n_hits; i++) idx[(size_t)i] = (uint32_t)i;
    std::sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) {
        if (kmatch[a] != kmatch[b]) return kmatch[a] > kmatch[b];
        return pid[a] < pid[b];
    });
    for (int64_t i = 0; i < n_hits; i++) order[i] = idx[(size_t)i];
}

}  // extern "C"
