// builder_device.hip — the builder of builder.cpp on the device: the same image, byte for byte.
//
// Offline side of the reference restated there:
//   emit     : pkg/makedb/inputFASTA.go:245-248, inputTSV.go:236-239
//   collapse : pkg/indexdb/indexdb.go:68-132 + pkg/kvstore/kv_store.go:284-305  => key -> set<proteinId>
//   sharing  : pkg/kvstore/kcomb_store.go:42-85 (identical sets stored once)
//
// builder.cpp walks the distinct keys in ascending order and, per key, (1) shares or appends its postings
// set and (2) puts the key into the first free slot along its probe sequence.  Both steps are stated here
// without the walk:
//   (1) a set's place in the arena is decided by the SMALLEST key that holds it (its representative): sets
//       are matched through a table on a 64-bit content hash, every match is verified word by word (a
//       mismatch = hash collision -> the pass is repeated under another seed), and the representatives'
//       padded sizes are prefix-summed in key order;
//   (2) a bucket keeps the 8 earliest arrivals (arrival order = key order) among its own keys and the ones
//       the previous bucket could not hold, and hands the rest on.  How MANY it hands on is
//       carry[b] = max(0, carry[b-1] + keys_at_home[b] - 8): a scan under  f o g  of the maps
//       c -> max(a, c + s).  WHICH ones is a merge along each run of buckets with a non-zero carry; runs are
//       independent, one thread walks one run.
// Sorting, scans and selections are rocPRIM (offline path); the kernels around them are below.
#include "kaamer_internal.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

namespace {

#define BD_HIP(expr)                                                                               \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return kaamer_fail(KAAMER_E_HIP, "device build: %s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

struct DevMem {  // freed on scope exit unless taken
    void *p = nullptr;
    DevMem() = default;
    DevMem(const DevMem &) = delete;
    DevMem &operator=(const DevMem &) = delete;
    ~DevMem() { release(); }
    hipError_t alloc(size_t bytes) { release(); return hipMalloc(&p, bytes ? bytes : 16); }
    void release() { if (p) { (void)hipFree(p); p = nullptr; } }
    void *take() { void *q = p; p = nullptr; return q; }
    template <class T> T *as() const { return (T *)p; }
};

struct Trace {
    bool on = getenv("KAAMER_BUILD_TRACE") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(const char *what, uint64_t n)
    {
        if (!on) return;
        (void)hipDeviceSynchronize();
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[kaamer device build] %-24s %8.3f s  (%llu)\n", what, std::chrono::duration<double>(now - t).count(), (unsigned long long)n);
        t = now;
    }
};

// counters the kernels add to
struct BuildStats {
    unsigned long long n_windows;   // pairs emitted (before unique)
    unsigned long long n_inline, n_lists, max_list, max_pid;
    unsigned long long ub_units;    // 16-byte units of all postings sets, unshared (builder.cpp's ub_words / 4)
    unsigned long long n_displaced, hops;
    unsigned int collision, bad;
};

__device__ inline uint64_t bd_mix64(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}

__device__ inline unsigned long long wave_sum(unsigned long long v)
{
    for (int d = 32; d; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__device__ inline unsigned long long wave_max(unsigned long long v)
{
    for (int d = 32; d; d >>= 1) { const unsigned long long o = __shfl_xor(v, d, 64); v = o > v ? o : v; }
    return v;
}

// ---- emit: every 7-mer window of this shard -> key<<32 | id, in any order -------------------------------
#define BD_TILE 4096u      // residues per workgroup
#define BD_THREADS 256u
#define BD_PER_THREAD (BD_TILE / BD_THREADS)

// largest p in [lo, hi) with off[p] <= pos   (off[lo] <= pos)
__device__ inline uint32_t protein_of(const uint64_t *off, uint32_t lo, uint32_t hi, uint64_t pos)
{
    while (hi - lo > 1) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (off[mid] <= pos) lo = mid; else hi = mid;
    }
    return lo;
}

template <bool EMIT>
__global__ __launch_bounds__(BD_THREADS) void bd_windows_kernel(const uint8_t *__restrict__ seqs, const uint64_t *__restrict__ off,
                                                                 const uint32_t *__restrict__ ids, uint32_t n_prot, uint64_t origin,
                                                                 uint64_t total_res, uint32_t shard, uint32_t n_shards,
                                                                 uint64_t *__restrict__ pairs, uint64_t cap, BuildStats *st)
{
    // A thread owns BD_PER_THREAD CONSECUTIVE positions: their residues are 22 bytes read once (a wave reads 1 KB in one
    // piece) and coded once, the windows slide over them, and the protein of a position is the previous position's or
    // one of the next (one binary search per thread, inside the tile's protein range).  Positions are absolute (as in
    // off[]); seqs is indexed with pos - origin and has 16 bytes of slack behind the last residue.
    __shared__ uint8_t s_lut[256];
    for (uint32_t i = threadIdx.x; i < 256; i += BD_THREADS) s_lut[i] = (uint8_t)kh_residue_code((uint8_t)i);
    __syncthreads();
    const uint64_t base = origin + (uint64_t)blockIdx.x * BD_TILE;
    const uint64_t end = origin + total_res;
    const uint64_t tile_last = (base + BD_TILE < end ? base + BD_TILE : end) - 1;
    const uint32_t p_lo = protein_of(off, 0, n_prot, base);
    const uint32_t p_hi = protein_of(off, p_lo, n_prot, tile_last) + 1;
    const uint64_t pos0 = base + (uint64_t)threadIdx.x * BD_PER_THREAD;
    uint32_t keys[BD_PER_THREAD], pid[BD_PER_THREAD];
    uint32_t okmask = 0, mine = 0;
#pragma unroll
    for (uint32_t k = 0; k < BD_PER_THREAD; k++) { keys[k] = 0; pid[k] = 0; }
    if (pos0 + KAAMER_KMER_SIZE <= end) {
        uint8_t c[BD_PER_THREAD + 8];
        {
            uint32_t w[(BD_PER_THREAD + 8) / 4];
            __builtin_memcpy(w, seqs + (pos0 - origin), sizeof w);
#pragma unroll
            for (uint32_t i = 0; i < BD_PER_THREAD + 8; i++) c[i] = s_lut[(w[i >> 2] >> (8u * (i & 3u))) & 255u];
        }
        uint32_t p = protein_of(off, p_lo, p_hi, pos0);
        uint64_t p_end = off[p + 1];
        uint32_t id = ids ? ids[p] : p;                                    // inputTSV.go:141-142
#pragma unroll
        for (uint32_t k = 0; k < BD_PER_THREAD; k++) {
            const uint64_t pos = pos0 + k;
            while (pos >= p_end && p + 1 < n_prot) { p++; p_end = off[p + 1]; id = ids ? ids[p] : p; }   // (empty proteins too)
            bool ok = false;
            if (pos + KAAMER_KMER_SIZE <= p_end && pos < end) {              // inputFASTA.go:228,245
                const uint32_t key = kh_key_from_codes(c[k], c[k + 1], c[k + 2], c[k + 3], c[k + 4], c[k + 5], c[k + 6]);
                ok = n_shards <= 1 || kh_shard_of(key, n_shards) == shard;
                keys[k] = key;
                pid[k] = id;
            }
            okmask |= (uint32_t)ok << k;
            mine += ok;
        }
    }
    // one reservation per wave
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t incl = mine;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d, 64); if ((int)lane >= d) incl += o; }
    const uint32_t wave_total = __shfl(incl, 63, 64);
    unsigned long long wbase = 0;
    if (lane == 63 && wave_total) wbase = atomicAdd(&st->n_windows, (unsigned long long)wave_total);
    wbase = __shfl(wbase, 63, 64);
    if (!EMIT) return;
    uint64_t at = wbase + (incl - mine);
#pragma unroll
    for (uint32_t k = 0; k < BD_PER_THREAD; k++)
        if ((okmask >> k) & 1u) {
            if (at < cap) pairs[at] = ((uint64_t)keys[k] << 32) | pid[k];
            else st->bad = 1;
            at++;
        }
}

// ---- keys -------------------------------------------------------------------------------------------------
struct KeyHead {  // flag of sorted unique pair i: first pair of its key
    const uint64_t *pairs;
    __device__ bool operator()(uint32_t i) const { return i == 0 || (pairs[i] >> 32) != (pairs[i - 1] >> 32); }
};

struct KeyArrays {
    const uint64_t *pairs;   // sorted, unique
    const uint32_t *kstart;  // [n_keys + 1]
    uint32_t n_keys;
    uint64_t *hv;            // [n_keys] content hash of a list key's set (0: the key stores its id inline)
    uint32_t *hk;            // [n_keys] key index (the value of the sort by hash)
    const uint64_t *hv_s;    // the two arrays sorted by hash (stable: within a hash, ascending key index)
    const uint32_t *hk_s;
    const uint32_t *run_start;  // [n_keys] sorted position of the first element with the same hash
    uint32_t *rep_of;        // [n_keys] representative of a list key's set = the smallest key index with its hash
    uint32_t *units;         // [n_keys] 16-byte units a key ADDS to the arena
    uint32_t *uoff;          // [n_keys] exclusive prefix of units, from 1
    uint32_t *arena;
    uint64_t *pkeys;         // [n_keys] home bucket << 32 | key
    uint32_t *pvals;         // [n_keys] slot value
    uint64_t n_buckets;
    uint32_t n_shards;
    uint64_t seed;
    BuildStats *st;
};

__device__ inline bool key_is_inline(const KeyArrays &a, uint32_t k, uint32_t &s, uint32_t &c)
{
    s = a.kstart[k];
    c = a.kstart[k + 1] - s;
    return c == 1 && (uint32_t)a.pairs[s] < KH_INLINE_BIT;
}

// pass 1: content hash of every postings set.  The representative of a set is the smallest key index that carries its hash:
// the keys are SORTED by hash (a stable radix sort: equal hashes stay in key order) and the head of every run is the
// representative.  (Rounds 2-3 found it with a compare-and-swap into an open-addressed table + atomicMin per key: 93 ms of
// the 0.3 s DB-SP build went into those contended atomics; the sort is ~10 ms.)
__global__ __launch_bounds__(256) void bd_hash_kernel(KeyArrays a)
{
    // grid-stride: the four statistics below are ONE word each, and an atomic per wave on a single word serialises at
    // ~90 atomics/us -- with a thread per key and 2 M waves that was 90 ms of this "hash" kernel (rounds 2-3 blamed the
    // table's atomics); a block now walks many keys and adds its totals once
    unsigned long long n_inline = 0, ub = 0, mx = 0, mpid = 0;
    for (uint64_t k64 = (uint64_t)blockIdx.x * 256u + threadIdx.x; k64 < a.n_keys; k64 += (uint64_t)gridDim.x * 256u) {
        const uint32_t k = (uint32_t)k64;
        uint32_t s, c;
        const bool inl = key_is_inline(a, k, s, c);
        const unsigned long long last = (uint32_t)a.pairs[s + c - 1];
        mpid = last > mpid ? last : mpid;
        uint64_t h = 0;
        if (inl) n_inline++;
        else {
            ub += (1ull + c + 3ull) / 4ull;
            mx = c > mx ? c : mx;
            h = a.seed ^ c;
            // seed 0 (tests only): the hash is the set's size, so unequal sets meet
            for (uint32_t t = 0; a.seed && t < c; t++) h = bd_mix64(h + (uint32_t)a.pairs[s + t] * 0x9E3779B97F4A7C15ull);
            if (!h) h = 1;
        }
        a.hv[k] = h;
        a.hk[k] = k;
    }
    n_inline = wave_sum(n_inline); ub = wave_sum(ub); mx = wave_max(mx); mpid = wave_max(mpid);
    if ((threadIdx.x & 63u) == 0) {
        if (n_inline) atomicAdd(&a.st->n_inline, n_inline);
        if (ub) atomicAdd(&a.st->ub_units, ub);
        atomicMax(&a.st->max_list, mx);
        atomicMax(&a.st->max_pid, mpid);
    }
}

// sorted position i -> i if it is the first of its hash, else 0 (an inclusive max-scan turns that into the run's start)
struct RunHead {
    const uint64_t *hv_s;
    __device__ uint32_t operator()(uint32_t i) const { return (i == 0 || hv_s[i] != hv_s[i - 1]) ? i : 0u; }
};
struct MaxU32 {
    __host__ __device__ uint32_t operator()(uint32_t x, uint32_t y) const { return x > y ? x : y; }
};

__global__ __launch_bounds__(256) void bd_rep_kernel(KeyArrays a)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= a.n_keys || a.hv_s[i] == 0) return;
    a.rep_of[a.hk_s[i]] = a.hk_s[a.run_start[i]];
}

// pass 2: representatives take arena space; everyone else proves it holds the representative's set
__global__ __launch_bounds__(256) void bd_share_kernel(KeyArrays a)
{
    unsigned long long n_lists = 0;
    for (uint64_t k64 = (uint64_t)blockIdx.x * 256u + threadIdx.x; k64 < a.n_keys; k64 += (uint64_t)gridDim.x * 256u) {   // (grid-stride: see bd_hash_kernel)
        const uint32_t k = (uint32_t)k64;
        uint32_t s, c, u = 0;
        if (!key_is_inline(a, k, s, c)) {
            const uint32_t rep = a.rep_of[k];
            if (rep == k) { u = (1u + c + 3u) / 4u; n_lists++; }
            else {
                const uint32_t rs = a.kstart[rep], rc = a.kstart[rep + 1] - rs;
                bool same = rc == c;
                for (uint32_t t = 0; same && t < c; t++) same = (uint32_t)a.pairs[rs + t] == (uint32_t)a.pairs[s + t];
                if (!same) a.st->collision = 1;
            }
        }
        a.units[k] = u;
    }
    n_lists = wave_sum(n_lists);
    if ((threadIdx.x & 63u) == 0 && n_lists) atomicAdd(&a.st->n_lists, n_lists);
}

// pass 3: postings into the arena, slot values, and the (home bucket, key) records of the placement
__global__ __launch_bounds__(256) void bd_lists_kernel(KeyArrays a)
{
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= a.n_keys) return;
    uint32_t s, c, val;
    const uint32_t key = (uint32_t)(a.pairs[a.kstart[k]] >> 32);
    if (key_is_inline(a, k, s, c)) val = KH_INLINE_BIT | (uint32_t)a.pairs[s];
    else {
        const uint32_t rep = a.rep_of[k];
        val = a.uoff[rep];
        if (rep == k) {
            uint32_t *l = a.arena + (uint64_t)val * 4;
            l[0] = c;
            for (uint32_t t = 0; t < c; t++) l[1 + t] = (uint32_t)a.pairs[s + t];
            for (uint32_t t = 1 + c; t < ((1 + c + 3) / 4) * 4; t++) l[t] = KH_EMPTY_PID;
        }
    }
    a.pkeys[k] = (kh_home_bucket(key, a.n_shards, a.n_buckets) << 32) | key;
    a.pvals[k] = val;
}

// ---- placement ------------------------------------------------------------------------------------------------
// bstart[b] = first record of bucket b among the records sorted by (home, key); bstart[n_buckets] = n_keys
__global__ __launch_bounds__(256) void bd_bucket_start_kernel(const uint64_t *__restrict__ pkeys, uint32_t n_keys, uint64_t n_buckets,
                                                              uint32_t *__restrict__ bstart)
{
    const uint64_t b = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (b > n_buckets) return;
    const uint64_t want = b << 32;
    uint32_t lo = 0, hi = n_keys;  // first i with pkeys[i] >= want
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (pkeys[mid] < want) lo = mid + 1; else hi = mid;
    }
    bstart[b] = lo;
}

// the map c -> max(a, c + s); compose(l, r) applies l first
struct CarryMap { long long a, s; };
struct CarryCompose {
    __host__ __device__ CarryMap operator()(const CarryMap &l, const CarryMap &r) const
    {
        const long long t = l.a + r.s;
        return CarryMap{ r.a > t ? r.a : t, l.s + r.s };
    }
};
struct CarryOfBucket {
    const uint32_t *bstart;
    __device__ CarryMap operator()(uint32_t b) const
    {
        return CarryMap{ 0, (long long)(bstart[b + 1] - bstart[b]) - KH_SLOTS_PER_BUCKET };
    }
};

__global__ __launch_bounds__(256) void bd_carry_kernel(const CarryMap *__restrict__ pre, uint64_t n_buckets, uint32_t *__restrict__ carry, BuildStats *st)
{
    const uint64_t b = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    unsigned long long c = 0;
    if (b < n_buckets) {
        // what the last bucket hands to bucket 0: the fixed point of the whole table's map (its slope is negative:
        // the table has more slots than keys)
        const long long c_in = pre[n_buckets - 1].a;
        const long long v = pre[b].a > c_in + pre[b].s ? pre[b].a : c_in + pre[b].s;
        c = (unsigned long long)v;
        carry[b] = (uint32_t)(c > 0xFFFFFFFFull ? 0xFFFFFFFFull : c);
    }
    c = wave_sum(c);
    if ((threadIdx.x & 63u) == 0 && c) atomicAdd(&st->hops, c);
}

// one thread per run: a bucket whose predecessor hands nothing on starts one and walks while buckets overflow
__global__ __launch_bounds__(256) void bd_place_kernel(const uint64_t *__restrict__ pkeys, const uint32_t *__restrict__ pvals,
                                                       const uint32_t *__restrict__ bstart, const uint32_t *__restrict__ carry,
                                                       const uint32_t *__restrict__ ooff, kh_slot *__restrict__ spill,
                                                       kh_bucket *__restrict__ buckets, uint64_t n_buckets, BuildStats *st)
{
    const uint64_t b0 = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    unsigned long long displaced = 0;
    if (b0 < n_buckets && carry[b0 == 0 ? n_buckets - 1 : b0 - 1] == 0) {
        uint64_t cur = b0;
        const kh_slot *in = nullptr;   // what the previous bucket handed on, in key order
        uint32_t n_in = 0;
        for (;;) {
            uint32_t h = bstart[cur];
            const uint32_t h_end = bstart[cur + 1];
            kh_slot *out = spill + ooff[cur];
            uint32_t i_in = 0, n_out = 0, filled = 0;
            kh_slot *dst = buckets[cur].s;
            while (h < h_end || i_in < n_in) {
                kh_slot e;
                bool from_in;
                if (h < h_end && i_in < n_in) from_in = in[i_in].key < (uint32_t)pkeys[h];
                else from_in = i_in < n_in;
                if (from_in) e = in[i_in++];
                else { e.key = (uint32_t)pkeys[h]; e.val = pvals[h]; h++; }
                if (filled < KH_SLOTS_PER_BUCKET) { dst[filled++] = e; displaced += from_in; }
                else out[n_out++] = e;
            }
            for (; filled < KH_SLOTS_PER_BUCKET; filled++) dst[filled] = kh_slot{ KH_EMPTY_KEY, 0xFFFFFFFFu };
            if (n_out != carry[cur]) st->bad = 2;
            if (n_out == 0) break;
            in = out;
            n_in = n_out;
            cur = cur + 1 == n_buckets ? 0 : cur + 1;
        }
    }
    displaced = wave_sum(displaced);
    if ((threadIdx.x & 63u) == 0 && displaced) atomicAdd(&st->n_displaced, displaced);
}

inline unsigned blocks_for(uint64_t n, unsigned per) { return (unsigned)((n + per - 1) / per); }

}  // namespace

// The whole build; on success *out owns two device allocations (hipFree).  Inputs are host buffers.
int kaamer_build_on_device(const uint8_t *seqs, const uint64_t *offsets, const uint32_t *ids, uint32_t n_proteins,
                           uint32_t shard, uint32_t n_shards, double load, int device, kaamer_device_image *out)
{
    out->d_buckets = nullptr;
    out->d_arena = nullptr;
    if (!(load > 0.05 && load <= 0.95)) load = 0.5;
    for (uint32_t p = 0; p < n_proteins; p++) {
        if (offsets[p + 1] < offsets[p]) return kaamer_fail(KAAMER_E_ARG, "build_proteins: offsets decrease at %u", p);
        if (ids && ids[p] == KH_EMPTY_PID) return kaamer_fail(KAAMER_E_ARG, "protein id 0xFFFFFFFF is reserved");
    }
    if (!ids && n_proteins && n_proteins - 1 == KH_EMPTY_PID) return kaamer_fail(KAAMER_E_ARG, "protein id 0xFFFFFFFF is reserved");
    BD_HIP(hipSetDevice(device));
    Trace tr;
    const uint64_t origin = n_proteins ? offsets[0] : 0, total_res = n_proteins ? offsets[n_proteins] - origin : 0;

    DevMem d_st;
    BD_HIP(d_st.alloc(sizeof(BuildStats)));
    BD_HIP(hipMemset(d_st.p, 0, sizeof(BuildStats)));
    BuildStats *st = d_st.as<BuildStats>();
    BuildStats hs;
    memset(&hs, 0, sizeof hs);

    // ---- emit ----------------------------------------------------------------------------------------------
    DevMem d_pairs, d_pairs_alt;
    uint64_t n_emit = 0;
    if (total_res >= KAAMER_KMER_SIZE) {
        DevMem d_seqs, d_off, d_ids;
        BD_HIP(d_seqs.alloc(total_res + 32));   // (a thread of the emit kernel reads 24 bytes from its first position)
        BD_HIP(d_off.alloc(((size_t)n_proteins + 1) * 8));
        BD_HIP(hipMemcpy(d_seqs.p, seqs + origin, total_res, hipMemcpyHostToDevice));
        BD_HIP(hipMemcpy(d_off.p, offsets, ((size_t)n_proteins + 1) * 8, hipMemcpyHostToDevice));
        if (ids) {
            BD_HIP(d_ids.alloc((size_t)n_proteins * 4));
            BD_HIP(hipMemcpy(d_ids.p, ids, (size_t)n_proteins * 4, hipMemcpyHostToDevice));
        }
        tr.lap("upload", total_res);
        const uint64_t n_tiles = (total_res + BD_TILE - 1) / BD_TILE;
        if (n_tiles >= (1ull << 31)) return kaamer_fail(KAAMER_E_CAPACITY, "device build: database too large for one call");
        hipLaunchKernelGGL(bd_windows_kernel<false>, dim3((unsigned)n_tiles), dim3(BD_THREADS), 0, 0, d_seqs.as<uint8_t>(), d_off.as<uint64_t>(),
                           d_ids.as<uint32_t>(), n_proteins, origin, total_res, shard, n_shards, (uint64_t *)nullptr, 0ull, st);
        BD_HIP(hipMemcpy(&hs, st, sizeof hs, hipMemcpyDeviceToHost));
        n_emit = hs.n_windows;
        tr.lap("count windows", n_emit);
        BD_HIP(d_pairs.alloc(n_emit * 8));
        BD_HIP(d_pairs_alt.alloc(n_emit * 8));
        BD_HIP(hipMemset(st, 0, sizeof(BuildStats)));
        hipLaunchKernelGGL(bd_windows_kernel<true>, dim3((unsigned)n_tiles), dim3(BD_THREADS), 0, 0, d_seqs.as<uint8_t>(), d_off.as<uint64_t>(),
                           d_ids.as<uint32_t>(), n_proteins, origin, total_res, shard, n_shards, d_pairs.as<uint64_t>(), n_emit, st);
        BD_HIP(hipMemcpy(&hs, st, sizeof hs, hipMemcpyDeviceToHost));
        if (hs.bad || hs.n_windows != n_emit) return kaamer_fail(KAAMER_E_HIP, "device build: the emit pass disagrees with the count pass");
        tr.lap("emit", n_emit);
    }

    // ---- sort + unique -------------------------------------------------------------------------------------
    DevMem d_tmp;
    size_t tmp_bytes = 0;
    uint64_t m = 0;
    uint64_t *pairs = nullptr;   // sorted unique pairs
    if (n_emit) {
        rocprim::double_buffer<uint64_t> db(d_pairs.as<uint64_t>(), d_pairs_alt.as<uint64_t>());
        BD_HIP(rocprim::radix_sort_keys(nullptr, tmp_bytes, db, (size_t)n_emit, 0u, 64u));
        BD_HIP(d_tmp.alloc(tmp_bytes));
        BD_HIP(rocprim::radix_sort_keys(d_tmp.p, tmp_bytes, db, (size_t)n_emit, 0u, 64u));
        tr.lap("sort pairs", n_emit);
        uint64_t *sorted = db.current(), *other = db.alternate();
        DevMem d_count;
        BD_HIP(d_count.alloc(8));
        size_t need = 0;
        BD_HIP(rocprim::unique(nullptr, need, sorted, other, d_count.as<size_t>(), (size_t)n_emit, rocprim::equal_to<uint64_t>()));
        if (need > tmp_bytes) { BD_HIP(d_tmp.alloc(need)); tmp_bytes = need; }
        BD_HIP(rocprim::unique(d_tmp.p, need, sorted, other, d_count.as<size_t>(), (size_t)n_emit, rocprim::equal_to<uint64_t>()));
        size_t cnt = 0;
        BD_HIP(hipMemcpy(&cnt, d_count.p, sizeof cnt, hipMemcpyDeviceToHost));
        m = cnt;
        pairs = other;
        // the buffer the unique pairs are NOT in goes back
        if (sorted == d_pairs.as<uint64_t>()) d_pairs.release(); else d_pairs_alt.release();
        tr.lap("unique", m);
    }
    if (m >= 0xFFFFFFFFull) return kaamer_fail(KAAMER_E_CAPACITY, "device build: %llu pairs in one shard; use more shards", (unsigned long long)m);

    // ---- keys ----------------------------------------------------------------------------------------------
    DevMem d_kstart;
    uint32_t n_keys = 0;
    BD_HIP(d_kstart.alloc(((size_t)m + 2) * 4));
    if (m) {
        DevMem d_count;
        BD_HIP(d_count.alloc(8));
        auto flags = rocprim::make_transform_iterator(rocprim::counting_iterator<uint32_t>(0), KeyHead{ pairs });
        size_t need = 0;
        BD_HIP(rocprim::select(nullptr, need, rocprim::counting_iterator<uint32_t>(0), flags, d_kstart.as<uint32_t>(), d_count.as<size_t>(), (size_t)m));
        if (need > tmp_bytes) { BD_HIP(d_tmp.alloc(need)); tmp_bytes = need; }
        BD_HIP(rocprim::select(d_tmp.p, need, rocprim::counting_iterator<uint32_t>(0), flags, d_kstart.as<uint32_t>(), d_count.as<size_t>(), (size_t)m));
        size_t cnt = 0;
        BD_HIP(hipMemcpy(&cnt, d_count.p, sizeof cnt, hipMemcpyDeviceToHost));
        n_keys = (uint32_t)cnt;
        const uint32_t m32 = (uint32_t)m;
        BD_HIP(hipMemcpy(d_kstart.as<uint32_t>() + n_keys, &m32, 4, hipMemcpyHostToDevice));
        tr.lap("key starts", n_keys);
    }

    kh_image_header &hdr = out->hdr;
    memset(&hdr, 0, sizeof hdr);
    hdr.magic = KH_IMAGE_MAGIC;
    hdr.version = KH_IMAGE_VERSION;
    hdr.kmer_size = KAAMER_KMER_SIZE;
    hdr.shard = shard;
    hdr.n_shards = n_shards;
    hdr.load_factor = load;
    hdr.n_pairs = m;
    hdr.n_keys = n_keys;
    const uint64_t n_buckets = (uint64_t)((double)n_keys / (KH_SLOTS_PER_BUCKET * load)) + 1;   // builder.cpp
    if (n_buckets >= (1ull << 32)) return kaamer_fail(KAAMER_E_ARG, "too many buckets");
    hdr.n_buckets = n_buckets;

    DevMem d_buckets, d_arena;
    BD_HIP(d_buckets.alloc((size_t)n_buckets * sizeof(kh_bucket)));
    uint64_t arena_words = 4;
    if (n_keys) {
        KeyArrays a;
        memset(&a, 0, sizeof a);
        a.pairs = pairs;
        a.kstart = d_kstart.as<uint32_t>();
        a.n_keys = n_keys;
        a.n_buckets = n_buckets;
        a.n_shards = n_shards;
        a.st = st;
        DevMem d_units, d_uoff, d_hv, d_hv_alt, d_hk, d_hk_alt, d_run, d_rep, d_pkeys, d_pkeys_alt, d_pvals, d_pvals_alt;
        BD_HIP(d_units.alloc((size_t)n_keys * 4));
        BD_HIP(d_uoff.alloc((size_t)n_keys * 4));
        BD_HIP(d_hv.alloc((size_t)n_keys * 8));
        BD_HIP(d_hv_alt.alloc((size_t)n_keys * 8));
        BD_HIP(d_hk.alloc((size_t)n_keys * 4));
        BD_HIP(d_hk_alt.alloc((size_t)n_keys * 4));
        BD_HIP(d_run.alloc((size_t)n_keys * 4));
        BD_HIP(d_rep.alloc((size_t)n_keys * 4));
        a.units = d_units.as<uint32_t>();
        a.uoff = d_uoff.as<uint32_t>();
        a.rep_of = d_rep.as<uint32_t>();
        const unsigned kb = blocks_for(n_keys, 256);
        for (int attempt = 0;; attempt++) {
            a.seed = 0x9E3779B97F4A7C15ull * (uint64_t)(attempt + 1);
            if (getenv("KAAMER_BUILD_WEAK_HASH") && attempt == 0) a.seed = 0;  // tests: forces the collision path
            BD_HIP(hipMemset(st, 0, sizeof(BuildStats)));
            a.hv = d_hv.as<uint64_t>();
            a.hk = d_hk.as<uint32_t>();
            hipLaunchKernelGGL(bd_hash_kernel, dim3(kb < 4096u ? kb : 4096u), dim3(256), 0, 0, a);
            {
                rocprim::double_buffer<uint64_t> hb(d_hv.as<uint64_t>(), d_hv_alt.as<uint64_t>());
                rocprim::double_buffer<uint32_t> kb3(d_hk.as<uint32_t>(), d_hk_alt.as<uint32_t>());
                size_t need = 0;
                BD_HIP(rocprim::radix_sort_pairs(nullptr, need, hb, kb3, (size_t)n_keys, 0u, 64u));
                if (need > tmp_bytes) { BD_HIP(d_tmp.alloc(need)); tmp_bytes = need; }
                BD_HIP(rocprim::radix_sort_pairs(d_tmp.p, need, hb, kb3, (size_t)n_keys, 0u, 64u));
                a.hv_s = hb.current();
                a.hk_s = kb3.current();
                auto heads = rocprim::make_transform_iterator(rocprim::counting_iterator<uint32_t>(0), RunHead{ a.hv_s });
                need = 0;
                BD_HIP(rocprim::inclusive_scan(nullptr, need, heads, d_run.as<uint32_t>(), (size_t)n_keys, MaxU32()));
                if (need > tmp_bytes) { BD_HIP(d_tmp.alloc(need)); tmp_bytes = need; }
                BD_HIP(rocprim::inclusive_scan(d_tmp.p, need, heads, d_run.as<uint32_t>(), (size_t)n_keys, MaxU32()));
                a.run_start = d_run.as<uint32_t>();
            }
            hipLaunchKernelGGL(bd_rep_kernel, dim3(kb), dim3(256), 0, 0, a);
            hipLaunchKernelGGL(bd_share_kernel, dim3(kb < 4096u ? kb : 4096u), dim3(256), 0, 0, a);
            BD_HIP(hipMemcpy(&hs, st, sizeof hs, hipMemcpyDeviceToHost));
            if (!hs.collision) {
                if (tr.on) fprintf(stderr, "[kaamer device build] content-hash attempts: %d\n", attempt + 1);
                break;
            }
            if (attempt == 3) return kaamer_fail(KAAMER_E_CAPACITY, "device build: content hashes collide under four seeds");
        }
        tr.lap("hash + share", hs.n_lists);
        if (hs.ub_units + 1 >= KH_INLINE_BIT) return kaamer_fail(KAAMER_E_ARG, "arena exceeds 32 GiB per shard");   // builder.cpp's bound
        d_hv.release(); d_hv_alt.release(); d_hk.release(); d_hk_alt.release(); d_run.release();
        {
            size_t need = 0;
            BD_HIP(rocprim::exclusive_scan(nullptr, need, a.units, a.uoff, 1u, (size_t)n_keys, rocprim::plus<uint32_t>()));
            if (need > tmp_bytes) { BD_HIP(d_tmp.alloc(need)); tmp_bytes = need; }
            BD_HIP(rocprim::exclusive_scan(d_tmp.p, need, a.units, a.uoff, 1u, (size_t)n_keys, rocprim::plus<uint32_t>()));
            uint32_t last_off = 0, last_units = 0;
            BD_HIP(hipMemcpy(&last_off, a.uoff + (n_keys - 1), 4, hipMemcpyDeviceToHost));
            BD_HIP(hipMemcpy(&last_units, a.units + (n_keys - 1), 4, hipMemcpyDeviceToHost));
            arena_words = ((uint64_t)last_off + last_units) * 4;
        }
        BD_HIP(d_arena.alloc((size_t)arena_words * 4));
        BD_HIP(hipMemset(d_arena.p, 0, 16));
        BD_HIP(d_pkeys.alloc((size_t)n_keys * 8));
        BD_HIP(d_pvals.alloc((size_t)n_keys * 4));
        a.arena = d_arena.as<uint32_t>();
        a.pkeys = d_pkeys.as<uint64_t>();
        a.pvals = d_pvals.as<uint32_t>();
        hipLaunchKernelGGL(bd_lists_kernel, dim3(kb), dim3(256), 0, 0, a);
        BD_HIP(hipGetLastError());
        tr.lap("arena", arena_words);
        // the pairs are not needed any more
        BD_HIP(hipDeviceSynchronize());
        d_pairs.release(); d_pairs_alt.release(); d_units.release(); d_uoff.release(); d_rep.release();

        // ---- placement -------------------------------------------------------------------------------------
        BD_HIP(d_pkeys_alt.alloc((size_t)n_keys * 8));
        BD_HIP(d_pvals_alt.alloc((size_t)n_keys * 4));
        rocprim::double_buffer<uint64_t> kb2(d_pkeys.as<uint64_t>(), d_pkeys_alt.as<uint64_t>());
        rocprim::double_buffer<uint32_t> vb2(d_pvals.as<uint32_t>(), d_pvals_alt.as<uint32_t>());
        {
            size_t need = 0;
            BD_HIP(rocprim::radix_sort_pairs(nullptr, need, kb2, vb2, (size_t)n_keys, 0u, 64u));
            if (need > tmp_bytes) { BD_HIP(d_tmp.alloc(need)); tmp_bytes = need; }
            BD_HIP(rocprim::radix_sort_pairs(d_tmp.p, need, kb2, vb2, (size_t)n_keys, 0u, 64u));
        }
        tr.lap("sort by home bucket", n_keys);
        DevMem d_bstart, d_pre, d_carry, d_ooff, d_spill;
        BD_HIP(d_bstart.alloc(((size_t)n_buckets + 1) * 4));
        BD_HIP(d_pre.alloc((size_t)n_buckets * sizeof(CarryMap)));
        BD_HIP(d_carry.alloc((size_t)n_buckets * 4));
        BD_HIP(d_ooff.alloc((size_t)n_buckets * 4));
        hipLaunchKernelGGL(bd_bucket_start_kernel, dim3(blocks_for(n_buckets + 1, 256)), dim3(256), 0, 0, kb2.current(), n_keys, n_buckets, d_bstart.as<uint32_t>());
        {
            auto maps = rocprim::make_transform_iterator(rocprim::counting_iterator<uint32_t>(0), CarryOfBucket{ d_bstart.as<uint32_t>() });
            size_t need = 0;
            BD_HIP(rocprim::inclusive_scan(nullptr, need, maps, d_pre.as<CarryMap>(), (size_t)n_buckets, CarryCompose()));
            if (need > tmp_bytes) { BD_HIP(d_tmp.alloc(need)); tmp_bytes = need; }
            BD_HIP(rocprim::inclusive_scan(d_tmp.p, need, maps, d_pre.as<CarryMap>(), (size_t)n_buckets, CarryCompose()));
        }
        BD_HIP(hipMemset(st, 0, sizeof(BuildStats)));
        hipLaunchKernelGGL(bd_carry_kernel, dim3(blocks_for(n_buckets, 256)), dim3(256), 0, 0, d_pre.as<CarryMap>(), n_buckets, d_carry.as<uint32_t>(), st);
        BuildStats ps;
        BD_HIP(hipMemcpy(&ps, st, sizeof ps, hipMemcpyDeviceToHost));
        if (ps.hops >= 0xFFFFFFFFull) return kaamer_fail(KAAMER_E_CAPACITY, "device build: probe sequences too long (load factor %.2f)", load);
        {
            size_t need = 0;
            BD_HIP(rocprim::exclusive_scan(nullptr, need, d_carry.as<uint32_t>(), d_ooff.as<uint32_t>(), 0u, (size_t)n_buckets, rocprim::plus<uint32_t>()));
            if (need > tmp_bytes) { BD_HIP(d_tmp.alloc(need)); tmp_bytes = need; }
            BD_HIP(rocprim::exclusive_scan(d_tmp.p, need, d_carry.as<uint32_t>(), d_ooff.as<uint32_t>(), 0u, (size_t)n_buckets, rocprim::plus<uint32_t>()));
        }
        BD_HIP(d_spill.alloc(((size_t)ps.hops + 1) * sizeof(kh_slot)));
        hipLaunchKernelGGL(bd_place_kernel, dim3(blocks_for(n_buckets, 256)), dim3(256), 0, 0, kb2.current(), vb2.current(), d_bstart.as<uint32_t>(),
                           d_carry.as<uint32_t>(), d_ooff.as<uint32_t>(), d_spill.as<kh_slot>(), d_buckets.as<kh_bucket>(), n_buckets, st);
        BD_HIP(hipMemcpy(&ps, st, sizeof ps, hipMemcpyDeviceToHost));
        if (ps.bad) return kaamer_fail(KAAMER_E_HIP, "device build: placement disagrees with its carry scan");
        tr.lap("placement", ps.hops);
        hdr.n_displaced = ps.n_displaced;
        hdr.n_inline = hs.n_inline;
        hdr.n_lists = hs.n_lists;
        hdr.max_list = hs.max_list;
        hdr.max_protein_id = (uint32_t)hs.max_pid;
    } else {
        BD_HIP(hipMemset(d_buckets.p, 0xFF, (size_t)n_buckets * sizeof(kh_bucket)));
        BD_HIP(d_arena.alloc(16));
        BD_HIP(hipMemset(d_arena.p, 0, 16));
    }
    hdr.arena_words = arena_words;
    BD_HIP(hipDeviceSynchronize());
    out->d_buckets = (kh_bucket *)d_buckets.take();
    out->d_arena = (uint32_t *)d_arena.take();
    return KAAMER_OK;
}

extern "C" int kaamer_image_build_proteins_device(const uint8_t *seqs, const uint64_t *offsets, const uint32_t *ids,
                                                  uint32_t n_proteins, uint32_t shard, uint32_t n_shards, double load_factor,
                                                  int device, kaamer_image **out)
{
    if (!out || !offsets || (!seqs && n_proteins) || n_shards == 0 || shard >= n_shards)
        return kaamer_fail(KAAMER_E_ARG, "build_proteins_device: bad argument");
    *out = nullptr;
    kaamer_device_image di;
    int rc = kaamer_build_on_device(seqs, offsets, ids, n_proteins, shard, n_shards, load_factor, device, &di);
    if (rc) return rc;
    Trace tr;
    kaamer_image *img = new (std::nothrow) kaamer_image();
    if (!img) rc = kaamer_fail(KAAMER_E_NOMEM, "image alloc");
    if (!rc) {
        img->hdr = di.hdr;
        rc = kaamer_image_alloc(img, di.hdr.n_buckets, di.hdr.arena_words);
        if (rc) kaamer_fail(rc, "image buffers");
    }
    if (!rc) {
        hipError_t e = hipMemcpy(img->buckets, di.d_buckets, (size_t)di.hdr.n_buckets * sizeof(kh_bucket), hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(img->arena, di.d_arena, (size_t)di.hdr.arena_words * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = kaamer_fail(KAAMER_E_HIP, "device build: download: %s", hipGetErrorString(e));
    }
    (void)hipFree(di.d_buckets);
    (void)hipFree(di.d_arena);
    if (rc) { if (img) kaamer_image_free(img); return rc; }
    tr.lap("download", di.hdr.n_buckets * 64 + di.hdr.arena_words * 4);
    *out = img;
    return KAAMER_OK;
}

extern "C" int kaamer_image_build_makedb_device(const kaamer_proteins *p, uint32_t shard, uint32_t n_shards, double load_factor,
                                                int device, kaamer_image **out)
{
    if (!p || !out) return kaamer_fail(KAAMER_E_ARG, "image_build_makedb_device: bad argument");
    const uint8_t *seqs; const uint64_t *offsets; const uint32_t *ids; uint32_t n;
    kaamer_proteins_raw(p, &seqs, &offsets, &ids, &n);
    return kaamer_image_build_proteins_device(seqs, offsets, ids, n, shard, n_shards, load_factor, device, out);
}
