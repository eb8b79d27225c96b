// search.hip — gfx950 kernels and the C-ABI entry points of the k-mer search path.
//
// What runs on the device, per batch (all on the caller's stream, no host sync):
//
//   translate_short_kernel / translate_kernel (COUNT + WRITE) + orf_order_kernel
//                         nucleotide input only (translate.hip.inc): GetORFs (dna.go:65-181)
//                         -> ORF batch; a lane per read for reads, a wave per (sequence, frame)
//                         for longer sequences
//   prep_layout_schedule_kernel   protein input (count_group.hip.inc): Query.SizeInKmer etc.
//                         (search.go:290-293; search_protein.go:70-76), the bitmap of positions
//                         that start no k-mer, an LDS table capacity per query, the table
//                         layout E (exclusive scan), the first query of every query group and
//                         the order the groups are handed out in (longest first) -- one launch
//   prep_orf_kernel + layout_kernel + schedule_kernel    the same for an ORF batch
//   probe_kernel          the dominant kernel: flat over residue positions; sliding 7-mer
//                         encode (k_store.go:91-117, search_protein.go:94-98) and bucket
//                         probe (replaces KmerStore.Get, search.go:421) -> vals[pos]
//   count_group_kernel    (count_group.hip.inc) postings expansion + Counter increments
//                         (search.go:427-436, 442-452) in LDS hash tables, one 8-wave
//                         workgroup per query group; a wave per query packs its table into
//                         the query's hit list at E[q]
//   count_global_kernel   G tier: queries whose distinct hits exceed their LDS table count
//                         into an exactly sized table in HBM; its last workgroup finalizes
//                         the batch (counters, status, per-batch state left zeroed)
//   [scan + gather_hits_kernel]   opts.compact: hit lists -> CSR in query order
//   [positions pass]      PositionHits bitmaps when asked for
//   [topn_kernel]         kaamer_topn_device: sortMapByValue order, SetBestStartCodon,
//                         FilterResults (topn.hip.inc)
//
// This is integer hashing/indexing: no MFMA.  The probe is bound by the memory system's
// random-request rate (~51e9 requests/s on MI355X whatever the size up to 128 B,
// tools/random_read_bench.hip), i.e. ~3.3 TB/s for 64-byte buckets.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <mutex>
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
struct kaamer_workspace;
// the host-buffer calls keep their workspace and device staging buffers between calls
// (creating and freeing ~1 GB of device memory per call cost 10x the search itself)
struct HostSlot {
    kaamer_workspace *ws = nullptr;
    kaamer_workspace_opts opts{};
    uint8_t *d_seqs = nullptr;
    uint64_t *d_off = nullptr;
    size_t seq_cap = 0, off_cap = 0;
    hipStream_t stream = nullptr;   // every slot works on its own stream
    bool busy = false;              // (under kaamer_index::pool_mu)
};

// one in-flight call of the pipelined host-buffer boundary (host_top.hip.inc)
struct TopSlot {
    kaamer_workspace *ws = nullptr;
    kaamer_workspace_opts opts{};
    hipStream_t stream = nullptr;
    uint8_t *d_seqs = nullptr;
    uint64_t *d_off = nullptr;
    size_t seq_cap = 0, off_cap = 0;
    uint8_t *h_in = nullptr;   // pinned: the caller's sequences, then (8-byte aligned) its offsets
    size_t h_in_cap = 0;
    uint8_t *d_block = nullptr;
    size_t d_block_cap = 0;
    size_t guess = 1u << 16;   // bytes of the block copied before its size is known (adapts to the previous call)
    bool busy = false;
};
#define KAAMER_MAX_HOST_SLOTS 8
static void top_slot_free(TopSlot &h);

struct kaamer_index {
    int device;
    kh_image_header hdr;
    kh_bucket *d_buckets;
    uint32_t *d_arena;
    // kaamer_search_batch (full hit lists): a call takes a free slot (workspace + staging + stream) and gives it back;
    // callers beyond the slots wait for one.  Same pool lock as the slots below.
    HostSlot host[4];
    // kaamer_submit_batch_top / kaamer_search_batch_top / kaamer_stream_*: a pool of slots, one per call in flight
    std::mutex pool_mu;
    std::condition_variable pool_cv;
    TopSlot top[KAAMER_MAX_HOST_SLOTS];
    int n_top;
};

enum { ST_POOL_FULL = 1u, ST_LIST_FULL = 2u, ST_QUERY_CAP = 4u, ST_AA_CAP = 8u, ST_G_ARENA_FULL = 16u, ST_G_TABLE_FULL = 32u,
       ST_POS_UNSUPPORTED = 64u, ST_POS_CAP = 128u, ST_CHAIN_TIMEOUT = 256u, ST_EXCHANGE_CAP = 512u, ST_PEER_FAILED = 1024u };
enum { CTR_IN = 0, CTR_QUERIES, CTR_LOOKUP, CTR_PROBE, CTR_FOUND, CTR_POST, CTR_HITS, CTR_OVERFLOW, CTR_LISTS, CTR_LIST_IDS, CTR_N };
static_assert(sizeof(kaamer_counters) == CTR_N * 8, "counter layout");
#define CTR_REPLICAS 64

// work lists of the counting tiers (device memory)
enum { LIST_S = 0, LIST_L, LIST_SO, LIST_G, N_LISTS };
// small per-batch device state after the list counters (all zeroed by the finalize step)
enum { SLOT_QUEUE_HEAD = N_LISTS, SLOT_STATUS = N_LISTS + 1, SLOT_GROUP_QUEUE = N_LISTS + 2, SLOT_GROUP_QUEUE_POS = N_LISTS + 3,
       SLOT_N_LONG = N_LISTS + 4, SLOT_TILES_DONE = N_LISTS + 5, SLOT_TR_TICKET = N_LISTS + 6,
       SLOT_QUEUE_SUB = N_LISTS + 7 /* 8 words */, SLOT_G_TICKET = N_LISTS + 15, SLOT_PACK_TICKETS = N_LISTS + 16 /* 64 words */,
       SLOT_G_HITS16 = N_LISTS + 80 /* hits the G tier found, in sixteens (saturating) */,
       SLOT_G_MON_HITS16 = N_LISTS + 81 /* those of its monsters (G_MONSTER_HITS) */, SLOT_G_MONSTERS = N_LISTS + 82 /* how many monsters */,
       N_SMALL_SLOTS = N_LISTS + 16 + 64 + 3 };
// a query with more distinct hits than this fits no LDS table at any scale: it says nothing about the tables of the others
#define G_MONSTER_HITS (GRP_MAX_TABLE / 4u * 3u)
static_assert(N_SMALL_SLOTS <= 256, "the finalize step zeroes one slot per thread");
// one entry: everything a tier needs to start on a query, in one 16-byte load
struct alignas(16) WorkItem {
    uint32_t q;
    int32_t size;     // SizeInKmer
    uint64_t aa_off;  // first residue position of the query
};

#define CURSOR_STRIDE 32u  /* unsigned long long words between cursors (256 B) */
#define MAX_TIMED_CALLS 1024u
#define L_WAVES 8
#define L_LOG2CAP 12
#define G_WAVES 8
#ifndef S_MIN_WAVES
#define S_MIN_WAVES 1
#endif
#ifndef L_MIN_WAVES
#define L_MIN_WAVES 1
#endif
#ifndef S_NWIN
#define S_NWIN 3
#endif

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

__device__ __forceinline__ uint32_t wave_sum32(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// wave total as a scalar (uniform) value
__device__ __forceinline__ uint32_t wave_total(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_sum32(v));
}

__device__ __forceinline__ void add_counter(unsigned long long *replicas, uint32_t replica, int which, unsigned long long v)
{
    if (v) atomicAdd(&replicas[(size_t)(replica % CTR_REPLICAS) * CTR_N + which], v);
}

// ====================================================================================
// Kernel P — flat probe over residue positions
// ====================================================================================
// Position i of the packed residue buffer is a k-mer start iff bit (i & 63) of
// valid[i >> 6] is set (prep clears the tail of every query and whole queries that are
// too short).  One wave handles 64 consecutive positions: residue codes are staged in
// LDS, each lane encodes its 7-mer (k_store.go:91-117 in closed form), then the wave
// probes the bucket table with 4 lanes per 64-byte bucket (16 B each, one fabric
// sector per probe) and writes one u32 per position:
//     vals[i] = 0            key absent (or position not a k-mer start)
//             = slot.val     key present: inline protein id or postings-list offset
// This replaces KmerStore.GetValueFromBadger (search.go:421) for the whole batch at
// once; the work is perfectly balanced whatever the query lengths are.
struct ProbeParams {
    const uint4 *table;  // buckets viewed as 4 x uint4 each
    uint64_t n_buckets;
    uint32_t n_shards, shard;
    const uint8_t *residues;
    unsigned long long *invalid;  // one bit per position, set = not a k-mer start; self-cleaning
    const unsigned long long *d_n_pos;  // device scalar: number of residue positions
    uint32_t *vals;
    unsigned long long *counters;
    uint32_t nontemporal;  // bucket loads bypass the caches (small batch against a large table)
};

#define P_WAVES 4
#define P_RING 128u            /* deferred lookups per wave (a power of two): fewer than 64 waiting + the 64 of a window */
#define KH_NO_KEY 0xFFFFFFFDu  /* "no lookup for this position": equals no slot key (valid keys <= 0xE773B9D4, 0xFFFFFFFF = empty) */

// SKIP_IDLE: lanes without a lookup issue no bucket load.  A nucleotide batch has one position in six that starts no
// k-mer (the gaps between ORFs); a load is a request whatever it hits, and the probe runs at the request rate of the
// memory system, so there the exec-masked loads win although their waits are coarser.  Protein batches (1.7 % idle
// positions, a few windows per wave) keep the unconditional, exactly counted loads.
template <bool SKIP_IDLE> __global__ __launch_bounds__(64 * P_WAVES) void probe_kernel(ProbeParams p)
{
    __shared__ uint8_t s_lut[256];
    __shared__ uint8_t s_stage[P_WAVES][80];
    // Lookups whose bucket is full and does not hold the key (about 2 %) go on in the NEXT bucket of the probe sequence.
    // Waiting for that second request inside the window stalls the whole wave (three windows out of four have such a
    // lookup), so they are put on a ring in LDS (key, position, buckets walked) and looked up later, 64 at a time, as an
    // iteration of their own through the same code.  The loop body is ONE straight line for both kinds of iteration, and
    // every global load and store in it is issued unconditionally (idle lanes and idle iterations touch harmless
    // addresses): the compiler's s_waitcnt vmcnt(N) then counts exactly -- a load or store inside a branch or an inner
    // loop anywhere in the body makes every wait of the loop a vmcnt(0).
    __shared__ uint32_t s_ring[P_WAVES][3][P_RING];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));  // wave-uniform: window bases stay scalar
    const unsigned long long n_pos = *p.d_n_pos;
    const unsigned long long n_win = (n_pos + 63) >> 6;
    for (uint32_t i = tid; i < 256; i += 64 * P_WAVES) s_lut[i] = (uint8_t)kh_residue_code((uint8_t)i);
    __syncthreads();  // the only workgroup barrier: the waves run independently from here on
    uint32_t c_lookup = 0, c_probe = 0, c_found = 0;
    uint8_t *stage = s_stage[wv];
    uint32_t *const ring_key = s_ring[wv][0], *const ring_pos = s_ring[wv][1], *const ring_step = s_ring[wv][2];
    uint32_t ring_head = 0, ring_n = 0;  // wave-uniform
    const uint32_t n_buckets = (uint32_t)p.n_buckets;
    const uint32_t mine = 16u * (lane & 3u) + (lane >> 2);  // the lane whose lookup this lane ends up owning (see the rounds)

    const unsigned long long stride = (unsigned long long)gridDim.x * P_WAVES;
    unsigned long long w = (unsigned long long)blockIdx.x * P_WAVES + wv;
    // the bitmap word and the residues of the NEXT window are loaded behind the bucket loads of the current one;
    // windows and reads past the end are clamped (the positions involved are marked invalid or unused)
    unsigned long long mask = 0;
    uint32_t ra = 0, rb = 0;
    auto fetch = [&](unsigned long long win, unsigned long long &m, uint32_t &a, uint32_t &b) {
        const unsigned long long wc = win < n_win ? win : n_win - 1;
        m = p.invalid[wc];
        const unsigned long long i0 = (wc << 6) + lane, last = n_pos - 1;
        a = p.residues[i0 < last ? i0 : last];
        b = p.residues[i0 + 64 < last ? i0 + 64 : last];  // lanes 0..5 hold the halo
    };
    if (n_win) fetch(w, mask, ra, rb);

    while (n_win) {
        const bool ring = ring_n >= 64u || (w >= n_win && ring_n > 0u);  // wave-uniform: this iteration serves the ring
        if (!ring && w >= n_win) break;
        unsigned long long base = 0;
        uint32_t key = KH_NO_KEY, off = lane, step = 0;
        if (ring) {
            const uint32_t n = ring_n < 64u ? ring_n : 64u;
            if (lane < n) {
                const uint32_t slot = (ring_head + lane) & (P_RING - 1u);
                key = ring_key[slot]; off = ring_pos[slot]; step = ring_step[slot];
            }
            ring_head = (ring_head + n) & (P_RING - 1u);
            ring_n -= n;
        } else {
            base = w << 6;
            stage[lane] = s_lut[ra];
            if (lane < 6) stage[64 + lane] = s_lut[rb];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint8_t *st = stage + lane;
            key = kh_key_from_codes(st[0], st[1], st[2], st[3], st[4], st[5], st[6]);
            // not a k-mer start, or (sharded index) a key another device owns: no lookup, the position reads as absent
            const bool own = p.n_shards <= 1 || kh_shard_of(key, p.n_shards) == p.shard;
            key = (((~mask >> lane) & 1ull) && own) ? key : KH_NO_KEY;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            c_lookup += key != KH_NO_KEY ? 1u : 0u;
        }
        c_probe += key != KH_NO_KEY ? 1u : 0u;
        // ---- the next regular window's inputs first (a ring iteration fetches the pending window's again): they are
        // older than the bucket loads, so waiting for the buckets covers them, and the stores at the end of the
        // iteration are younger than every load -- nothing ever waits for a store to be acknowledged
        const unsigned long long w_cur = w;
        if (!ring) w += stride;
        fetch(w, mask, ra, rb);

        // ---- issue: round j serves the lookups of lanes 16j..16j+15, four lanes per 64-byte bucket (16 B each: one
        // fabric sector per probe); nontemporal when the batch is small against the table (a bucket is then read once per
        // batch; a 1 M-read batch touches every bucket several times and wants them cached).  Every lane hashes ITS OWN key
        // once and the rounds pass the bucket index around (the hash per round was computed four times per lookup: 60 of
        // the loop's 270 VALU instructions per window).
        uint32_t my_bucket = 0u;
        if (key != KH_NO_KEY) {
            my_bucket = (uint32_t)kh_home_bucket(key, p.n_shards, p.n_buckets) + step;
            my_bucket = my_bucket >= n_buckets ? my_bucket - n_buckets : my_bucket;
        }
        uint4 ld[4];
        uint32_t rk[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int src = 16 * j + (int)(lane >> 2);
            rk[j] = __shfl(key, src, 64);
            const uint32_t bucket = __shfl(my_bucket, src, 64);
            typedef uint32_t v4u __attribute__((ext_vector_type(4)));
            const v4u *srcp = reinterpret_cast<const v4u *>(p.table) + (uint64_t)bucket * 4 + (lane & 3u);
            if (SKIP_IDLE) {
                ld[j] = make_uint4(KH_EMPTY_KEY, 0u, KH_EMPTY_KEY, 0u);
                if (rk[j] != KH_NO_KEY) {
                    const v4u t = *srcp;
                    ld[j] = make_uint4(t.x, t.y, t.z, t.w);
                }
            } else {
                const v4u t = p.nontemporal ? __builtin_nontemporal_load(srcp) : *srcp;
                ld[j] = make_uint4(t.x, t.y, t.z, t.w);
            }
        }
        // ---- consume: lane 4g+j ends up owning the lookup of lane 16j+g
        uint32_t okey = KH_NO_KEY, oval = 0, oempty = 1u;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t kk = rk[j];
            uint32_t r = (ld[j].x == kk) ? ld[j].y : ((ld[j].z == kk) ? ld[j].w : 0u);
            uint32_t e = (ld[j].x == KH_EMPTY_KEY || ld[j].z == KH_EMPTY_KEY) ? 1u : 0u;
            r |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r, 0xB1, 0xf, 0xf, false);  // quad_perm [1,0,3,2]
            e |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)e, 0xB1, 0xf, 0xf, false);
            r |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r, 0x4E, 0xf, 0xf, false);  // quad_perm [2,3,0,1]
            e |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)e, 0x4E, 0xf, 0xf, false);
            const bool take = (lane & 3u) == (uint32_t)j;
            okey = take ? kk : okey; oval = take ? r : oval; oempty = take ? e : oempty;
        }
        const uint32_t ooff = __shfl(off, (int)mine, 64);  // position of the lookup this lane owns (window offset, or absolute)
        const bool valid = okey != KH_NO_KEY;
        c_found += (valid && oval != 0u) ? 1u : 0u;
        // the key is not in this bucket and the bucket is full: the lookup goes on one bucket further, from the ring
        // (the number of buckets it has walked is fetched only here: a few windows in a hundred get this far)
        const unsigned long long W = __ballot(valid && oval == 0u && oempty == 0u);
        if (W) {  // wave-uniform; LDS only
            const uint32_t ostep = __shfl(step, (int)mine, 64);
            const bool wk = valid && oval == 0u && oempty == 0u && ostep + 1u < n_buckets;
            const unsigned long long W2 = __ballot(wk);
            if (wk) {
                const uint32_t slot = (ring_head + ring_n + (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(W2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)W2, 0u))) & (P_RING - 1u);
                ring_key[slot] = okey;
                ring_pos[slot] = (uint32_t)base + ooff;  // positions fit 32 bits (checked on the host)
                ring_step[slot] = ostep + 1u;
            }
            ring_n += (uint32_t)__popcll(W2);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // one coalesced store per window: vals[i] = 0 (key absent, or not a k-mer start) or slot.val; a ring iteration
        // stores only what it found (the regular pass has already written 0 there)
        if (!ring || oval != 0u) p.vals[base + ooff] = oval;
        // leave the bitmap clean for the next batch (a ring iteration stores into the slack word behind the bitmap)
        if (lane == 0) p.invalid[ring ? n_win + 1 : w_cur] = 0ull;
    }
    const uint32_t t_lookup = wave_total(c_lookup), t_probe = wave_total(c_probe), t_found = wave_total(c_found);
    if (lane == 0) {
        const uint32_t rep = blockIdx.x * P_WAVES + wv;
        add_counter(p.counters, rep, CTR_LOOKUP, t_lookup);
        add_counter(p.counters, rep, CTR_PROBE, t_probe);
        add_counter(p.counters, rep, CTR_FOUND, t_found);
    }
}

// ====================================================================================
// Kernel C — per-query counting (postings expansion + Counter increments)
// ====================================================================================
struct QInfo;
struct CountParams {
    const uint32_t *arena;
    const uint32_t *vals;  // from kernel P, indexed like the residue buffer
    // query groups (count_group.hip.inc)
    const struct QInfo *qinfo;
    const uint64_t *slot_off;   // exclusive scan of the table capacities
    uint32_t *group_first;      // first query of each group (self-cleaning)
    const uint32_t *d_n_groups;
    const uint32_t *d_nq;
    uint32_t last_group_pass;   // this launch may clear group_first behind itself
    uint32_t pack_shift;        // count_pack_kernel: a pack = the tables that start in one window of (1 << pack_shift) slots
    uint32_t *pack_tickets;     // count_pack_kernel: PK_TICKETS counters, one per range of packs (zeroed by finalize)
    // PositionHits pass (count_group_kernel MODE 1)
    const uint64_t *pos_base;
    unsigned long long *pos_bits;
    // merge of partial hit lists (count_group_kernel MODE 2)
    const uint32_t *m_pid, *m_km, *m_fp;
    uint32_t merge_fp;  // 0: the partial entries carry no first positions
    // tier input / overflow output lists
    const WorkItem *list;
    const uint32_t *list_count;
    WorkItem *ovf_list;
    uint32_t *ovf_count;
    uint32_t list_cap;
    uint32_t *queue_head;  // workgroups of the last kernel that have finished (zeroed by finalize)
    uint32_t *group_queue; // tickets of the group kernel (zeroed by finalize)
    const uint2 *sched;    // ticket -> (group, first query), longest groups first
    const uint32_t *d_n_sched;
    // results, written straight into their final place: hits of query q are
    // [hit_off[q], hit_off[q] + q_cnt[q]) of the three SoA arrays, with hit_off[q] = E[q], the
    // first slot of the query's counting table in the batch's table layout (a table never holds
    // more distinct ids than its capacity): no allocation, no atomics.  G-tier queries (rare)
    // take space after E[n_queries] from a cursor.  An optional pass packs the lists (CSR).
    // (Measured and dropped: exact reservations on 64 cursors, +1 exposed atomic round trip and a
    // counting pass per group; an ordered layout by look-back over the groups, +30 us.)
    uint64_t *hit_off;
    uint32_t *q_cnt;
    uint32_t *hit_pid, *hit_km, *hit_fp;
    uint64_t hit_cap;                 // entries of the hit arrays
    unsigned long long *tail_cursor;  // entries handed out after E[n_queries] (G tier)
    // G tier arena
    uint32_t *g_keys;  // G-tier tables: 16-byte slots {protein id, count, lowest position, -}
    uint64_t g_slots;
    unsigned long long *g_cursor;
    uint32_t n_proteins;
    unsigned long long *counters;  // [CTR_REPLICAS][CTR_N]
    uint32_t *status;
    // set when count_global_kernel is the last kernel of the batch: its last workgroup finalizes
    kaamer_counters *fin_out;
    uint32_t *fin_small, *fin_status_out;
    unsigned long long *fin_cursors;
    uint32_t *fin_slot_scale;   // table capacity scale for the next batch (count_group.hip.inc: table_slots_for)
    uint32_t fin_scale_cap, fin_scale_margin;
};

// G-tier hit lists live after the table layout's total
__device__ __forceinline__ unsigned long long tail_alloc(const CountParams &p, uint32_t total)
{
    const unsigned long long b = p.slot_off[*p.d_nq] + atomicAdd(p.tail_cursor, (unsigned long long)total);
    if (b + total > p.hit_cap) { atomicOr(p.status, (uint32_t)ST_POOL_FULL); return ~0ull; }
    return b;
}

// counting tables: protein id -> (count, lowest matching position)
template <int LOG2CAP, bool FIRSTPOS = true> struct LdsTable {
    volatile uint32_t *keys;
    uint32_t *cnt, *minpos, *nd;
    static constexpr uint32_t CAP = 1u << LOG2CAP;
    static constexpr uint32_t LIMIT = CAP - CAP / 4;  // keep 25 % free
    // n matches of `pid`, the lowest of them at `pos`.  `nnew` counts the entries this lane
    // created: the distinct-hit counter is updated once per wave, not once per insertion
    // (64 lanes bumping one LDS word serialise).
    __device__ __forceinline__ bool add_n(uint32_t pid, uint32_t pos, uint32_t n, uint32_t &nnew) const
    {
        constexpr uint32_t MASK = CAP - 1u;
        uint32_t h = (pid * 0x9E3779B1u) >> (32 - LOG2CAP);
        for (uint32_t t = 0; t <= MASK; t++) {
            uint32_t k = keys[h];
            if (k == KH_EMPTY_PID) {
                const uint32_t old = atomicCAS((uint32_t *)&keys[h], KH_EMPTY_PID, pid);
                if (old == KH_EMPTY_PID) { nnew++; k = pid; }
                else k = old;
            }
            if (k == pid) {
                atomicAdd(&cnt[h], n);
                if (FIRSTPOS) atomicMin(&minpos[h], pos);
                return true;
            }
            h = (h + 1u) & MASK;
        }
        return false;
    }
    __device__ __forceinline__ bool over_limit() const { return *(volatile uint32_t *)nd > LIMIT; }
};

// The G tier's first attempt: a table of 8192 slots in LDS (64 KB: protein id | count in the low and lowest position in the
// high 16 bits of one word, so the query must be shorter than 65 535 k-mers).  A query of a skewed database that overflows
// its group table typically has a few thousand distinct hits behind tens of thousands of postings: here those postings
// are LDS atomics instead of line fills and write-backs of a multi-megabyte table in HBM.
#define BIG_LOG2CAP 13
struct BigLdsTable {
    uint32_t *keys, *val, *nd;
    static constexpr uint32_t CAP = 1u << BIG_LOG2CAP;
    static constexpr uint32_t LIMIT = CAP - CAP / 4;  // keep 25 % free
    __device__ __forceinline__ bool add_n(uint32_t pid, uint32_t pos, uint32_t n, uint32_t &nnew) const
    {
        uint32_t h = (pid * 0x9E3779B1u) >> (32 - BIG_LOG2CAP);
        for (uint32_t t = 0; t < 128u; t++) {  // a table this crowded is handed to the HBM path
            uint32_t k = __hip_atomic_load(&keys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (k == KH_EMPTY_PID) {
                const uint32_t old = atomicCAS(&keys[h], KH_EMPTY_PID, pid);
                // the distinct hits are counted as they come (at most 8192 bumps of one LDS word per query): the attempt is
                // given up the moment the table is three quarters full, not after every further id has walked a full table
                if (old == KH_EMPTY_PID) { k = pid; if (atomicAdd(nd, 1u) >= LIMIT) return false; }
                else k = old;
            }
            if (k == pid) {
                atomicAdd(&val[h], n);  // counts stay below 65 536: one per position of the query at most
                uint32_t old = __hip_atomic_load(&val[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                while (pos < (old >> 16)) {
                    const uint32_t seen = atomicCAS(&val[h], old, (old & 0xFFFFu) | (pos << 16));
                    if (seen == old) break;
                    old = seen;
                }
                return true;
            }
            h = (h + 1u) & (CAP - 1u);
        }
        return false;
    }
    __device__ __forceinline__ bool over_limit() const { return false; }
};

// A query with more distinct hits than the LDS table holds is partitioned by protein-id range (count_global_kernel):
// these two "tables" take the place of a counting table in the postings sweep -- the first counts the items per range,
// the second writes them to their range's bucket.  An item is (protein id, position | run length << 16).
#define G_MAX_RANGES 64u
__device__ __forceinline__ uint32_t g_range_of(uint32_t pid, uint32_t R) { return (uint32_t)(((uint64_t)(pid * 0x85EBCA6Bu) * R) >> 32); }
struct RangeHist {
    uint32_t *cnt, *nd;
    uint32_t R;
    __device__ __forceinline__ bool add_n(uint32_t pid, uint32_t, uint32_t, uint32_t &) const { atomicAdd(&cnt[g_range_of(pid, R)], 1u); return true; }
    __device__ __forceinline__ bool over_limit() const { return false; }
};
struct RangeScatter {
    uint32_t *cur, *nd;
    uint2 *items;
    uint32_t R;
    const uint32_t *first;  // first item of every bucket
    uint32_t cap;           // items a bucket holds (optimistic partition: the same for all; exact sizes: no limit)
    uint32_t *ovf;          // set when a bucket is full
    __device__ __forceinline__ bool add_n(uint32_t pid, uint32_t pos, uint32_t n, uint32_t &) const
    {
        const uint32_t r = g_range_of(pid, R);
        const uint32_t at = atomicAdd(&cur[r], 1u);
        if (at - first[r] < cap) items[at] = make_uint2(pid, pos | (n << 16));
        else *ovf = 1u;
        return true;
    }
    __device__ __forceinline__ bool over_limit() const { return false; }
};

// A slot is 16 bytes {protein id, count, lowest position, -}: one memory line per table add.  As three arrays an add
// touched three random lines of a table that is megabytes large, and the tier ran at the speed of those line fills and
// write-backs (14 ms per batch on the skewed database, 53 GB of traffic).
struct GlobalTable {
    uint32_t *slots;  // 4 words per slot
    uint32_t *nd;     // nd lives in LDS
    uint32_t log2cap;
    // One workgroup owns the table (its region comes from a cursor, once per batch), so the adds are atomics at
    // WORKGROUP scope: performed in the L2 of the XCD the workgroup runs on, where the plain 16-byte stores of the
    // initialisation went (the L1 is write-through and holds no line of the table: nothing ever loads one before the
    // final read-out).
    __device__ __forceinline__ bool add_n(uint32_t pid, uint32_t pos, uint32_t n, uint32_t &nnew) const
    {
        const uint32_t mask = (1u << log2cap) - 1u;
        uint32_t h = (pid * 0x9E3779B1u) >> (32 - log2cap);
        for (uint32_t t = 0; t <= mask; t++) {
            uint32_t *sl = slots + 4ull * h;
            uint32_t k = KH_EMPTY_PID;  // expected; receives what the slot holds when it is not empty
            if (__hip_atomic_compare_exchange_strong(&sl[0], &k, pid, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                nnew++;
                k = pid;
            }
            if (k == pid) {
                __hip_atomic_fetch_add(&sl[1], n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_min(&sl[2], pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                return true;
            }
            h = (h + 1u) & mask;
        }
        return false;
    }
    __device__ __forceinline__ bool over_limit() const { return false; }
};

struct PostCtr {
    uint32_t post, lists, lids;
    __device__ void clear() { post = lists = lids = 0; }
};

// Adjacent positions of a query mostly hit the same proteins (a homologous stretch), so
// lane i and lane i+1 usually carry the same id in the same head slot.  Sixty-four lanes
// adding to one LDS word serialise; instead the first lane of every run of equal ids adds
// the whole run at once (run length from a ballot of the change points).
template <class Table>
__device__ __forceinline__ bool add_runs(const Table &tab, uint32_t x, uint32_t pos, uint32_t &nnew)
{
    const uint32_t lane = lane_id();
    const uint32_t prev = __shfl_up(x, 1, 64);
    const bool change = (lane == 0) || (prev != x);
    const unsigned long long cm = __ballot(change);
    bool ok = true;
    if (change && x != KH_EMPTY_PID) {
        const unsigned long long above = (lane == 63) ? 0ull : (cm >> (lane + 1));
        const uint32_t len = above ? (uint32_t)__ffsll((long long)above) : 64u - lane;
        ok = tab.add_n(x, pos, len, nnew);
    }
    return ok;
}

// NWIN 64-position windows of one query, processed by one wave with all their memory
// round trips overlapped: (1) the probe results of every window, (2) the 16-byte heads of
// every postings list (count + first three ids), (3) the remaining ids of all lists,
// spread evenly over the 64 lanes whatever the individual list lengths are (per-window
// prefix sum of the leftovers, owner found by binary search in LDS), then the counter
// increments (KCombStore.Get + the id loop of search.go:427-436).  Window k starts at
// c0 + k*stride.  `s_pref` is 64 words of LDS private to the wave.  No workgroup
// barriers inside.  COUNT_ONLY: only sum the postings (G tier sizing pass).
// G tier: postings lists of more than LONG_LIST ids are not expanded by the wave that meets them: they go on a list in
// LDS and the whole workgroup expands them afterwards, FLAT -- item t of all the lists together belongs to the list found
// by binary search in the prefix of their lengths -- a thread per id.  One wave walking a 9 000-protein list 64 ids at a
// time, for each of the 21 positions of a shared motif, is 3 000 dependent round trips while fifteen waves wait.
#define LONG_LIST 8u
#define LONG_SINK_CAP 960u   /* 15 entries per lane; with 1024 the kernel's LDS was 56 bytes over half a CU's: one workgroup per CU */
#define LONG_SINK_LIFT 512u  /* first step of the binary lifting over the prefix: the largest power of two below the capacity */
struct LongSink {
    uint32_t n;
    uint32_t off[LONG_SINK_CAP], pos[LONG_SINK_CAP], cnt[LONG_SINK_CAP];  // cnt becomes the exclusive prefix
    uint32_t total;
};

template <class Table, int NWIN, bool COUNT_ONLY>
__device__ __forceinline__ bool count_windows(const CountParams &p, const uint32_t *vals,
                                              int32_t size, int32_t c0, int32_t stride, const Table &tab, PostCtr &c,
                                              volatile uint32_t *s_pref, LongSink *sink = nullptr)
{
    const uint32_t lane = lane_id();
    uint32_t v[NWIN];
    uint4 h[NWIN];
#pragma unroll
    for (int k = 0; k < NWIN; k++) {
        const int32_t pos = c0 + k * stride + (int32_t)lane;
        v[k] = 0u;
        h[k] = make_uint4(0, 0, 0, 0);
        if (pos < size) {
            v[k] = vals[pos];
            if (v[k] & KH_INLINE_BIT) h[k] = make_uint4(1u, v[k] & ~KH_INLINE_BIT, 0, 0);
            else if (v[k] != 0u) h[k] = reinterpret_cast<const uint4 *>(p.arena)[v[k]];  // {count, id0, id1, id2}
        }
        if (!COUNT_ONLY && sink && h[k].x > LONG_LIST) {
            const uint32_t slot = atomicAdd(&sink->n, 1u);
            if (slot < LONG_SINK_CAP) {  // (a full list: the wave expands it itself, below)
                sink->off[slot] = v[k]; sink->pos[slot] = (uint32_t)pos; sink->cnt[slot] = h[k].x;
                c.post += h[k].x; c.lists++; c.lids += h[k].x;
                h[k] = make_uint4(0, 0, 0, 0);
                v[k] = 0u;
            }
        }
    }
    bool ok = true;
    uint32_t nnew = 0;
    // leftovers (ids beyond the three that came with the head): all their loads are issued
    // before anything is consumed
    constexpr int XIT = 2;  // leftover rounds kept in registers per window (64 ids each)
    uint32_t xid[NWIN][XIT], xpos[NWIN][XIT];
    uint32_t xtotal[NWIN];
    if (!COUNT_ONLY) {
#pragma unroll
        for (int k = 0; k < NWIN; k++) {
            const uint32_t lcnt = h[k].x;
            const uint32_t extra = lcnt > 3u ? lcnt - 3u : 0u;
            uint32_t inc = extra;  // inclusive prefix over the wave
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t t = __shfl_up(inc, o, 64);
                if ((int)lane >= o) inc += t;
            }
            xtotal[k] = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
            s_pref[k * 64 + lane] = inc - extra;  // exclusive prefix
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < NWIN; k++) {
            const volatile uint32_t *pref = s_pref + k * 64;
#pragma unroll
            for (int it = 0; it < XIT; it++) {
                const uint32_t t = (uint32_t)it * 64u + lane;
                const bool act = t < xtotal[k];
                uint32_t lo = 0;
                if (act) {  // largest lane with pref[lane] <= t owns leftover t
#pragma unroll
                    for (int sft = 32; sft > 0; sft >>= 1)
                        if (pref[lo + sft] <= t) lo += sft;
                }
                // executed by all lanes: the owner may be a lane that is idle in this round
                const uint32_t off = __shfl(v[k], (int)lo, 64);
                xid[k][it] = KH_EMPTY_PID;
                xpos[k][it] = (uint32_t)(c0 + k * stride) + lo;
                if (act) xid[k][it] = p.arena[(uint64_t)off * 4 + 4 + (t - pref[lo])];
            }
        }
        // very long lists (more than XIT*64 leftovers in one window): counted as they arrive
#pragma unroll
        for (int k = 0; k < NWIN; k++) {
            const volatile uint32_t *pref = s_pref + k * 64;
            for (uint32_t t0 = XIT * 64u; t0 < xtotal[k]; t0 += 64) {
                const uint32_t t = t0 + lane;
                const bool act = t < xtotal[k];
                uint32_t lo = 0;
                if (act) {
#pragma unroll
                    for (int sft = 32; sft > 0; sft >>= 1)
                        if (pref[lo + sft] <= t) lo += sft;
                }
                const uint32_t off = __shfl(v[k], (int)lo, 64);
                if (act) ok = ok && tab.add_n(p.arena[(uint64_t)off * 4 + 4 + (t - pref[lo])], (uint32_t)(c0 + k * stride) + lo, 1u, nnew);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    // heads: inline ids and the first three ids of every list
#pragma unroll
    for (int k = 0; k < NWIN; k++) {
        const uint32_t pos = (uint32_t)(c0 + k * stride) + lane;
        const uint32_t lcnt = h[k].x;
        if (lcnt != 0u) {
            c.post += lcnt;
            if (!(v[k] & KH_INLINE_BIT)) { c.lists++; c.lids += lcnt; }
        }
        if (!COUNT_ONLY) {  // all lanes take part: run heads add for their whole run
            ok = add_runs(tab, lcnt > 0 ? h[k].y : KH_EMPTY_PID, pos, nnew) && ok;
            ok = add_runs(tab, lcnt > 1 ? h[k].z : KH_EMPTY_PID, pos, nnew) && ok;
            ok = add_runs(tab, lcnt > 2 ? h[k].w : KH_EMPTY_PID, pos, nnew) && ok;
        }
    }
    if (!COUNT_ONLY) {
#pragma unroll
        for (int k = 0; k < NWIN; k++)
#pragma unroll
            for (int it = 0; it < XIT; it++)
                if (xid[k][it] != KH_EMPTY_PID) ok = ok && tab.add_n(xid[k][it], xpos[k][it], 1u, nnew);
        const uint32_t wave_new = wave_total(nnew);
        if (lane == 0 && wave_new) atomicAdd(tab.nd, wave_new);
    }
    return __all(ok);
}

// sets bits [b, e) of the not-a-k-mer-start bitmap
__device__ __forceinline__ void mark_invalid_range(unsigned long long *invalid, uint64_t b, uint64_t e)
{
    // sets bits [b, e)
    while (b < e) {
        const uint64_t w = b >> 6;
        const uint64_t hi = ((w + 1) << 6) < e ? ((w + 1) << 6) : e;
        const unsigned nb = (unsigned)(hi - b);
        const unsigned long long m = (nb == 64 ? ~0ull : ((1ull << nb) - 1ull)) << (b & 63);
        atomicOr(&invalid[w], m);
        b = hi;
    }
}

#include "count_group.hip.inc"
#include "count_async.hip.inc"
#define PK_ARENA_PROT 1600u
#ifndef PK_ARENA_ORF
#define PK_ARENA_ORF 640u
#endif
#define PK_ARENA_ORF_LONG 1024u  /* batches of longer sequences (mean > 200 nt): ORFs of a few hundred residues are common */
#include "count_pack.hip.inc"

// ---- G tier: counting table in HBM, sized from the query's exact postings count -------------------
struct NullTable {
    uint32_t *nd;
    __device__ __forceinline__ bool add_n(uint32_t, uint32_t, uint32_t, uint32_t &) const { return true; }
    __device__ __forceinline__ bool over_limit() const { return false; }
};

// Sums the counter replicas, publishes status, and leaves every piece of per-batch device
// state zeroed for the next batch (no memsets in the steady state).  One workgroup.
// IN_FLIGHT: called by the last workgroup of a running kernel.  Everything read here was written
// with device-scope atomics (performed at the memory side), so device-scope atomic loads see
// it without an agent-scope fence -- which on this part writes back the whole L2 (a fence per
// workgroup made the kernel 30x slower).
template <bool IN_FLIGHT>
__device__ __forceinline__ void finalize_body(unsigned long long *replicas, kaamer_counters *out, uint32_t *small_state,
                                              uint32_t *status_out, unsigned long long *cursors, uint32_t *slot_scale = nullptr,
                                              uint32_t scale_cap = 16u, uint32_t margin_q4 = 30u)
{
    // 256 threads: lane <-> replica, wave w sums counters w, w+4, ...
    static_assert(CTR_REPLICAS == 64, "one replica per lane");
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    for (uint32_t c = wv; c < CTR_N; c += blockDim.x >> 6) {
        unsigned long long *w = &replicas[(size_t)lane * CTR_N + c];
        unsigned long long s = IN_FLIGHT ? __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *w;
        *w = 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) ((unsigned long long *)out)[c] = s;
    }
    unsigned long long g_hits = 0, g_mon_hits = 0, g_monsters = 0;
    if (threadIdx.x == 0) {
        *status_out = IN_FLIGHT ? __hip_atomic_load(&small_state[SLOT_STATUS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : small_state[SLOT_STATUS];
        g_hits = 16ull * (IN_FLIGHT ? __hip_atomic_load(&small_state[SLOT_G_HITS16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : small_state[SLOT_G_HITS16]);
        g_mon_hits = 16ull * (IN_FLIGHT ? __hip_atomic_load(&small_state[SLOT_G_MON_HITS16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : small_state[SLOT_G_MON_HITS16]);
        g_monsters = IN_FLIGHT ? __hip_atomic_load(&small_state[SLOT_G_MONSTERS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : small_state[SLOT_G_MONSTERS];
    }
    __syncthreads();
    if (threadIdx.x < N_SMALL_SLOTS) small_state[threadIdx.x] = 0;
    if (threadIdx.x < 3) cursors[threadIdx.x * CURSOR_STRIDE] = 0;  // G-tier tail cursor, G arena cursors (count, positions)
    // the next batch's counting tables: 1.5 x SizeInKmer x (distinct proteins per k-mer of THIS batch x the margin), never
    // below 1.5 x SizeInKmer, never above scale_cap (8); the prep kernel of the next batch cuts it to what that batch's
    // positions leave room for in the hit arrays (fit_table_scale)
    if (slot_scale && threadIdx.x == 0) {
        // (written by this workgroup, above).  Queries that left their LDS table are left out on both sides: on a skewed
        // database a few monsters with tens of thousands of hits say nothing about the tables of the others, and tables
        // 1.3 x larger for everybody cost the group kernel more than the few G-tier queries they save (measured: + 13 %
        // per batch on --db zipf).  Unless MANY queries left a table that a larger one would have held (more than an
        // eighth of the batch, the monsters -- more distinct hits than the largest LDS table holds -- not counted): reads
        // whose one coding frame finds all the hits, a dense database at scale 1.  Then only the monsters are left out
        const unsigned long long nq = out->n_queries;
        const unsigned long long n_ovf = out->n_overflow < nq ? out->n_overflow : nq;
        const unsigned long long n_mon = g_monsters < n_ovf ? g_monsters : n_ovf;
        const bool widen = 8ull * (n_ovf - n_mon) > nq;
        const unsigned long long ovf = widen ? n_mon : n_ovf;
        const unsigned long long left_out = widen ? g_mon_hits : g_hits;
        const unsigned long long hits = out->n_hits > left_out ? out->n_hits - left_out : 0ull;
        const unsigned long long lookups = nq ? out->n_lookup / nq * (nq - ovf) + out->n_lookup % nq * (nq - ovf) / nq : 0ull;
        if (lookups) {
            unsigned long long t = (hits * margin_q4 + lookups - 1ull) / lookups;   // sixteenths: 16 x hits / lookups x margin / 16
            if (t < 24ull) t = 16ull;   // (below 1.5 the tables' own slack covers it; measured on the skewed database: a scale
                                        // of ~1.3 cost the group kernel 7 % and saved a tenth of the G-tier queries)
            if (t > scale_cap) t = scale_cap;
            *slot_scale = (uint32_t)t;
        }
    }
}

#ifndef G_MIN_BLOCKS
#define G_MIN_BLOCKS 1
#endif
#ifndef G_NWIN
#define G_NWIN 2   /* windows of 64 positions a wave has in flight in the G tier's sweeps */
#endif
#ifdef KAAMER_PHASE_CLOCK
__device__ unsigned long long g_gtier_clock[2048][16];
#endif
__global__ __launch_bounds__(64 * G_WAVES, G_MIN_BLOCKS) void count_global_kernel(CountParams p)
{
    constexpr int WAVES = G_WAVES;
    __shared__ uint32_t s_nd, s_fail, s_cursor;
    constexpr int NWIN = G_NWIN;
    __shared__ uint32_t s_pref[WAVES][NWIN * 64];
    __shared__ unsigned long long s_post, s_off, s_base, s_reserved;
    __shared__ uint32_t s_roff[G_MAX_RANGES], s_rcur[G_MAX_RANGES], s_rtotal;  // buckets of a partitioned query
    __shared__ LongSink s_long;
    __shared__ uint32_t b_keys[BigLdsTable::CAP], b_val[BigLdsTable::CAP];

    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const uint32_t n_items = *p.list_count < p.list_cap ? *p.list_count : p.list_cap;
    // the grid is sized for a batch full of overflowing queries; most batches have none, and the workgroups beyond the
    // items (all but one when there are none) leave at once -- they take no part in the completion count either
    const uint32_t n_part = n_items < gridDim.x ? (n_items ? n_items : 1u) : gridDim.x;
    if (blockIdx.x >= n_part) return;
    unsigned long long tot_hits = 0, f_post = 0, f_lists = 0, f_lids = 0, mon_hits = 0;
    uint32_t mon_q = 0;
#ifdef KAAMER_PHASE_CLOCK   /* tools/phase_clock_gtier.py: where the G tier's time goes, per workgroup (thread 0) */
    unsigned long long gq_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long gq_start = wall_clock64();
#endif
    PostCtr pc;
    NullTable nt;
    nt.nd = &s_nd;
    // the long postings lists a sweep set aside, expanded by all threads into `tab`
    auto expand_long = [&](const auto &tab) {
        const uint32_t nl = s_long.n < LONG_SINK_CAP ? s_long.n : LONG_SINK_CAP;
        if (wv == 0 && nl) {  // lengths -> exclusive prefix (16 entries per lane)
            constexpr uint32_t PER = LONG_SINK_CAP / 64;
            uint32_t sum = 0;
            for (uint32_t i = 0; i < PER; i++) { const uint32_t e = lane * PER + i; sum += e < nl ? s_long.cnt[e] : 0u; }
            uint32_t inc = sum;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t t = __shfl_up(inc, o, 64);
                if ((int)lane >= o) inc += t;
            }
            uint32_t run = inc - sum;
            for (uint32_t i = 0; i < PER; i++) {
                const uint32_t e = lane * PER + i;
                if (e < nl) { const uint32_t c = s_long.cnt[e]; s_long.cnt[e] = run; run += c; }
            }
            if (lane == 63) s_long.total = inc;
        }
        __syncthreads();
        uint32_t nnew = 0;
        bool ok = true;
        const uint32_t total = nl ? s_long.total : 0u;
        // Eight items per thread and round, their arena loads all issued before the first id is used, and the list of an
        // item found from the previous item's (the next item of a thread is 512 further on: the same list, or the one
        // after it) instead of a ten-step search in LDS per id.  One dependent (search -> load -> add) chain per id made
        // a monster query's sweep ~1 500 cycles per id and thread: 0.6 ms for half a million postings, three sweeps per
        // query, and the tier ended when the slowest workgroup had done two of them (4.5 ms per skewed batch).
#ifndef G_EXPAND_UNROLL
#define G_EXPAND_UNROLL 8  /* measured on the skewed batch: 4 -> 5.28 ms, 8 -> 4.36, 16 -> 4.42, 32 -> 4.55 */
#endif
        constexpr int EU = G_EXPAND_UNROLL;
        uint32_t e = 0;  // largest e with prefix[e] <= the thread's current item
        auto find_list = [&](uint32_t t) {
            if (e + 1u < nl && s_long.cnt[e + 1u] <= t) {      // not in the current list any more
                e++;
                if (e + 1u < nl && s_long.cnt[e + 1u] <= t) {  // nor in the next one: search
                    e = 0;
#pragma unroll
                    for (uint32_t sft = LONG_SINK_LIFT; sft > 0; sft >>= 1)
                        if (e + sft < nl && s_long.cnt[e + sft] <= t) e += sft;
                }
            }
        };
        for (uint32_t t0 = tid; t0 < total; t0 += EU * 64 * WAVES) {
            if (s_fail) break;  // (the table gave up: no point in the rest)
            uint32_t ids[EU], pos_[EU];
#pragma unroll
            for (int u = 0; u < EU; u++) {
                const uint32_t t = t0 + (uint32_t)u * 64 * WAVES;
                const uint32_t tt = t < total ? t : total - 1u;  // (clamped: the load is issued whatever; used only if t < total)
                find_list(tt);
                ids[u] = p.arena[(uint64_t)s_long.off[e] * 4 + 1 + (tt - s_long.cnt[e])];
                pos_[u] = s_long.pos[e];
            }
#pragma unroll
            for (int u = 0; u < EU; u++)
                if (t0 + (uint32_t)u * 64 * WAVES < total)
                    if (!tab.add_n(ids[u], pos_[u], 1u, nnew)) { ok = false; s_fail = 1; }
        }
        const uint32_t wn = wave_total(nnew);
        if (lane == 0 && wn) atomicAdd(&s_nd, wn);
        if (!ok) s_fail = 1;
    };

    // items are handed out by ticket (after the first one): a query here costs anything from microseconds to
    // milliseconds, and two expensive ones on the same workgroup's static share were the tier's tail
    __shared__ uint32_t s_item;
    for (uint32_t item = blockIdx.x; item < n_items;) {
        const WorkItem wi = p.list[item];
        const uint32_t q = wi.q;
        // a negative size: the query was sent here unseen (its table does not fit a wave's arena, count_pack.hip.inc), so
        // its postings have not been counted anywhere yet; otherwise it overflowed its table and they have
        const bool fresh = wi.size < 0;
        const int32_t size = fresh ? -wi.size : wi.size;
        const uint32_t *vals = p.vals + wi.aa_off;
        if (tid == 0) {
            s_nd = 0; s_fail = 0; s_post = 0; s_cursor = 0; s_long.n = 0;
            s_item = atomicAdd(p.queue_head + (SLOT_G_TICKET - SLOT_QUEUE_HEAD), 1u) + n_part;  // the next item of this workgroup
        }
        __syncthreads();
        item = s_item;
#ifdef KAAMER_PHASE_CLOCK
        unsigned long long gq_t = wall_clock64();
        const unsigned long long gq_t0 = gq_t;
#define GQ_LAP(i) { if (tid == 0) { const unsigned long long t_ = wall_clock64(); gq_acc[i] += t_ - gq_t; gq_t = t_; } }
#else
#define GQ_LAP(i) { }
#endif
        // pass 1: exact number of postings (an upper bound of the distinct proteins)
        pc.clear();
        for (int32_t r0 = 0; r0 < size; r0 += 64 * WAVES * NWIN)
            count_windows<NullTable, NWIN, true>(p, vals, size, r0 + 64 * (int32_t)wv, 64 * WAVES, nt, pc, s_pref[wv]);
        {
            const unsigned long long wp = wave_total(pc.post);
            if (lane == 0 && wp) atomicAdd(&s_post, wp);
            if (fresh) {
                const uint32_t wl = wave_total(pc.lists), wi_ = wave_total(pc.lids);
                if (lane == 0) { f_post += wp; f_lists += wl; f_lids += wi_; }
            }
        }
        __syncthreads();
        // ---- in LDS: the whole query in the big table first.  A query with more distinct hits than it holds (a skewed
        // database: tens of thousands of proteins behind a few motifs) has its postings PARTITIONED by protein-id range
        // (a hash of the id) into R buckets in the G arena -- one counting sweep for the bucket sizes, one sweep that
        // scatters (id, position, run length) -- and then every bucket is counted in the LDS table on its own and its
        // hits appended to the query's list (space for min(postings, proteins) entries is taken once).  Three sweeps of
        // the query and sequential bucket reads instead of one random line fill and write-back per posting in a
        // multi-megabyte table in HBM (6.5 of the skewed batch's 8.3 ms); counting the ranges by re-sweeping the whole
        // query per range (tried first) costs R sweeps and lost for the monsters (profiles/r03_count_kernels.md).
        if (size < 65535) {
            BigLdsTable bt;
            bt.keys = b_keys; bt.val = b_val; bt.nd = &s_nd;
            bool q_done = false, give_up = false;
            uint32_t written = 0;
            if (tid == 0) s_base = ~0ull;
            // counts bucket `r` (or, r == R: everything, straight from the postings) in the LDS table and appends its hits
            auto count_bucket = [&](uint32_t R, uint32_t r, const uint2 *items) -> int {  // 0 ok, 1 table overflow, 2 no space
                __syncthreads();
                if (tid == 0) { s_nd = 0; s_fail = 0; s_cursor = 0; s_long.n = 0; }
                for (uint32_t i = tid; i < BigLdsTable::CAP; i += 64 * WAVES) { b_keys[i] = KH_EMPTY_PID; b_val[i] = 0xFFFF0000u; }
                __syncthreads();
                if (R == 1u) {
                    pc.clear();
                    for (int32_t r0 = 0; r0 < size && !s_fail; r0 += 64 * WAVES * NWIN) {
                        const bool ok = count_windows<BigLdsTable, NWIN, false>(p, vals, size, r0 + 64 * (int32_t)wv, 64 * WAVES, bt, pc,
                                                                               s_pref[wv], &s_long);
                        if (!ok) s_fail = 1;
                    }
                    __syncthreads();
                    expand_long(bt);
                } else {
                    uint32_t nnew = 0;
                    const uint32_t b0 = s_roff[r], b1 = s_rcur[r];   // (the scatter sweep left every cursor at its bucket's end)
                    for (uint32_t i = b0 + tid; i < b1; i += 64 * WAVES) {
                        if (s_fail) break;
                        const uint2 it = items[i];
                        if (!bt.add_n(it.x, it.y & 0xFFFFu, it.y >> 16, nnew)) s_fail = 1;
                    }
                }
                __syncthreads();
                const uint32_t total = s_nd;
                if (s_fail != 0u || total > BigLdsTable::LIMIT) return 1;
                if (s_base == ~0ull && total > 0) {  // (workgroup-uniform: read after a barrier)
                    __syncthreads();
                    if (tid == 0) {
                        unsigned long long want = total;  // one bucket: exactly the hits; more: the bound
                        if (R > 1u) want = s_post < p.n_proteins ? s_post : p.n_proteins;
                        s_base = tail_alloc(p, (uint32_t)(want > 0xFFFFFFF0ull ? 0xFFFFFFF0ull : want));
                        s_reserved = want;
                        if (s_base == ~0ull) s_fail = 2;
                    }
                    __syncthreads();
                    if (s_fail == 2u) return 2;
                }
                const unsigned long long base = s_base;
                if (total > 0) {
                    if ((unsigned long long)written + total > s_reserved) { if (tid == 0) atomicOr(p.status, (uint32_t)ST_POOL_FULL); return 2; }
                    for (uint32_t i0 = wv * 64u; i0 < BigLdsTable::CAP; i0 += 64 * WAVES) {
                        const uint32_t k = b_keys[i0 + lane];
                        const bool has = k != KH_EMPTY_PID;
                        const unsigned long long bm = __ballot(has);
                        uint32_t wbase = 0;
                        if (lane == 0 && bm) wbase = atomicAdd(&s_cursor, (uint32_t)__popcll(bm));
                        wbase = __shfl(wbase, 0, 64);
                        if (has) {
                            const unsigned long long idx = base + written + wbase + (uint32_t)__popcll(bm & ((1ull << lane) - 1ull));
                            const uint32_t v = b_val[i0 + lane];
                            p.hit_pid[idx] = k;
                            p.hit_km[idx] = v & 0xFFFFu;
                            p.hit_fp[idx] = v >> 16;
                        }
                    }
                    written += total;
                }
                return 0;
            };
            GQ_LAP(0);   // pass 1
            int rc1 = count_bucket(1u, 0u, nullptr);
            GQ_LAP(1);   // the whole query in the LDS table
            if (rc1 == 0 || rc1 == 2) q_done = true;
            if (rc1 == 1) {
                // ---- partition: bucket sizes, offsets, scatter.  The sizes are GUESSED first: the ranges are a hash of the
                // protein id, so every bucket gets postings / R items give or take a few per cent, and buckets of 5/4 of that
                // (+ 2 048) spare the sweep that counts them -- one of a monster's three sweeps over hundreds of thousands
                // of postings.  A bucket that is full after all (a few ids with very many positions in one range) sends the
                // query through the exact sizes (attempt 1).
                uint32_t R = (uint32_t)(s_post / 16384ull) + 2u;  // ~3 000 distinct ids per bucket when ids repeat ~6 times
                if (R > G_MAX_RANGES) R = G_MAX_RANGES;
                for (int attempt = 0; attempt < 2 && !q_done; attempt++) {
                    uint32_t cap_items = 0xFFFFFFFFu;
                    __syncthreads();
                    if (tid == 0) { s_fail = 0; s_long.n = 0; s_cursor = 0; }   // (s_cursor: "a guessed bucket was full"; count_bucket resets it)
                    if (attempt == 0) {
                        const unsigned long long per = s_post / R;
                        const unsigned long long c = per + per / 4ull + 2048ull;
                        if (c * R > 0x7FFFFFFFull) continue;   // (workgroup-uniform) too many postings to guess: count them
                        cap_items = (uint32_t)c;
                        if (tid < G_MAX_RANGES) { s_roff[tid] = tid * cap_items; s_rcur[tid] = tid * cap_items; }
                        if (tid == 0) s_rtotal = R * cap_items;
                        __syncthreads();
                    } else {
                        if (tid < G_MAX_RANGES) s_roff[tid] = 0;
                        __syncthreads();
                        RangeHist rh;
                        rh.cnt = s_roff; rh.nd = &s_nd; rh.R = R;
                        pc.clear();
                        for (int32_t r0 = 0; r0 < size; r0 += 64 * WAVES * NWIN)
                            count_windows<RangeHist, NWIN, false>(p, vals, size, r0 + 64 * (int32_t)wv, 64 * WAVES, rh, pc, s_pref[wv], &s_long);
                        __syncthreads();
                        expand_long(rh);
                        __syncthreads();
                        if (wv == 0) {  // exclusive prefix of the bucket sizes (R <= 64: one lane each); cursors start at the offsets
                            const uint32_t v = lane < R ? s_roff[lane] : 0u;
                            uint32_t inc = v;
#pragma unroll
                            for (int o = 1; o < 64; o <<= 1) {
                                const uint32_t t = __shfl_up(inc, o, 64);
                                if ((int)lane >= o) inc += t;
                            }
                            s_roff[lane] = inc - v;
                            s_rcur[lane] = inc - v;
                            if (lane == 63) s_rtotal = inc;
                        }
                        __syncthreads();
                    }
                    const uint32_t n_items = s_rtotal;
                    if (tid == 0) {  // 8 bytes per item: half a 16-byte slot of the arena
                        const unsigned long long need = ((unsigned long long)n_items + 1ull) / 2ull + 1ull;
                        const unsigned long long off = atomicAdd(p.g_cursor, need);
                        if (off + need > p.g_slots) { atomicOr(p.status, (uint32_t)ST_G_ARENA_FULL); s_off = ~0ull; }
                        else s_off = off;
                        s_long.n = 0;
                    }
                    __syncthreads();
                    if (s_off == ~0ull) { give_up = true; q_done = true; break; }
                    uint2 *items = reinterpret_cast<uint2 *>(p.g_keys + 4ull * s_off);
                    RangeScatter rsc;
                    rsc.cur = s_rcur; rsc.items = items; rsc.nd = &s_nd; rsc.R = R;
                    rsc.first = s_roff; rsc.cap = cap_items; rsc.ovf = &s_cursor;
                    pc.clear();
                    for (int32_t r0 = 0; r0 < size; r0 += 64 * WAVES * NWIN)
                        count_windows<RangeScatter, NWIN, false>(p, vals, size, r0 + 64 * (int32_t)wv, 64 * WAVES, rsc, pc, s_pref[wv], &s_long);
                    __syncthreads();
                    expand_long(rsc);
                    // the buckets were written by this workgroup's own stores into a region nobody has loaded from (g_cursor
                    // never hands a region out twice within a launch): after the barrier (which drains vmcnt) the loads
                    // below miss the L1 and find them in the L2
                    __syncthreads();
                    GQ_LAP(2);   // partition sweeps
                    if (s_cursor != 0u) continue;   // (workgroup-uniform: read after a barrier) a guessed bucket was too small
                    bool overflow = false;
                    for (uint32_t r = 0; r < R && !q_done; r++) {
                        const int rc = count_bucket(R, r, items);
                        if (rc == 1) { overflow = true; break; }
                        if (rc == 2) { q_done = true; give_up = true; }
                    }
                    GQ_LAP(3);   // buckets
#ifdef KAAMER_PHASE_CLOCK
                    if (tid == 0) { gq_acc[6] += 1; gq_acc[7] += s_post; }
#endif
                    if (!overflow) q_done = true;  // else: a bucket with more distinct ids than the table holds -> the table in HBM
                    else { written = 0; }
                    break;
                }
            }
            __syncthreads();
            if (q_done) {
                const unsigned long long base = give_up ? ~0ull : s_base;
                if (wv == 0 && base != ~0ull) {
                    tot_hits += written;
                    if (written > G_MONSTER_HITS) { mon_hits += written; mon_q++; }
                }
                if (tid == 0) { p.q_cnt[q] = (base != ~0ull) ? written : 0u; p.hit_off[q] = (base == ~0ull || written == 0) ? 0 : base; }
                __syncthreads();
#ifdef KAAMER_PHASE_CLOCK
                if (tid == 0) { const unsigned long long d_ = wall_clock64() - gq_t0; gq_acc[5] += 1; if (d_ > gq_acc[8]) { gq_acc[8] = d_; gq_acc[9] = s_post; } }
#endif
                continue;
            }
            if (tid == 0) { s_nd = 0; s_fail = 0; s_cursor = 0; s_long.n = 0; }  // (a bucket overflowed: the table in HBM)
            __syncthreads();
        }
        unsigned long long bound = s_post;
        if (bound > p.n_proteins) bound = p.n_proteins;
        uint32_t log2cap = 10;
        while ((1ull << log2cap) < 2 * bound && log2cap < 31) log2cap++;
        const unsigned long long cap = 1ull << log2cap;
        if (tid == 0) {
            const unsigned long long off = atomicAdd(p.g_cursor, cap);
            if (off + cap > p.g_slots) { atomicOr(p.status, (uint32_t)ST_G_ARENA_FULL); s_off = ~0ull; }
            else s_off = off;
        }
        __syncthreads();
        const unsigned long long off = s_off;
        if (off == ~0ull) {
            if (tid == 0) { p.q_cnt[q] = 0; p.hit_off[q] = 0; }
            __syncthreads();
            continue;
        }
        GlobalTable gt;
        gt.slots = p.g_keys + 4ull * off; gt.nd = &s_nd; gt.log2cap = log2cap;
        // The table belongs to this workgroup alone (g_cursor never hands out a region twice within one launch) and is
        // touched with plain 16-byte initialisation stores, workgroup-scope atomics and, at the end, plain read-out
        // loads.  Nothing loads from it before the read-out, so no stale line can sit in the L1 (write-through, holds
        // no line of the table); the atomics are performed in the XCD's L2, where the initialisation stores went.  A
        // workgroup barrier (which drains vmcnt) between the phases is enough: no agent-scope fence -- a full L2
        // write-back on this part.
        for (unsigned long long i = tid; i < cap; i += 64 * WAVES)
            reinterpret_cast<uint4 *>(gt.slots)[i] = make_uint4(KH_EMPTY_PID, 0u, 0xFFFFFFFFu, 0u);
        __syncthreads();
        // pass 2: count
        pc.clear();
        for (int32_t r0 = 0; r0 < size; r0 += 64 * WAVES * NWIN) {
            const bool ok = count_windows<GlobalTable, NWIN, false>(p, vals, size, r0 + 64 * (int32_t)wv, 64 * WAVES, gt,
                                                                   pc, s_pref[wv], &s_long);
            if (!ok) s_fail = 1;
        }
        __syncthreads();
        expand_long(gt);
        __syncthreads();
        // lookups, postings and queries were already counted by the group kernel
        const uint32_t total = s_nd;
        const bool failed = s_fail != 0;
        if (tid == 0) {
            if (failed) { atomicOr(p.status, (uint32_t)ST_G_TABLE_FULL); s_base = ~0ull; }
            else if (total == 0) s_base = 0;
            else s_base = tail_alloc(p, total);
        }
        __syncthreads();
        const unsigned long long base = s_base;
        if (base != ~0ull && total > 0) {
            // four stripes of the table per round trip, a 16-byte slot per lane
            for (unsigned long long i0 = (unsigned long long)wv * 64; i0 < cap; i0 += 4ull * 64 * WAVES) {
                uint32_t k4[4], c4[4], m4[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const unsigned long long i = i0 + (unsigned long long)u * 64 * WAVES + lane;
                    const uint4 sv = reinterpret_cast<const uint4 *>(gt.slots)[i < cap ? i : cap - 1];
                    k4[u] = i < cap ? sv.x : KH_EMPTY_PID; c4[u] = sv.y; m4[u] = sv.z;
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const bool has = k4[u] != KH_EMPTY_PID;
                    const unsigned long long bm = __ballot(has);
                    uint32_t wbase = 0;
                    if (lane == 0 && bm) wbase = atomicAdd(&s_cursor, (uint32_t)__popcll(bm));
                    wbase = __shfl(wbase, 0, 64);
                    if (has) {
                        const uint32_t idx = wbase + (uint32_t)__popcll(bm & ((1ull << lane) - 1ull));
                        p.hit_pid[base + idx] = k4[u];
                        p.hit_km[base + idx] = c4[u];
                        p.hit_fp[base + idx] = m4[u];
                    }
                }
            }
            if (wv == 0) {
                tot_hits += total;
                if (total > G_MONSTER_HITS) { mon_hits += total; mon_q++; }
            }
        }
        if (tid == 0) { p.q_cnt[q] = (base != ~0ull) ? total : 0u; p.hit_off[q] = base == ~0ull ? 0 : base; }
        __syncthreads();
    }
#ifdef KAAMER_PHASE_CLOCK
    if (tid == 0 && blockIdx.x < 2048) {
        gq_acc[4] = wall_clock64() - gq_start;
        for (int i = 0; i < 8; i++) g_gtier_clock[blockIdx.x][i] += gq_acc[i];
        if (gq_acc[8] > g_gtier_clock[blockIdx.x][8]) { g_gtier_clock[blockIdx.x][8] = gq_acc[8]; g_gtier_clock[blockIdx.x][9] = gq_acc[9]; }
        if (gq_acc[4] > g_gtier_clock[blockIdx.x][10]) g_gtier_clock[blockIdx.x][10] = gq_acc[4];
    }
#endif
    if (lane == 0) {
        const uint32_t rep = blockIdx.x * WAVES + wv;
        add_counter(p.counters, rep, CTR_HITS, tot_hits);
        // (what the queries that left their LDS tables found, and the monsters among them: finalize_body works the next
        // batch's table scale out without the ones or the others)
        if (tot_hits) atomicAdd(p.queue_head + (SLOT_G_HITS16 - SLOT_QUEUE_HEAD), (uint32_t)((tot_hits + 15ull) >> 4 > 0x0FFFFFFFull ? 0x0FFFFFFFull : (tot_hits + 15ull) >> 4));
        if (mon_q) {
            atomicAdd(p.queue_head + (SLOT_G_MON_HITS16 - SLOT_QUEUE_HEAD), (uint32_t)((mon_hits + 15ull) >> 4 > 0x0FFFFFFFull ? 0x0FFFFFFFull : (mon_hits + 15ull) >> 4));
            atomicAdd(p.queue_head + (SLOT_G_MONSTERS - SLOT_QUEUE_HEAD), mon_q);
        }
        add_counter(p.counters, rep, CTR_POST, f_post);
        add_counter(p.counters, rep, CTR_LISTS, f_lists);
        add_counter(p.counters, rep, CTR_LIST_IDS, f_lids);
    }
    if (p.fin_out) {
        // the batch ends here: the last workgroup to arrive does the finalize step
        __shared__ uint32_t s_last;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // this wave's counter atomics are done
        __syncthreads();
        // two levels: 512 workgroups bumping ONE word serialise at ~90 per microsecond (6 us of a 0.2 ms batch);
        // eight sub-counters (workgroup index mod 8), and the last arrival of each bumps the top one
        if (tid == 0) {
            const uint32_t sub = blockIdx.x & 7u, n_sub = (n_part - sub + 7u) >> 3, n_top = n_part < 8u ? n_part : 8u;
            s_last = 0;
            if (atomicAdd(p.queue_head + (SLOT_QUEUE_SUB - SLOT_QUEUE_HEAD) + sub, 1u) == n_sub - 1u)
                s_last = atomicAdd(p.queue_head, 1u) == n_top - 1u;
        }
        __syncthreads();
        if (s_last) {
            finalize_body<true>(p.counters, p.fin_out, p.fin_small, p.fin_status_out, p.fin_cursors, p.fin_slot_scale, p.fin_scale_cap, p.fin_scale_margin);
        }
    }
}

// PositionHits of the G-tier queries (search.go:442-452): the query's final hit list goes into a
// table in HBM (id -> index in the list), then every (position, id) sets one bit of that hit's bitmap.
struct BitsTable {
    const uint32_t *slots;  // 4 words per slot: {protein id, index in the hit list, -, -}
    uint32_t *nd;
    uint32_t log2cap, words;
    unsigned long long *bits;  // first bitmap of the query
    __device__ __forceinline__ bool add_n(uint32_t pid, uint32_t pos, uint32_t n, uint32_t &) const
    {
        const uint32_t mask = (1u << log2cap) - 1u;
        uint32_t h = (pid * 0x9E3779B1u) >> (32 - log2cap);
        for (uint32_t t = 0; t <= mask; t++) {
            const uint32_t k = __hip_atomic_load(&slots[4ull * h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (k == pid) {
                unsigned long long *bm = bits + (unsigned long long)__hip_atomic_load(&slots[4ull * h + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * words;
                const uint32_t w0 = pos >> 6, b0 = pos & 63u;
                const uint32_t n0 = n < 64u - b0 ? n : 64u - b0;
                atomicOr(&bm[w0], (n0 == 64u ? ~0ull : ((1ull << n0) - 1ull)) << b0);
                if (n > n0) atomicOr(&bm[w0 + 1], (1ull << (n - n0)) - 1ull);
                return true;
            }
            if (k == KH_EMPTY_PID) return false;  // cannot happen: every id of the postings is a hit
            h = (h + 1u) & mask;
        }
        return false;
    }
    __device__ __forceinline__ bool over_limit() const { return false; }
};

__global__ __launch_bounds__(64 * G_WAVES) void positions_global_kernel(CountParams p)
{
    constexpr int WAVES = G_WAVES;
    constexpr int NWIN = 2;
    __shared__ uint32_t s_nd;
    __shared__ uint32_t s_pref[WAVES][NWIN * 64];
    __shared__ unsigned long long s_off;
    const uint32_t tid = threadIdx.x, wv = tid >> 6;
    const uint32_t n_items = *p.list_count < p.list_cap ? *p.list_count : p.list_cap;
    PostCtr pc;
    for (uint32_t item = blockIdx.x; item < n_items; item += gridDim.x) {
        const WorkItem wi = p.list[item];
        const uint32_t q = wi.q, cnt = p.q_cnt[q];
        const int32_t size = wi.size < 0 ? -wi.size : wi.size;  // (negative: sent to the G tier unseen, see count_global_kernel)
        const unsigned long long hoff = p.hit_off[q];
        uint32_t log2cap = 10;
        while ((1ull << log2cap) < 2ull * cnt && log2cap < 31) log2cap++;
        const unsigned long long cap = 1ull << log2cap;
        if (tid == 0) {
            s_nd = 0;
            const unsigned long long off = atomicAdd(p.g_cursor, cap);
            if (off + cap > p.g_slots) { atomicOr(p.status, (uint32_t)ST_G_ARENA_FULL); s_off = ~0ull; }
            else s_off = off;
        }
        __syncthreads();
        const unsigned long long off = s_off;
        if (off == ~0ull || cnt == 0 || size <= 0) { __syncthreads(); continue; }
        uint32_t *slots = p.g_keys + 4ull * off;
        for (unsigned long long i = tid; i < cap; i += 64 * WAVES) __hip_atomic_store(&slots[4 * i], KH_EMPTY_PID, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const uint32_t mask = (uint32_t)cap - 1u;
        for (uint32_t i = tid; i < cnt; i += 64 * WAVES) {
            const uint32_t pid = p.hit_pid[hoff + i];
            uint32_t h = (pid * 0x9E3779B1u) >> (32 - log2cap);
            while (atomicCAS(&slots[4ull * h], KH_EMPTY_PID, pid) != KH_EMPTY_PID) h = (h + 1u) & mask;  // ids are distinct, load <= 0.5
            __hip_atomic_store(&slots[4ull * h + 1], i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        BitsTable bt;
        bt.slots = slots; bt.nd = &s_nd; bt.log2cap = log2cap;
        bt.words = ((uint32_t)size + 63u) >> 6;
        bt.bits = p.pos_bits + p.pos_base[q];
        pc.clear();
        bool ok = true;
        for (int32_t r0 = 0; r0 < size; r0 += 64 * WAVES * NWIN)
            ok = count_windows<BitsTable, NWIN, false>(p, p.vals + wi.aa_off, size, r0 + 64 * (int32_t)wv, 64 * WAVES, bt, pc, s_pref[wv]) && ok;
        if (!ok && (tid & 63u) == 0) atomicOr(p.status, (uint32_t)ST_G_TABLE_FULL);
        __syncthreads();
    }
}

#include "translate.hip.inc"
#include "topn.hip.inc"

// ------------------------------------------------------------------------------------
// exclusive scan of q_cnt[0..nq) -> hit_off[0..nq]
// ------------------------------------------------------------------------------------
#define SCAN_BLOCK 1024
#define SCAN_ITEMS 16
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

// one block walks all tiles with a carry: one launch, for batches of up to ~1e5 queries
__global__ __launch_bounds__(SCAN_BLOCK) void scan_single_kernel(const uint32_t *cnt, const uint32_t *d_nq, uint64_t *hit_off)
{
    const uint32_t nq = *d_nq;
    uint64_t carry = 0;
    for (uint64_t base = 0; base <= nq; base += SCAN_TILE) {
        uint32_t v[SCAN_ITEMS];
        uint64_t s = 0;
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; i++) {
            const uint64_t idx = base + (uint64_t)threadIdx.x * SCAN_ITEMS + i;
            v[i] = idx < nq ? cnt[idx] : 0u;
            s += v[i];
        }
        uint64_t tot;
        uint64_t ex = block_exclusive_scan(s, &tot) + carry;
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; i++) {
            const uint64_t idx = base + (uint64_t)threadIdx.x * SCAN_ITEMS + i;
            if (idx <= nq) hit_off[idx] = ex;
            ex += v[i];
        }
        carry += tot;
        __syncthreads();
    }
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

#include "exchange.hip.inc"

// optional compaction: sharded hit arrays -> CSR in query order (one wave per query)
__global__ __launch_bounds__(256) void gather_hits_kernel(const uint32_t *d_nq, const uint64_t *csr_off, const uint64_t *hit_off,
                                                          const uint32_t *q_cnt, const uint32_t *in_pid, const uint32_t *in_km,
                                                          const uint32_t *in_fp, uint32_t *out_pid, uint32_t *out_km,
                                                          uint32_t *out_fp, uint64_t out_cap, uint32_t *status, int firstpos)
{
    const uint32_t nq = *d_nq;
    const uint64_t n_hits = csr_off[nq];
    if (n_hits > out_cap) { if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(status, (uint32_t)ST_POOL_FULL); return; }
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t q = wave; q < nq; q += n_waves) {
        const uint32_t n = q_cnt[q];
        if (n == 0) continue;
        const uint64_t s = hit_off[q], d = csr_off[q];
        for (uint32_t i = lane; i < n; i += 64) {
            out_pid[d + i] = in_pid[s + i];
            out_km[d + i] = in_km[s + i];
            if (firstpos) out_fp[d + i] = in_fp[s + i];
        }
    }
}

__global__ void finalize_kernel(unsigned long long *replicas, kaamer_counters *out, uint32_t *small_state,
                                uint32_t *status_out, unsigned long long *cursors, uint32_t *slot_scale, uint32_t scale_cap, uint32_t margin_q4)
{
    finalize_body<false>(replicas, out, small_state, status_out, cursors, slot_scale, scale_cap, margin_q4);
}

// ------------------------------------------------------------------------------------
// workspace
// ------------------------------------------------------------------------------------
struct kaamer_workspace {
    int device;
    kaamer_workspace_opts opts;
    uint32_t q_cap;
    uint64_t hit_cap, sparse_cap, g_slots, pos_cap;
    bool compact;                       // finish with CSR in query order (scan + gather pass)
    uint64_t *d_csr_off;                // compact form: CSR offsets
    uint32_t *d_c_pid, *d_c_km, *d_c_fp;
    int g_grid, p_grid, n_cu, pack_grid, pack_grid_long;
    bool use_group;
    bool count_async;                   // protein batches are counted by count_async_kernel instead of count_group_kernel<., 0> (KAAMER_COUNT_ASYNC=1)
    int unit_windows;                   // group windows per unit of count_group_kernel<., 0>: 2, or 4 with one workgroup per CU
    uint32_t pack_shift;                // log2 of the pack window (slots) of a search: 8 for protein, 10 for ORF batches
    // device buffers
    kaamer_query_meta *d_q;
    uint32_t *d_nq;
    unsigned long long *d_n_pos;
    unsigned long long *d_valid;        // one bit per residue position
    uint32_t *d_vals;                   // probe result per residue position
    uint32_t *d_q_cnt;
    unsigned long long *d_pool_cursor;  // CURSOR_STRIDE apart: G-tier tail cursor, G arena cursor (count), G arena cursor (positions)
    WorkItem *d_lists;                  // [N_LISTS][q_cap] (only the G tier's overflow list is used)
    QInfo *d_qinfo;
    uint32_t *d_slots;
    uint64_t *d_slot_off;
    uint32_t *d_group_first;
    uint32_t *d_n_groups;
    uint32_t groups_cap;
    // the counting stage on a stream of its own (kaamer_workspace_set_count_stream): the caller's stream then carries
    // prep + probe only, so that the probe kernel of batch i + 1 starts while batch i is still counting
    hipStream_t count_stream;
    hipEvent_t ev_probe, ev_count;
    bool split_pending;                 // the last search's counting stage is on count_stream: consumers wait for ev_count
    int grp_grid;
    bool sched_identity;                // groups handed out in index order (no longest-first schedule is built)
    // nucleotide / reads input: 6-frame translation products
    bool nucleotide;
    uint64_t aa_cap, sa_cap;
    uint32_t *d_cnt3;                   // [3][6*max_seqs] ORF / aa / starts counts per (sequence, frame)
    uint64_t *d_off3;                   // [3][6*max_seqs+1] their exclusive scans
    uint32_t *d_n6;                     // device scalar 6*n_seqs
    // long sequences (> TS_MAX nt) are translated piece-wise: list, pieces per frame, per-piece counts and their scans
    uint32_t max_long;
    uint64_t max_piece_items;
    uint32_t *d_long_seq, *d_long_np, *d_pcnt3, *d_n_piece_items, *d_piece_li;
    uint64_t *d_piece_base, *d_poff3;
    kaamer_query_meta *d_tmp_meta;
    uint8_t *d_orf_aa;
    int32_t *d_starts_alt;
    uint32_t max_seqs;
    unsigned long long *d_chain;        // layout_kernel: one word per tile, tagged with the batch epoch
    uint2 *d_sched;                     // schedule_kernel: ticket -> (group, first query)
    uint64_t *d_group_start;            // layout_kernel: slot at which the group's first table starts
    unsigned long long *d_lay_total;    // total table slots of the batch
    uint32_t *d_slot_scale;             // table capacity scale of the next batch, sixteenths (finalize_body)
    uint32_t slot_scale_cap;            // ceiling of the scale (8: beyond it the tables leave the pack kernel's arena)
    unsigned long long table_room;      // slots of the hit arrays a batch's tables may take (the G tier's lists come after them)
    uint32_t slot_scale_margin;         // sixteenths: tables of (hits per k-mer) x this
    bool dense_tables;                  // the last finished batch left a table scale of 2 or more: ORF packs take the larger arena
    uint32_t *d_n_sched;
    // post-steps (kaamer_topn_device), allocated on first use
    uint32_t topn_k;
    uint32_t *d_top_cnt, *d_top_pid, *d_top_km, *d_top_fp;
    int32_t *d_top_trim, *d_top_start, *d_top_size;
    bool last_was_merge;
    // exchange step of the sharded index (kaamer_exchange_pack / _merge), allocated on first use
    // exchange scratch: grow-only (the block layout may change from batch to batch within the buffers' capacity)
    size_t x_dst_cap, x_src_cap, x_ent_cap, x_m_cap;
    unsigned long long *d_x_stats;      // {queries, largest block needed, overflow, -} of the last merge's headers
    unsigned long long *h_x_stats;      // pinned: two slots of 4 words, slot = merge sequence number & 1
    hipEvent_t ev_x_stats[2];
    uint64_t x_merge_seq;               // merges enqueued on this workspace
    uint32_t *d_x_dst_off, *d_x_src_off, *d_x_nq_owned, *d_x_pid, *d_x_km, *d_x_fp;
    uint64_t *d_x_ent_off;
    uint64_t *d_x_tiles;                // tile sums of the exchange's tiled scans
    size_t x_tiles_cap;
    // reported-only packing of the top-N results (kaamer_search_batch_top), allocated on first use
    uint32_t rep_k;
    uint32_t *d_rep_flag, *d_rep_aalen, *d_rep_query, *d_rep_pid, *d_rep_km, *d_rep_fp;
    int32_t *d_rep_trim;
    uint64_t *d_rep_rank, *d_rep_eoff, *d_rep_aoff, *d_rep_off;
    kaamer_query_meta *d_rep_q;
    uint8_t *d_rep_aa;
    uint32_t lay_epoch;
    unsigned long long *d_tr_chain;     // translate_reads_kernel: three words per 64-sequence chunk, tagged with tr_epoch
    uint32_t tr_epoch;
    uint32_t *d_list_counts;            // [N_LISTS] + queue head + status (zeroed by finalize)
    uint32_t *d_status_out;             // status of the last finished batch
    bool clean;                         // per-batch device state is known to be zeroed
    bool firstpos;                      // track the lowest matching position per hit
    bool want_positions;                // full PositionHits bitmaps
    uint64_t bits_cap;                  // u64 words of bitmap storage
    uint32_t *d_pos_words;              // per query: hits x words per hit
    uint64_t *d_pos_base;               // exclusive scan of the above
    uint64_t *d_pos_off;                // per hit: first word of its bitmap
    unsigned long long *d_pos_bits;
    uint32_t *d_g_keys;
    unsigned long long *d_counter_replicas;
    kaamer_counters *d_counters;
    uint64_t *d_bsum;
    uint32_t n_scan_blocks;
    uint64_t *d_hit_off;
    uint32_t *d_hit_pid, *d_hit_km, *d_hit_fp;
    std::vector<hipEvent_t> *ev;  // 5 events per timed call: total0, probe0, probe1(=count0), count1, total1
    uint32_t n_timed, time_every, call_no;
};
#define EV_PER_CALL 5

template <class T> static int dev_alloc(T **p, size_t n)
{
    *p = nullptr;
    hipError_t e = hipMalloc((void **)p, (n ? n : 1) * sizeof(T));
    if (e != hipSuccess) return kaamer_fail(KAAMER_E_NOMEM, "hipMalloc(%zu bytes): %s", n * sizeof(T), hipGetErrorString(e));
    return KAAMER_OK;
}

static void launch_group(const CountParams &p, int grid, bool firstpos, hipStream_t s, bool async, int unit_windows)
{
    if (!async && unit_windows == 4) {   // units of four windows: 98 KB of LDS, one workgroup per CU (overlapping batches)
        if (firstpos) hipLaunchKernelGGL((count_group_kernel<true, 0, 4>), dim3(grid), dim3(64 * GRP_WAVES), 0, s, p);
        else hipLaunchKernelGGL((count_group_kernel<false, 0, 4>), dim3(grid), dim3(64 * GRP_WAVES), 0, s, p);
        return;
    }
    if (async) {   // the barrier-free form (count_async.hip.inc); KAAMER_COUNT_ASYNC=0: the round-3 kernel
        if (firstpos) hipLaunchKernelGGL((count_async_kernel<true>), dim3(grid), dim3(64 * GRP_WAVES), 0, s, p);
        else hipLaunchKernelGGL((count_async_kernel<false>), dim3(grid), dim3(64 * GRP_WAVES), 0, s, p);
        return;
    }
    if (firstpos) hipLaunchKernelGGL((count_group_kernel<true, 0>), dim3(grid), dim3(64 * GRP_WAVES), 0, s, p);
    else hipLaunchKernelGGL((count_group_kernel<false, 0>), dim3(grid), dim3(64 * GRP_WAVES), 0, s, p);
}
static void launch_group_positions(const CountParams &p, int grid, hipStream_t s)
{
    hipLaunchKernelGGL((count_group_kernel<false, 1>), dim3(grid), dim3(64 * GRP_WAVES), 0, s, p);
}
static void launch_group_merge(const CountParams &p, int grid, bool firstpos, hipStream_t s)
{
    // without first positions the merge tables need no third array: three workgroups per CU instead of two
    if (firstpos) hipLaunchKernelGGL((count_group_kernel<true, 2>), dim3(grid), dim3(64 * GRP_WAVES), 0, s, p);
    else hipLaunchKernelGGL((count_group_kernel<false, 2>), dim3(grid), dim3(64 * GRP_WAVES), 0, s, p);
}

extern "C" {

const char *kaamer_last_error(void) { return g_err; }
int kaamer_abi_version(void) { return KAAMER_ABI_VERSION; }

#ifdef KAAMER_PHASE_CLOCK
// measurement build only: the phase clocks of count_group_kernel<., 0> since the last reset
// per-workgroup clocks of count_global_kernel: out[b * 16 + i]
int kaamer_debug_gtier_clock(unsigned long long *out, int reset)
{
    HIPCHK(hipDeviceSynchronize());
    static unsigned long long all[2048][16];
    HIPCHK(hipMemcpyFromSymbol(all, HIP_SYMBOL(g_gtier_clock), sizeof all));
    memcpy(out, all, sizeof all);
    if (reset) { memset(all, 0, sizeof all); HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_gtier_clock), all, sizeof all)); }
    return KAAMER_OK;
}
int kaamer_debug_phase_clock(unsigned long long out[16], int reset)
{
    HIPCHK(hipDeviceSynchronize());
    static unsigned long long all[2048][16];
    HIPCHK(hipMemcpyFromSymbol(all, HIP_SYMBOL(g_phase_clock), sizeof all));
    for (int i = 0; i < 16; i++) { out[i] = 0; for (int b = 0; b < 2048; b++) out[i] += all[b][i]; }
    if (reset) { memset(all, 0, sizeof all); HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_phase_clock), all, sizeof all)); }
    return KAAMER_OK;
}
#endif

int kaamer_index_open_image(const kaamer_image *img, int device, kaamer_index **out)
{
    if (!img || !out) return kaamer_fail(KAAMER_E_ARG, "index_open_image: bad argument");
    *out = nullptr;
    HIPCHK(hipSetDevice(device));
    kaamer_index *ix = new (std::nothrow) kaamer_index();
    if (!ix) return kaamer_fail(KAAMER_E_NOMEM, "index alloc");
    ix->device = device;
    ix->n_top = 4;  // calls in flight through the host-buffer boundary
    if (const char *e = getenv("KAAMER_HOST_SLOTS")) { const int v = atoi(e); if (v >= 1 && v <= KAAMER_MAX_HOST_SLOTS) ix->n_top = v; }
    ix->hdr = img->hdr;
    ix->d_buckets = nullptr;
    ix->d_arena = nullptr;
    int rc = dev_alloc(&ix->d_buckets, (size_t)img->hdr.n_buckets);
    if (!rc) rc = dev_alloc(&ix->d_arena, (size_t)(img->hdr.arena_words < 4 ? 4 : img->hdr.arena_words));
    if (rc) { kaamer_index_close(ix); return rc; }
    const kh_bucket *up_buckets = img->buckets;
    const uint32_t *up_arena = img->arena;
    std::vector<kh_bucket> ex_buckets;
    std::vector<uint32_t> ex_arena;
    if (const char *ev = getenv("KAAMER_EXP_ARENA_ORDER")) {
        // EXPERIMENT (tools/r4_arena_order.sh), not a product path: the postings lists re-ordered before the upload, to
        // measure what locality between the lists of one protein's k-mers is worth.  1 = by (first id, old offset)
        struct Ref { uint32_t first_id, off; };
        std::vector<Ref> refs;
        const uint64_t nb = img->hdr.n_buckets;
        for (uint64_t b = 0; b < nb; b++)
            for (int t = 0; t < KH_SLOTS_PER_BUCKET; t++) {
                const kh_slot &sl = img->buckets[b].s[t];
                if (sl.key != KH_EMPTY_KEY && !(sl.val & KH_INLINE_BIT) && sl.val) refs.push_back(Ref{ img->arena[(uint64_t)sl.val * 4 + 1], sl.val });
            }
        std::sort(refs.begin(), refs.end(), [](const Ref &a, const Ref &b) { return a.off < b.off; });
        refs.erase(std::unique(refs.begin(), refs.end(), [](const Ref &a, const Ref &b) { return a.off == b.off; }), refs.end());
        if (atoi(ev) == 1) std::stable_sort(refs.begin(), refs.end(), [](const Ref &a, const Ref &b) { return a.first_id < b.first_id; });
        ex_arena.assign((size_t)img->hdr.arena_words, 0u);
        std::vector<uint32_t> new_off((size_t)(img->hdr.arena_words / 4 + 1), 0u);
        uint32_t cur = 1;   // unit 0 is "no list"
        for (const Ref &r : refs) {
            const uint32_t cnt = img->arena[(uint64_t)r.off * 4];
            const uint32_t units = (1u + cnt + 3u) / 4u;
            memcpy(&ex_arena[(size_t)cur * 4], &img->arena[(size_t)r.off * 4], (size_t)units * 16);
            new_off[r.off] = cur;
            cur += units;
        }
        ex_buckets.assign(img->buckets, img->buckets + nb);
        for (uint64_t b = 0; b < nb; b++)
            for (int t = 0; t < KH_SLOTS_PER_BUCKET; t++) {
                kh_slot &sl = ex_buckets[b].s[t];
                if (sl.key != KH_EMPTY_KEY && !(sl.val & KH_INLINE_BIT) && sl.val) sl.val = new_off[sl.val];
            }
        fprintf(stderr, "[kaamer experiment] arena re-ordered (%zu lists, %u units of %llu)\n", refs.size(), cur, (unsigned long long)(img->hdr.arena_words / 4));
        up_buckets = ex_buckets.data();
        up_arena = ex_arena.data();
    }
    hipError_t e = hipMemcpy(ix->d_buckets, up_buckets, (size_t)img->hdr.n_buckets * sizeof(kh_bucket), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(ix->d_arena, up_arena, (size_t)img->hdr.arena_words * 4, hipMemcpyHostToDevice);
    // list offset 0 is never handed out (kaamer_layout.h) and reads as "no list": the counting kernel loads it for
    // positions without a list instead of branching around the load (count_pack.hip.inc)
    if (e == hipSuccess) e = hipMemset(ix->d_arena, 0, 16);
    if (e != hipSuccess) { kaamer_index_close(ix); return kaamer_fail(KAAMER_E_HIP, "index upload: %s", hipGetErrorString(e)); }
    *out = ix;
    return KAAMER_OK;
}

// makedb -> serving without an image file in between: the table is built on the device it will be searched on
int kaamer_index_build_proteins(const uint8_t *seqs, const uint64_t *offsets, const uint32_t *ids, uint32_t n_proteins,
                                uint32_t shard, uint32_t n_shards, double load_factor, int device, kaamer_index **out)
{
    if (!out || !offsets || (!seqs && n_proteins) || n_shards == 0 || shard >= n_shards)
        return kaamer_fail(KAAMER_E_ARG, "index_build_proteins: bad argument");
    *out = nullptr;
    kaamer_device_image di;
    const int rc = kaamer_build_on_device(seqs, offsets, ids, n_proteins, shard, n_shards, load_factor, device, &di);
    if (rc) return rc;
    kaamer_index *ix = new (std::nothrow) kaamer_index();
    if (!ix) { (void)hipFree(di.d_buckets); (void)hipFree(di.d_arena); return kaamer_fail(KAAMER_E_NOMEM, "index alloc"); }
    ix->device = device;
    ix->n_top = 4;
    if (const char *e = getenv("KAAMER_HOST_SLOTS")) { const int v = atoi(e); if (v >= 1 && v <= KAAMER_MAX_HOST_SLOTS) ix->n_top = v; }
    ix->hdr = di.hdr;
    ix->d_buckets = di.d_buckets;
    ix->d_arena = di.d_arena;   // its first 16 bytes are zero (builder_device.hip), as kaamer_index_open_image leaves them
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
    {   // calls still in flight (a ticket not yet waited for, a caller inside kaamer_search_batch) keep their slot busy:
        // the close waits for them to be given back, then for every slot's stream
        std::unique_lock<std::mutex> lock(ix->pool_mu);
        for (;;) {
            bool busy = false;
            for (const HostSlot &h : ix->host) busy = busy || h.busy;
            for (int i = 0; i < ix->n_top; i++) busy = busy || ix->top[i].busy;
            if (!busy) break;
            ix->pool_cv.wait(lock);
        }
    }
    for (TopSlot &h : ix->top) if (h.stream) (void)hipStreamSynchronize(h.stream);
    for (HostSlot &h : ix->host) {
        if (h.stream) (void)hipStreamSynchronize(h.stream);
        if (h.stream) (void)hipStreamDestroy(h.stream);
        if (h.ws) kaamer_workspace_free(h.ws);
        if (h.d_seqs) (void)hipFree(h.d_seqs);
        if (h.d_off) (void)hipFree(h.d_off);
    }
    for (TopSlot &h : ix->top) top_slot_free(h);
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
    void *bufs[] = { ws->d_q, ws->d_nq, ws->d_n_pos, ws->d_valid, ws->d_vals, ws->d_cnt3, ws->d_off3, ws->d_n6, ws->d_long_seq, ws->d_long_np, ws->d_pcnt3, ws->d_n_piece_items, ws->d_piece_li, ws->d_piece_base, ws->d_poff3,
                     ws->d_tmp_meta, ws->d_orf_aa, ws->d_starts_alt, ws->d_q_cnt, ws->d_csr_off, ws->d_c_pid, ws->d_c_km, ws->d_c_fp,
                     ws->d_pool_cursor, ws->d_lists, ws->d_list_counts, ws->d_status_out, ws->d_qinfo, ws->d_slots,
                     ws->d_slot_off, ws->d_group_first, ws->d_n_groups, ws->d_pos_words, ws->d_pos_base, ws->d_pos_off, ws->d_pos_bits, ws->d_g_keys,
                     ws->d_counter_replicas, ws->d_counters, ws->d_bsum, ws->d_chain, ws->d_tr_chain, ws->d_sched, ws->d_n_sched, ws->d_group_start, ws->d_lay_total, ws->d_slot_scale, ws->d_top_cnt, ws->d_top_pid, ws->d_top_km, ws->d_top_fp, ws->d_top_trim, ws->d_top_start, ws->d_top_size, ws->d_rep_flag, ws->d_rep_aalen, ws->d_rep_query, ws->d_rep_pid, ws->d_rep_km, ws->d_rep_fp, ws->d_rep_trim, ws->d_rep_rank, ws->d_rep_eoff, ws->d_rep_aoff, ws->d_rep_off, ws->d_rep_q, ws->d_rep_aa, ws->d_hit_off,
                     ws->d_hit_pid, ws->d_hit_km, ws->d_hit_fp, ws->d_x_dst_off, ws->d_x_src_off, ws->d_x_nq_owned, ws->d_x_pid, ws->d_x_km, ws->d_x_fp, ws->d_x_ent_off, ws->d_x_tiles, ws->d_x_stats };
    for (void *b : bufs) if (b) (void)hipFree(b);
    if (ws->h_x_stats) {
        (void)hipHostFree(ws->h_x_stats);
        for (hipEvent_t e : ws->ev_x_stats) if (e) (void)hipEventDestroy(e);
    }
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
    ws->nucleotide = opts->seq_type == KAAMER_NUCLEOTIDE || opts->seq_type == KAAMER_READS;
    ws->max_seqs = opts->max_seqs ? opts->max_seqs : 1;
    if (ws->nucleotide) {
        // six frames of len/3 codons: at most 2 amino acids per nucleotide, an ORF needs >= 21 of them
        ws->aa_cap = 2 * opts->max_seq_bytes + 64;
        ws->sa_cap = ws->aa_cap;
        uint64_t qc = opts->max_queries ? opts->max_queries : (uint64_t)ws->max_seqs * 4 + opts->max_seq_bytes / 32 + 64;
        if (qc > 0xFFFFFFF0ull) qc = 0xFFFFFFF0ull;
        ws->q_cap = (uint32_t)qc;
        ws->pos_cap = ws->aa_cap + 64;
    } else {
        // one query per protein record: never fewer slots than sequences (the prep kernel writes one
        // descriptor per sequence)
        ws->q_cap = opts->max_queries > ws->max_seqs ? opts->max_queries : ws->max_seqs;
        ws->pos_cap = opts->max_seq_bytes + 64;
    }
    if (ws->q_cap < 1) ws->q_cap = 1;
    // default bound of the (query, protein) pairs: 256 per query, but never more than one per residue position (an ORF
    // batch has millions of queries of ~40 positions: 256 each sized a 1 M-read workspace at 67 GB of hit arrays)
    ws->hit_cap = opts->max_hits ? opts->max_hits : std::min<uint64_t>((uint64_t)ws->q_cap * 256, ws->pos_cap) + (1u << 20);
    ws->g_slots = opts->g_tier_slots ? opts->g_tier_slots : (32ull << 20);
    {
        // hit arrays: the lists sit at the table layout's offsets (<= 1.5 x positions + 64 per query;
        // for a merge the "positions" are partial entries), G-tier lists after them
        const uint64_t items = ws->pos_cap > ws->hit_cap ? ws->pos_cap : ws->hit_cap;
        ws->sparse_cap = items + items / 2 + 64ull * ws->q_cap + GRP_MAX_TABLE + ws->hit_cap / 2 + (1u << 20);
    }
    // the reference fills PositionHits only for nucleotide/reads input or with -pos (search.go:416)
    ws->firstpos = opts->first_pos == 1 || (opts->first_pos == 0 && (opts->seq_type == KAAMER_NUCLEOTIDE || opts->seq_type == KAAMER_READS));
    int grp_per_cu = 0, p_per_cu = 0;
    // Which counting kernel takes protein batches: count_group_kernel over units of two group windows (count_group.hip.inc).
    // The barrier-free count_async_kernel (count_async.hip.inc; KAAMER_COUNT_ASYNC=1) beat the single-window group kernel
    // when batches overlap (ONE counting workgroup per CU, below: 0.1290-0.1308 against 0.1343-0.1344 ms per batch, same box)
    // and lost alone on the device (115 against 95 us); units then took the group kernel to 0.1268-0.1273 ms with three
    // batches in flight (the barrier-free kernel in the same call: 0.1294-0.1320) and to 89 us alone
    // (profiles/r04_experiments.md).  Both kernels stay under the parity tests (test_both_counting_kernels).
    ws->count_async = false;
    if (const char *e = getenv("KAAMER_COUNT_ASYNC")) ws->count_async = atoi(e) != 0;
    hipError_t oe = ws->count_async
        ? (ws->firstpos ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&grp_per_cu, count_async_kernel<true>, 64 * GRP_WAVES, 0)
                        : hipOccupancyMaxActiveBlocksPerMultiprocessor(&grp_per_cu, count_async_kernel<false>, 64 * GRP_WAVES, 0))
        : (ws->firstpos ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&grp_per_cu, count_group_kernel<true, 0>, 64 * GRP_WAVES, 0)
                        : hipOccupancyMaxActiveBlocksPerMultiprocessor(&grp_per_cu, count_group_kernel<false, 0>, 64 * GRP_WAVES, 0));
    if (oe == hipSuccess)
        oe = ws->nucleotide ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&p_per_cu, probe_kernel<true>, 64 * P_WAVES, 0)
                  : hipOccupancyMaxActiveBlocksPerMultiprocessor(&p_per_cu, probe_kernel<false>, 64 * P_WAVES, 0);
    if (oe != hipSuccess) { delete ws; return kaamer_fail(KAAMER_E_HIP, "occupancy query: %s", hipGetErrorString(oe)); }
    hipDeviceProp_t prop;
    hipError_t pe = hipGetDeviceProperties(&prop, ix->device);
    if (pe != hipSuccess) { delete ws; return kaamer_fail(KAAMER_E_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(pe)); }
    if (grp_per_cu < 1) grp_per_cu = 1;
    if (p_per_cu < 1) p_per_cu = 1;
    // tuning knobs (tools/coresident.sh): workgroups per CU of the counting / probe kernels, never above what fits
    // batches in flight next to this one: ONE counting workgroup per CU (8 waves x 72-80 registers) leaves the CU's other
    // registers and wave slots to the neighbours' probe kernels; three fill the register file and lock them out
    // (tools/r4_sweep.sh: 0.134-0.143 ms per batch against 0.148-0.155 with three batches in flight, 0.26 against 0.20 alone)
    if (opts->concurrent_batches > 1 && !ws->nucleotide) grp_per_cu = 1;
    // (measured with one workgroup per CU and three batches in flight, same box, A/B/A/B: 0.1381-0.1406 ms per batch with the
    // longest-first schedule, 0.1430-0.1441 with the groups in index order: the schedule stays)
    ws->sched_identity = false;
    if (const char *e = getenv("KAAMER_SCHED_IDENTITY")) ws->sched_identity = atoi(e) != 0;
    if (const char *e = getenv("KAAMER_GRP_PER_CU")) { const int v = atoi(e); if (v >= 1 && v <= 3) grp_per_cu = v < grp_per_cu || opts->concurrent_batches > 1 ? v : grp_per_cu; }
    if (const char *e = getenv("KAAMER_P_PER_CU")) { const int v = atoi(e); if (v >= 1 && v < p_per_cu) p_per_cu = v; }
    ws->n_cu = prop.multiProcessorCount;
    ws->grp_grid = ws->n_cu * grp_per_cu;
    // the G tier is rare on a database like DB-SP, but with a skewed database 15 % of the queries overflow their LDS
    // table: enough workgroups to keep the memory system busy (the last one to finish also finalizes the batch)
    {
        int g_per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&g_per_cu, count_global_kernel, 64 * G_WAVES, 0) != hipSuccess || g_per_cu < 1) g_per_cu = 1;
        ws->g_grid = ws->n_cu * (g_per_cu > 2 ? 2 : g_per_cu);
    }
    ws->p_grid = ws->n_cu * p_per_cu;
    {
        int pk_per_cu = 0;
        hipError_t ke = ws->nucleotide ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&pk_per_cu, count_pack_kernel<true, PK_ARENA_ORF>, 64, 0)
                        : ws->firstpos ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&pk_per_cu, count_pack_kernel<true, PK_ARENA_PROT>, 64, 0)
                                       : hipOccupancyMaxActiveBlocksPerMultiprocessor(&pk_per_cu, count_pack_kernel<false, PK_ARENA_PROT>, 64, 0);
        if (ke != hipSuccess || pk_per_cu < 1) pk_per_cu = 1;
        if (const char *e = getenv("KAAMER_PACK_PER_CU")) { const int v = atoi(e); if (v >= 1 && v < pk_per_cu) pk_per_cu = v; }
        ws->pack_grid = ws->n_cu * pk_per_cu;
        int pl = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&pl, count_pack_kernel<true, PK_ARENA_ORF_LONG>, 64, 0) != hipSuccess || pl < 1) pl = 1;
        ws->pack_grid_long = ws->n_cu * pl;
    }
    // the pack window: a protein query's table is ~530 slots, so a window of 512 slots holds one or two table starts
    // (tables up to PK_ARENA - 448 slots at full size); an ORF batch has tables of 64-128 slots: 8-16 ORFs to a pack
    ws->pack_shift = 9u;
    if (const char *e = getenv("KAAMER_PACK_SHIFT")) { const int v = atoi(e); if (v >= 8 && v <= 10) ws->pack_shift = (uint32_t)v; }
    // Which counting kernel: ORF batches (tables of 64-128 slots, 24 one-wave workgroups per CU) take the pack kernel;
    // protein batches the group kernel -- a wave alone on a 700-position query is a 50 us critical path, and with the
    // 11 waves per CU a protein arena allows the pack kernel took 117-156 us against the group kernel's 100
    // (profiles/r03_count_kernels.md).  KAAMER_COUNT_GROUP / KAAMER_COUNT_PACK force one or the other (A/B runs).
    ws->use_group = !ws->nucleotide;
    if (getenv("KAAMER_COUNT_GROUP")) ws->use_group = true;
    if (getenv("KAAMER_COUNT_PACK")) ws->use_group = false;
    // protein batches: count_group_kernel takes UNITS of two group windows (count_group.hip.inc); the barrier-free kernel
    // keeps single windows (two of them alive in its arena)
    // (units of four windows -- an arena of 98 KB, one workgroup per CU: KAAMER_UNIT_WINDOWS=4 -- were measured for the
    // overlapping launch, where there is one counting workgroup per CU anyway: 0.157-0.159 against 0.1255-0.1287 ms per batch;
    // the probe kernels of the neighbouring batches miss the LDS and the registers.  Two it is.)
    ws->unit_windows = 2;
    if (const char *e = getenv("KAAMER_UNIT_WINDOWS")) { const int v = atoi(e); if (v == 2 || (v == 4 && grp_per_cu == 1)) ws->unit_windows = v; }
    if (ws->use_group) ws->pack_shift = ws->count_async ? GRP_SHIFT : ws->unit_windows == 4 ? GRP_SHIFT + 2u : GRP_SHIFT + 1u;
    if (const char *e = getenv("KAAMER_GROUP_SHIFT")) { const int v = atoi(e); if (ws->use_group && !ws->count_async && ws->unit_windows == 2 && (v == (int)GRP_SHIFT || v == (int)GRP_SHIFT + 1)) ws->pack_shift = (uint32_t)v; }
    // a table has at most max(64, 3 x SizeInKmer) slots
    {
        // (a merge counts partial entries instead of positions: as many as max_hits)
        const uint64_t items = ws->pos_cap > ws->hit_cap ? ws->pos_cap : ws->hit_cap;
        const uint64_t gc = (((uint64_t)GRP_MIN_TABLE * ws->q_cap + 3 * items) >> 8) + 4;  // windows of 256 slots at the least
        ws->groups_cap = (uint32_t)(gc > 0x7FFFFFFFull ? 0x7FFFFFFFull : gc);
    }
    int rc = KAAMER_OK;
    {
        uint64_t n = ws->q_cap;
        if (ws->nucleotide && (uint64_t)ws->max_seqs * 6 > n) n = (uint64_t)ws->max_seqs * 6;
        if (ws->nucleotide) {
            const uint64_t ml = opts->max_seq_bytes / (TS_MAX + 1) + 1;
            ws->max_long = (uint32_t)(ml < ws->max_seqs ? ml : ws->max_seqs);
            ws->max_piece_items = 6ull * (opts->max_seq_bytes / 3 / TP_PIECE + ws->max_long + 1);
            if (ws->max_piece_items > n) n = ws->max_piece_items;
        }
        ws->n_scan_blocks = (uint32_t)((n + 1 + SCAN_TILE - 1) / SCAN_TILE);
    }
    rc = dev_alloc(&ws->d_q, ws->q_cap);
    if (!rc) rc = dev_alloc(&ws->d_nq, 1);
    if (!rc) rc = dev_alloc(&ws->d_n_pos, 1);
    if (!rc) rc = dev_alloc(&ws->d_valid, (size_t)(ws->pos_cap / 64 + 2));
    if (!rc) rc = dev_alloc(&ws->d_vals, (size_t)ws->pos_cap);
    if (!rc && ws->nucleotide) {
        const size_t n6 = (size_t)ws->max_seqs * 6;
        rc = dev_alloc(&ws->d_cnt3, 3 * n6);
        if (!rc) rc = dev_alloc(&ws->d_off3, 3 * (n6 + 1));
        if (!rc) rc = dev_alloc(&ws->d_n6, 1);
        if (!rc) rc = dev_alloc(&ws->d_long_seq, (size_t)ws->max_long + 1);
        if (!rc) rc = dev_alloc(&ws->d_piece_li, (size_t)ws->max_piece_items / 6 + 2);
        if (!rc) rc = dev_alloc(&ws->d_long_np, (size_t)ws->max_long + 1);
        if (!rc) rc = dev_alloc(&ws->d_piece_base, (size_t)ws->max_long + 2);
        if (!rc) rc = dev_alloc(&ws->d_pcnt3, 3 * (size_t)ws->max_piece_items);
        if (!rc) rc = dev_alloc(&ws->d_poff3, 3 * ((size_t)ws->max_piece_items + 1));
        if (!rc) rc = dev_alloc(&ws->d_n_piece_items, 1);
        if (!rc) rc = dev_alloc(&ws->d_tmp_meta, ws->q_cap);
        const size_t tcw = 3 * ((size_t)ws->max_seqs / 64 + 2);
        if (!rc) rc = dev_alloc(&ws->d_tr_chain, tcw);
        if (!rc && hipMemset(ws->d_tr_chain, 0, tcw * sizeof(unsigned long long)) != hipSuccess) rc = kaamer_fail(KAAMER_E_HIP, "memset");
        if (!rc) rc = dev_alloc(&ws->d_orf_aa, (size_t)ws->aa_cap + 64);
        if (!rc) rc = dev_alloc(&ws->d_starts_alt, (size_t)ws->sa_cap + 64);
    }
    if (!rc) rc = dev_alloc(&ws->d_q_cnt, ws->q_cap);
    if (!rc) rc = dev_alloc(&ws->d_pool_cursor, (size_t)3 * CURSOR_STRIDE);
    if (!rc) rc = dev_alloc(&ws->d_lists, (size_t)N_LISTS * ws->q_cap);
    if (!rc) rc = dev_alloc(&ws->d_list_counts, N_SMALL_SLOTS);
    if (!rc) rc = dev_alloc(&ws->d_status_out, 1);
    if (!rc) rc = dev_alloc(&ws->d_qinfo, ws->q_cap);
    if (!rc) rc = dev_alloc(&ws->d_slots, ws->q_cap);
    if (!rc) rc = dev_alloc(&ws->d_slot_off, (size_t)ws->q_cap + 1);
    if (!rc) rc = dev_alloc(&ws->d_group_first, ws->groups_cap);
    if (!rc) rc = dev_alloc(&ws->d_sched, ws->groups_cap);
    if (!rc) rc = dev_alloc(&ws->d_group_start, ws->groups_cap);
    if (!rc) rc = dev_alloc(&ws->d_lay_total, 1);
    if (!rc) rc = dev_alloc(&ws->d_slot_scale, 4);
    if (!rc) {
        const uint32_t one = SLOT_SCALE_ONE;
        if (hipMemcpy(ws->d_slot_scale, &one, 4, hipMemcpyHostToDevice) != hipSuccess) rc = kaamer_fail(KAAMER_E_HIP, "workspace init");
        // the tables of a batch may take 70 % of the hit arrays (the G tier's lists come after them).  How far a batch's
        // tables scale inside that room is settled per batch on the device, from the batch's own positions
        // (fit_table_scale): a workspace sized for 2 amino acids per nucleotide holds ORFs of 0.5 per nucleotide at 4 x
        ws->table_room = (unsigned long long)(0.7 * (double)ws->sparse_cap);
        double cap = 128.0;
        // (never above 8: at 3e9 residues on one device, 6.6 hits per k-mer, tables of 12 x 1.5 x SizeInKmer leave the pack
        // kernel's arena and the batch takes 92 ms instead of 44; profiles/r03_dense_database.md)
        if (getenv("KAAMER_SLOT_SCALE_MAX")) cap = atof(getenv("KAAMER_SLOT_SCALE_MAX")) * 16.0;
        ws->slot_scale_cap = cap < 16.0 ? 16u : cap > 128.0 ? 128u : (uint32_t)cap;
        ws->slot_scale_margin = 30u;   // x 1.9: the hits of a query spread around the batch's mean (measured on DB-UR-lite: 1.2 -> 22.8 ms per
                                       // batch, 1.5 -> 13.9, 1.9 -> 11.9; profiles/r03_dense_database.md)
        ws->dense_tables = getenv("KAAMER_PACK_LONG") != nullptr;
        if (getenv("KAAMER_SLOT_MARGIN")) ws->slot_scale_margin = (uint32_t)(atof(getenv("KAAMER_SLOT_MARGIN")) * 16.0);
        if (getenv("KAAMER_WS_TRACE"))
            fprintf(stderr, "[kaamer workspace] pos_cap %llu q_cap %u hit_cap %llu sparse_cap %llu table_room %llu slot_scale_cap %u/16\n", (unsigned long long)ws->pos_cap,
                    ws->q_cap, (unsigned long long)ws->hit_cap, (unsigned long long)ws->sparse_cap, ws->table_room, ws->slot_scale_cap);
    }
    if (!rc) rc = dev_alloc(&ws->d_n_sched, 1);
    if (!rc) rc = dev_alloc(&ws->d_n_groups, 1);

    ws->want_positions = opts->want_positions != 0;
    if (!rc && ws->want_positions) {
        ws->bits_cap = opts->max_pos_words ? opts->max_pos_words : ws->hit_cap * 8;
        rc = dev_alloc(&ws->d_pos_words, ws->q_cap);
        if (!rc) rc = dev_alloc(&ws->d_pos_base, (size_t)ws->q_cap + 1);
        if (!rc) rc = dev_alloc(&ws->d_pos_off, ws->sparse_cap);
        if (!rc) rc = dev_alloc(&ws->d_pos_bits, ws->bits_cap);
    }
    if (!rc) rc = dev_alloc(&ws->d_g_keys, 4 * ws->g_slots);  // 16-byte slots {id, count, lowest position, -}
    if (!rc) rc = dev_alloc(&ws->d_counter_replicas, (size_t)CTR_REPLICAS * CTR_N);
    if (!rc) rc = dev_alloc(&ws->d_counters, 1);
    if (!rc) rc = dev_alloc(&ws->d_bsum, ws->n_scan_blocks);
    if (!rc) rc = dev_alloc(&ws->d_chain, (size_t)ws->q_cap / PL_TILE + 2);
    if (!rc && hipMemset(ws->d_chain, 0, ((size_t)ws->q_cap / PL_TILE + 2) * sizeof(unsigned long long)) != hipSuccess) rc = kaamer_fail(KAAMER_E_HIP, "memset");
    if (!rc) rc = dev_alloc(&ws->d_hit_off, (size_t)ws->q_cap + 1);
    ws->compact = opts->compact != 0;
    if (!rc) rc = dev_alloc(&ws->d_hit_pid, ws->sparse_cap);
    if (!rc) rc = dev_alloc(&ws->d_hit_km, ws->sparse_cap);
    if (!rc) rc = dev_alloc(&ws->d_hit_fp, ws->sparse_cap);
    // first positions are written only when asked for; otherwise the array reads as zeros
    if (!rc && !ws->firstpos && hipMemset(ws->d_hit_fp, 0, ws->sparse_cap * sizeof(uint32_t)) != hipSuccess) rc = kaamer_fail(KAAMER_E_HIP, "memset");
    if (!rc && ws->compact) {
        rc = dev_alloc(&ws->d_csr_off, (size_t)ws->q_cap + 1);
        if (!rc) rc = dev_alloc(&ws->d_c_pid, ws->hit_cap);
        if (!rc) rc = dev_alloc(&ws->d_c_km, ws->hit_cap);
        if (!rc) rc = dev_alloc(&ws->d_c_fp, ws->hit_cap);
        if (!rc && !ws->firstpos && hipMemset(ws->d_c_fp, 0, ws->hit_cap * sizeof(uint32_t)) != hipSuccess) rc = kaamer_fail(KAAMER_E_HIP, "memset");
    }
    if (rc) { kaamer_workspace_free(ws); return rc; }
    ws->ev = new (std::nothrow) std::vector<hipEvent_t>();
    if (!ws->ev) { kaamer_workspace_free(ws); return kaamer_fail(KAAMER_E_NOMEM, "event ring"); }
    *out = ws;
    return KAAMER_OK;
}

// table layout + query groups (count_group.hip.inc): one single-pass kernel
static void launch_layout(kaamer_workspace *ws, uint32_t nq_bound, uint32_t *status, hipStream_t s, uint32_t bshift, bool schedule = true)
{
    const uint32_t cshift = bshift > GRP_SHIFT ? 10u : bshift >= GRP_SHIFT ? 9u : 7u;  // 16 schedule classes over the slots a group / pack can hold
    ws->lay_epoch = (ws->lay_epoch + 1u) & 0xFFFFFFu;
    if (ws->lay_epoch == 0) ws->lay_epoch = 1;
    const uint32_t tiles = (uint32_t)(((uint64_t)nq_bound + 1 + LAY_TILE - 1) / LAY_TILE);
    hipLaunchKernelGGL(layout_kernel, dim3(tiles), dim3(LAY_THREADS), 0, s, ws->d_slots, ws->d_nq, ws->d_slot_off, ws->d_group_first,
                       ws->d_group_start, ws->d_n_groups, ws->groups_cap, ws->d_chain, ws->lay_epoch, status, bshift);
    if (schedule)  // (the pack kernel takes its packs by ticket in layout order: no schedule)
        hipLaunchKernelGGL(schedule_kernel, dim3(1), dim3(1024), 0, s, ws->d_group_first, ws->d_group_start, ws->d_slot_off, ws->d_n_groups,
                           ws->d_nq, ws->d_sched, ws->d_n_sched, cshift);
}

// optional last step of a search / merge: CSR in query order from the sharded hit arrays
static void launch_compaction(kaamer_workspace *ws, uint32_t nq_bound, uint32_t *status, hipStream_t s)
{
    if (nq_bound <= 8 * SCAN_TILE) {
        hipLaunchKernelGGL(scan_single_kernel, dim3(1), dim3(SCAN_BLOCK), 0, s, ws->d_q_cnt, ws->d_nq, ws->d_csr_off);
    } else {
        const uint32_t nsb = (uint32_t)(((uint64_t)nq_bound + 1 + SCAN_TILE - 1) / SCAN_TILE);
        hipLaunchKernelGGL(scan_block_sums_kernel, dim3(nsb), dim3(SCAN_BLOCK), 0, s, ws->d_q_cnt, ws->d_nq, ws->d_bsum);
        hipLaunchKernelGGL(scan_top_kernel, dim3(1), dim3(SCAN_BLOCK), 0, s, ws->d_bsum, nsb);
        hipLaunchKernelGGL(scan_apply_kernel, dim3(nsb), dim3(SCAN_BLOCK), 0, s, ws->d_q_cnt, ws->d_nq, ws->d_bsum, ws->d_csr_off);
    }
    uint32_t gb = (nq_bound + 3) / 4;
    if (gb < 1) gb = 1;
    if (gb > (uint32_t)ws->n_cu * 8) gb = (uint32_t)ws->n_cu * 8;
    hipLaunchKernelGGL(gather_hits_kernel, dim3(gb), dim3(256), 0, s, ws->d_nq, ws->d_csr_off, ws->d_hit_off, ws->d_q_cnt, ws->d_hit_pid,
                       ws->d_hit_km, ws->d_hit_fp, ws->d_c_pid, ws->d_c_km, ws->d_c_fp, ws->hit_cap, status, 1);
}

static void fill_result_hits(const kaamer_workspace *ws, kaamer_device_result *out)
{
    out->d_hit_cnt = ws->d_q_cnt;
    out->hit_capacity = ws->compact ? ws->hit_cap : ws->sparse_cap;
    out->d_hit_off = ws->compact ? ws->d_csr_off : ws->d_hit_off;
    out->d_hit_pid = ws->compact ? ws->d_c_pid : ws->d_hit_pid;
    out->d_hit_kmatch = ws->compact ? ws->d_c_km : ws->d_hit_km;
    out->d_hit_first_pos = ws->compact ? ws->d_c_fp : ws->d_hit_fp;
}

// whoever consumes the workspace's last search on `stream` first waits for its counting stage, if that ran elsewhere
static int ws_join(kaamer_workspace *ws, hipStream_t stream)
{
    if (ws->split_pending && ws->count_stream && ws->count_stream != stream) HIPCHK(hipStreamWaitEvent(stream, ws->ev_count, 0));
    return KAAMER_OK;
}

int kaamer_workspace_set_count_stream(kaamer_workspace *ws, void *stream)
{
    if (!ws) return kaamer_fail(KAAMER_E_ARG, "workspace_set_count_stream: bad argument");
    HIPCHK(hipSetDevice(ws->device));
    if (ws->split_pending) HIPCHK(hipEventSynchronize(ws->ev_count));
    ws->split_pending = false;
    ws->count_stream = (hipStream_t)stream;
    if (stream && !ws->ev_probe) {
        HIPCHK(hipEventCreateWithFlags(&ws->ev_probe, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&ws->ev_count, hipEventDisableTiming));
    }
    return KAAMER_OK;
}

int kaamer_search_device(kaamer_index *ix, kaamer_workspace *ws, const uint8_t *d_seqs, const uint64_t *d_offsets,
                         uint32_t n_seqs, uint64_t seq_bytes, int32_t seq_type, void *stream,
                         kaamer_device_result *out)
{
    if (!ix || !ws || !out || (n_seqs && (!d_seqs || !d_offsets))) return kaamer_fail(KAAMER_E_ARG, "search_device: bad argument");
    if (ix->device != ws->device) return kaamer_fail(KAAMER_E_ARG, "workspace belongs to another device");
    const bool nucl = seq_type == KAAMER_NUCLEOTIDE || seq_type == KAAMER_READS;
    if (!nucl && seq_type != KAAMER_PROTEIN) return kaamer_fail(KAAMER_E_ARG, "search_device: unknown sequence type %d", seq_type);
    if (nucl != ws->nucleotide) return kaamer_fail(KAAMER_E_ARG, "workspace was created for %s input", ws->nucleotide ? "nucleotide" : "protein");
    if (n_seqs > ws->max_seqs) return kaamer_fail(KAAMER_E_CAPACITY, "batch of %u sequences exceeds workspace max_seqs %u", n_seqs, ws->max_seqs);
    if (ws->pos_cap > 0xFFFFFFF0ull) return kaamer_fail(KAAMER_E_CAPACITY, "a batch holds at most 2^32 residue positions");
    if (seq_bytes > ws->opts.max_seq_bytes) return kaamer_fail(KAAMER_E_CAPACITY, "batch of %llu bytes exceeds workspace max_seq_bytes", (unsigned long long)seq_bytes);
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipSetDevice(ix->device));
    {   // the previous batch's counting stage may still be reading this workspace on the count stream
        const int jrc = ws_join(ws, s);
        if (jrc) return jrc;
        ws->split_pending = false;
    }
    // kernel timers: an event record costs a few microseconds of stream idle time, so they are
    // off unless asked for, and can sample every k-th call (kaamer_workspace_set_timing)
    const bool timed = ws->time_every && (ws->call_no++ % ws->time_every) == 0;
    hipEvent_t *ev = nullptr;
    if (timed) {
        if (ws->n_timed >= MAX_TIMED_CALLS) ws->n_timed = 0;
        while (ws->ev->size() < (size_t)(ws->n_timed + 1) * EV_PER_CALL) {
            hipEvent_t e;
            HIPCHK(hipEventCreate(&e));
            ws->ev->push_back(e);
        }
        ev = ws->ev->data() + (size_t)ws->n_timed * EV_PER_CALL;
        HIPCHK(hipEventRecord(ev[0], s));
    }
    if (!ws->clean) {
        // first batch, or a previous batch did not run to its finalize kernel
        HIPCHK(hipMemsetAsync(ws->d_pool_cursor, 0, (size_t)3 * CURSOR_STRIDE * sizeof(unsigned long long), s));
        HIPCHK(hipMemsetAsync(ws->d_list_counts, 0, N_SMALL_SLOTS * sizeof(uint32_t), s));
        HIPCHK(hipMemsetAsync(ws->d_counter_replicas, 0, sizeof(unsigned long long) * CTR_REPLICAS * CTR_N, s));
        HIPCHK(hipMemsetAsync(ws->d_valid, 0, (size_t)(ws->pos_cap / 64 + 2) * sizeof(unsigned long long), s));
    }
    ws->clean = false;

    uint32_t *status = ws->d_list_counts + SLOT_STATUS;
    uint32_t *queue_head = ws->d_list_counts + SLOT_QUEUE_HEAD;
    const int pb = 256;
    const uint8_t *residues = d_seqs;   // what kernel P reads: the protein records, or the ORF amino acids
    uint64_t pos_bound = seq_bytes;     // host-side bound of the residue positions (grid sizing only)
    uint32_t nq_bound = n_seqs;         // host-side bound of the number of queries
    if (!nucl) {
        // prep + table layout + group schedule: one launch (count_group.hip.inc)
        PrepLayoutParams pl;
        memset(&pl, 0, sizeof pl);
        pl.seqs = d_seqs; pl.pos_bound = seq_bytes; pl.offsets = d_offsets; pl.n_seqs = n_seqs;
        pl.q = ws->d_q; pl.d_nq = ws->d_nq; pl.d_n_pos = ws->d_n_pos; pl.invalid = ws->d_valid; pl.qinfo = ws->d_qinfo;
        pl.hit_off = ws->d_hit_off; pl.q_cnt = ws->d_q_cnt;
        pl.E = ws->d_slot_off; pl.group_first = ws->d_group_first; pl.group_start = ws->d_group_start;
        pl.d_n_groups = ws->d_n_groups; pl.groups_cap = ws->groups_cap; pl.chain = ws->d_chain;
        ws->lay_epoch = (ws->lay_epoch + 1u) & 0xFFFFFFu;
        if (ws->lay_epoch == 0) ws->lay_epoch = 1;
        pl.epoch = ws->lay_epoch;
        pl.status = status; pl.sched = ws->d_sched; pl.d_n_sched = ws->d_n_sched;
        pl.tiles_done = ws->d_list_counts + SLOT_TILES_DONE;
        pl.slots = ws->d_slots; pl.bshift = ws->pack_shift; pl.cshift = ws->pack_shift > GRP_SHIFT ? 10u : ws->pack_shift >= GRP_SHIFT ? 9u : 7u;
        // (KAAMER_SCHED_IDENTITY=1: no longest-first schedule, groups in index order -- an A/B knob)
        pl.no_sched = (ws->use_group && !ws->sched_identity) ? 0u : 1u;
        pl.d_total = ws->d_lay_total;
        pl.slot_scale = ws->d_slot_scale;
        pl.room_slots = ws->table_room;
        const uint32_t tiles = (uint32_t)(((uint64_t)n_seqs + 1 + PL_TILE - 1) / PL_TILE);
        hipLaunchKernelGGL(prep_layout_schedule_kernel, dim3(tiles), dim3(LAY_THREADS), 0, s, pl);
    } else {
        // 6-frame translation: count, scan, write, order (translate.hip.inc)
        const size_t cap6 = (size_t)ws->max_seqs * 6;
        TranslateParams tp;
        memset(&tp, 0, sizeof tp);
        tp.seqs = d_seqs; tp.offsets = d_offsets; tp.n_seqs = n_seqs; tp.seq_bound = seq_bytes; tp.max_long = ws->max_long; tp.max_piece_items = ws->max_piece_items;
        tp.cnt_orf = ws->d_cnt3; tp.cnt_aa = ws->d_cnt3 + cap6; tp.cnt_sa = ws->d_cnt3 + 2 * cap6;
        tp.off_orf = ws->d_off3; tp.off_aa = ws->d_off3 + (cap6 + 1); tp.off_sa = ws->d_off3 + 2 * (cap6 + 1);
        tp.tmp_meta = ws->d_tmp_meta; tp.orf_aa = ws->d_orf_aa; tp.starts_alt = ws->d_starts_alt;
        tp.q_cap = ws->q_cap; tp.aa_cap = ws->aa_cap; tp.sa_cap = ws->sa_cap; tp.status = status;
        // which lane-per-read kernel: the wide one (reads up to 384 nt, 2 waves per SIMD) when the batch's mean length says that
        // reads beyond 192 nt are common; a batch of 100-150-nt reads is 4 % slower through it (translate.hip.inc)
        bool wide_reads = seq_bytes > 160ull * (n_seqs ? n_seqs : 1u);
        if (const char *e = getenv("KAAMER_WIDE_READS")) wide_reads = atoi(e) != 0;
        tp.ts_max = wide_reads ? TS_MAX_WIDE : TS_MAX;
        int tgrid = ws->n_cu * 8;  // 256-thread blocks, one (sequence, frame, piece) item per wave at a time
        tp.d_n6 = ws->d_n6;
        int sgrid = ws->n_cu * 8;  // lane-per-read kernel: 2-wave blocks, 13 KB (COUNT) / 30 KB (WRITE) of LDS each
        if ((uint64_t)sgrid * TS_WAVES * 64 > (uint64_t)n_seqs) sgrid = n_seqs > 0 ? (int)(((uint64_t)n_seqs + TS_WAVES * 64 - 1) / (TS_WAVES * 64)) : 1;
        tp.n_long = ws->d_list_counts + SLOT_N_LONG;
        const size_t mpi = (size_t)ws->max_piece_items;
        tp.long_seq = ws->d_long_seq; tp.long_np = ws->d_long_np; tp.piece_base = ws->d_piece_base; tp.piece_li = ws->d_piece_li;
        tp.pcnt_orf = ws->d_pcnt3; tp.pcnt_aa = ws->d_pcnt3 + mpi; tp.pcnt_sa = ws->d_pcnt3 + 2 * mpi;
        tp.poff_orf = ws->d_poff3; tp.poff_aa = ws->d_poff3 + (mpi + 1); tp.poff_sa = ws->d_poff3 + 2 * (mpi + 1);
        tp.d_n_piece_items = ws->d_n_piece_items;
        auto scan_u32 = [&](const uint32_t *cnt, const uint32_t *d_count, uint64_t bound, uint64_t *off) {
            if (bound <= 8 * (uint64_t)SCAN_TILE) {
                hipLaunchKernelGGL(scan_single_kernel, dim3(1), dim3(SCAN_BLOCK), 0, s, cnt, d_count, off);
            } else {
                const uint32_t nsb = (uint32_t)((bound + 1 + SCAN_TILE - 1) / SCAN_TILE);
                hipLaunchKernelGGL(scan_block_sums_kernel, dim3(nsb), dim3(SCAN_BLOCK), 0, s, cnt, d_count, ws->d_bsum);
                hipLaunchKernelGGL(scan_top_kernel, dim3(1), dim3(SCAN_BLOCK), 0, s, ws->d_bsum, nsb);
                hipLaunchKernelGGL(scan_apply_kernel, dim3(nsb), dim3(SCAN_BLOCK), 0, s, cnt, d_count, ws->d_bsum, off);
            }
        };
        // reads: a lane each; longer sequences: listed, cut in pieces, a wave per piece
        const uint64_t long_bound = (uint64_t)ws->max_long < n_seqs ? ws->max_long : n_seqs;
        const uint64_t seq_long_max = seq_bytes / (TS_MAX + 1) + 1;
        const uint64_t n_long_bound = long_bound < seq_long_max ? long_bound : seq_long_max;
        const uint64_t piece_bound = 6ull * (seq_bytes / 3 / TP_PIECE + n_long_bound + 1);
        if ((uint64_t)tgrid * 4 > piece_bound) tgrid = (int)((piece_bound + 3) / 4);
        if (tgrid < 1) tgrid = 1;
        tp.ticket = ws->d_list_counts + SLOT_TR_TICKET;
        tp.chain = ws->d_tr_chain;
        ws->tr_epoch = (ws->tr_epoch + 1u) & 0xFFFFFFu;
        if (ws->tr_epoch == 0) ws->tr_epoch = 1;
        tp.epoch = ws->tr_epoch;
        tp.out_meta = ws->d_q; tp.d_nq = ws->d_nq; tp.d_n_pos = ws->d_n_pos;
        tp.w_off_orf = ws->d_off3; tp.w_off_aa = ws->d_off3 + (cap6 + 1); tp.w_off_sa = ws->d_off3 + 2 * (cap6 + 1);
        // long sequences first (listed, cut in pieces, counted: a wave per piece) -- a reads-only batch falls through
        // these launches; then everything of the reads and the output offsets of both kinds in one kernel
        hipLaunchKernelGGL(list_long_kernel, dim3((unsigned)(((uint64_t)n_seqs + LL_BLOCK - 1) / LL_BLOCK + (n_seqs ? 0 : 1))), dim3(LL_BLOCK), 0, s, tp);
        scan_u32(ws->d_long_np, tp.n_long, n_long_bound, ws->d_piece_base);
        hipLaunchKernelGGL(piece_li_kernel, dim3((unsigned)((n_long_bound + 255) / 256)), dim3(256), 0, s, tp);
        hipLaunchKernelGGL(translate_kernel<false>, dim3(tgrid), dim3(256), 0, s, tp);
        for (int a = 0; a < 3; a++) scan_u32(ws->d_pcnt3 + a * mpi, ws->d_n_piece_items, piece_bound, ws->d_poff3 + a * (mpi + 1));
        hipLaunchKernelGGL(long_totals_kernel, dim3((unsigned)((n_long_bound * 6 + 255) / 256)), dim3(256), 0, s, tp);
        if (wide_reads) hipLaunchKernelGGL(translate_reads_kernel<TS_MAX_WIDE>, dim3(sgrid), dim3(64 * TS_WAVES), 0, s, tp);
        else hipLaunchKernelGGL(translate_reads_kernel<TS_MAX>, dim3(sgrid), dim3(64 * TS_WAVES), 0, s, tp);
        hipLaunchKernelGGL(translate_kernel<true>, dim3(tgrid), dim3(256), 0, s, tp);
        hipLaunchKernelGGL(orf_order_long_kernel, dim3((unsigned)(n_long_bound < (uint64_t)ws->n_cu * 32 ? (n_long_bound + 3) / 4 + 1 : (uint64_t)ws->n_cu * 8)), dim3(256), 0, s,
                           ws->d_tmp_meta, tp.off_orf, ws->d_long_seq, tp.n_long, ws->d_q, ws->d_nq);
        hipLaunchKernelGGL(prep_orf_kernel, dim3(ws->n_cu * 4), dim3(pb), 0, s, ws->d_q, ws->d_nq, ws->d_valid, ws->d_n_pos,
                           ws->d_qinfo, ws->d_slots, ws->d_hit_off, ws->d_q_cnt, ws->d_slot_scale, ws->table_room);
        residues = ws->d_orf_aa;
        pos_bound = ws->aa_cap;
        nq_bound = ws->q_cap;
    }

    // ---- query groups: table layout, first query of each group, schedule (protein: done with the prep)
    if (nucl) launch_layout(ws, nq_bound, status, s, ws->pack_shift, ws->use_group);

    // ---- kernel P: flat probe
    ProbeParams pp;
    pp.table = reinterpret_cast<const uint4 *>(ix->d_buckets);
    pp.n_buckets = ix->hdr.n_buckets;
    pp.n_shards = ix->hdr.n_shards;
    pp.shard = ix->hdr.shard;
    pp.residues = residues;
    pp.invalid = ws->d_valid;
    pp.d_n_pos = ws->d_n_pos;
    pp.vals = ws->d_vals;
    pp.counters = ws->d_counter_replicas;
    pp.nontemporal = pos_bound < ix->hdr.n_buckets ? 1u : 0u;
    uint64_t p_blocks = (pos_bound / 64 + 1 + P_WAVES - 1) / P_WAVES;
    if (p_blocks > (uint64_t)ws->p_grid) p_blocks = ws->p_grid;
    if (p_blocks < 1) p_blocks = 1;
    if (timed) HIPCHK(hipEventRecord(ev[1], s));
    if (ws->nucleotide) hipLaunchKernelGGL(probe_kernel<true>, dim3((unsigned)p_blocks), dim3(64 * P_WAVES), 0, s, pp);
    else hipLaunchKernelGGL(probe_kernel<false>, dim3((unsigned)p_blocks), dim3(64 * P_WAVES), 0, s, pp);
    if (timed) HIPCHK(hipEventRecord(ev[2], s));
    const bool split = ws->count_stream && ws->count_stream != s && !timed;
    if (split) {   // everything from here on runs on the count stream, behind the probe kernel
        HIPCHK(hipEventRecord(ws->ev_probe, s));
        HIPCHK(hipStreamWaitEvent(ws->count_stream, ws->ev_probe, 0));
        s = ws->count_stream;
    }
    // ---- kernel C: counting
    CountParams p;
    memset(&p, 0, sizeof p);
    p.arena = ix->d_arena;
    p.vals = ws->d_vals;
    p.qinfo = ws->d_qinfo;
    p.slot_off = ws->d_slot_off;
    p.group_first = ws->d_group_first;
    p.sched = ws->d_sched;
    p.d_n_sched = ws->d_n_sched;
    p.d_n_groups = ws->d_n_groups;
    p.d_nq = ws->d_nq;
    p.list_cap = ws->q_cap;
    p.queue_head = queue_head;
    p.group_queue = ws->d_list_counts + SLOT_GROUP_QUEUE;
    p.hit_off = ws->d_hit_off;
    p.q_cnt = ws->d_q_cnt;
    p.hit_pid = ws->d_hit_pid;
    p.hit_km = ws->d_hit_km;
    p.hit_fp = ws->d_hit_fp;
    p.hit_cap = ws->sparse_cap;
    p.tail_cursor = ws->d_pool_cursor;
    p.g_keys = ws->d_g_keys;
    p.g_slots = ws->g_slots;
    p.g_cursor = ws->d_pool_cursor + CURSOR_STRIDE;
    p.n_proteins = ix->hdr.max_protein_id + 1u ? ix->hdr.max_protein_id + 1u : 0xFFFFFFFFu;
    p.counters = ws->d_counter_replicas;
    p.status = status;
    auto list_ptr = [&](int which) { return ws->d_lists + (size_t)which * ws->q_cap; };

    CountParams pc = p;
    pc.ovf_list = list_ptr(LIST_G); pc.ovf_count = ws->d_list_counts + LIST_G;
    pc.last_group_pass = ws->want_positions ? 0u : 1u;
    pc.pack_shift = ws->pack_shift;
    pc.pack_tickets = ws->d_list_counts + SLOT_PACK_TICKETS;
    {
        // one wave per workgroup, packs dealt out statically: never more waves than packs
        uint64_t gb = (((uint64_t)GRP_MIN_TABLE * nq_bound + 3 * pos_bound) >> ws->pack_shift) + 1;
        if (gb > (uint64_t)ws->pack_grid) gb = ws->pack_grid;
        if (ws->use_group) {
            uint64_t gg = ((uint64_t)GRP_MIN_TABLE * nq_bound + 3 * pos_bound) / GRP_BUDGET + 1;
            if (gg > (uint64_t)ws->grp_grid) gg = ws->grp_grid;
            launch_group(pc, (int)gg, ws->firstpos, s, ws->count_async, ws->unit_windows);
        } else if (ws->nucleotide) {
            // reads of ~150 nt: ORFs of <= 50 residues, tables of 64-128 slots -> the small arena (20 waves per CU);
            // longer sequences (mixed read lengths, contigs): the arena that holds tables of up to 576 slots, or 66 000
            // ORFs of a 1 M mixed-read batch went to the G tier (measured: 8.19 -> 6.73 ms for the counting stage of
            // that batch, while the 150-nt batch loses 0.4 ms with the larger arena)
            const bool long_orfs = seq_bytes > 200ull * (n_seqs ? n_seqs : 1u) || ws->dense_tables;
            uint64_t g2 = gb;
            if (long_orfs) {
                if (g2 > (uint64_t)ws->pack_grid_long) g2 = ws->pack_grid_long;
                hipLaunchKernelGGL((count_pack_kernel<true, PK_ARENA_ORF_LONG>), dim3((unsigned)g2), dim3(64), 0, s, pc);
            } else {
                hipLaunchKernelGGL((count_pack_kernel<true, PK_ARENA_ORF>), dim3((unsigned)g2), dim3(64), 0, s, pc);
            }
        }
        else if (ws->firstpos) hipLaunchKernelGGL((count_pack_kernel<true, PK_ARENA_PROT>), dim3((unsigned)gb), dim3(64), 0, s, pc);
        else hipLaunchKernelGGL((count_pack_kernel<false, PK_ARENA_PROT>), dim3((unsigned)gb), dim3(64), 0, s, pc);
    }
    CountParams pg = p;
    pg.list = list_ptr(LIST_G); pg.list_count = ws->d_list_counts + LIST_G;
    int g_grid = ws->g_grid;
    if ((uint32_t)g_grid > nq_bound) g_grid = nq_bound > 0 ? (int)nq_bound : 1;
    const bool fused_finalize = !ws->compact && !ws->want_positions;
    if (fused_finalize) {
        pg.fin_out = ws->d_counters; pg.fin_small = ws->d_list_counts; pg.fin_status_out = ws->d_status_out;
        pg.fin_cursors = ws->d_pool_cursor;
        pg.fin_slot_scale = ws->d_slot_scale; pg.fin_scale_cap = ws->slot_scale_cap; pg.fin_scale_margin = ws->slot_scale_margin;
    }
    hipLaunchKernelGGL(count_global_kernel, dim3(g_grid), dim3(64 * G_WAVES), 0, s, pg);
    if (ws->compact) launch_compaction(ws, nq_bound, status, s);
    if (timed) HIPCHK(hipEventRecord(ev[3], s));

    if (ws->want_positions) {
        // PositionHits bitmaps (search.go:442-452): layout from the final hit lists, then one more
        // pass of the group kernel that sets one bit per (hit, position)
        uint32_t gb = (nq_bound + 255) / 256;
        if (gb < 1) gb = 1;
        if (gb > (uint32_t)ws->n_cu * 8) gb = (uint32_t)ws->n_cu * 8;
        hipLaunchKernelGGL(pos_words_kernel, dim3(gb), dim3(256), 0, s, ws->d_qinfo, ws->d_q_cnt, ws->d_nq, ws->d_pos_words);
        if (nq_bound <= 8 * SCAN_TILE) {
            hipLaunchKernelGGL(scan_single_kernel, dim3(1), dim3(SCAN_BLOCK), 0, s, ws->d_pos_words, ws->d_nq, ws->d_pos_base);
        } else {
            const uint32_t nsb = (uint32_t)(((uint64_t)nq_bound + 1 + SCAN_TILE - 1) / SCAN_TILE);
            hipLaunchKernelGGL(scan_block_sums_kernel, dim3(nsb), dim3(SCAN_BLOCK), 0, s, ws->d_pos_words, ws->d_nq, ws->d_bsum);
            hipLaunchKernelGGL(scan_top_kernel, dim3(1), dim3(SCAN_BLOCK), 0, s, ws->d_bsum, nsb);
            hipLaunchKernelGGL(scan_apply_kernel, dim3(nsb), dim3(SCAN_BLOCK), 0, s, ws->d_pos_words, ws->d_nq, ws->d_bsum, ws->d_pos_base);
        }
        hipLaunchKernelGGL(pos_layout_kernel, dim3(ws->n_cu * 8), dim3(256), 0, s, ws->d_qinfo, ws->d_q_cnt,
                           ws->compact ? ws->d_csr_off : ws->d_hit_off,
                           ws->d_pos_base, ws->d_nq, ws->d_pos_off, ws->d_pos_bits, ws->bits_cap, status);
        // the PositionHits pass runs on the group kernel: lay the tables out in its groups
        launch_layout(ws, nq_bound, status, s, GRP_SHIFT);
        uint64_t grp_blocks = ((uint64_t)GRP_MIN_TABLE * nq_bound + 3 * pos_bound) / GRP_BUDGET + 1;
        if (grp_blocks > (uint64_t)ws->grp_grid) grp_blocks = ws->grp_grid;
        CountParams pp2 = p;
        pp2.pack_shift = GRP_SHIFT;   // (the layout just above)
        pp2.last_group_pass = 1u;
        pp2.pos_base = ws->d_pos_base;
        pp2.pos_bits = ws->d_pos_bits;
        pp2.group_queue = ws->d_list_counts + SLOT_GROUP_QUEUE_POS;
        pp2.ovf_list = list_ptr(LIST_SO); pp2.ovf_count = ws->d_list_counts + LIST_SO;
        launch_group_positions(pp2, (int)grp_blocks, s);
        CountParams pg2 = pg;   // the G tier's queries: their hit lists are final, the counting arena is free again
        pg2.fin_out = nullptr;
        pg2.pos_base = ws->d_pos_base;
        pg2.pos_bits = ws->d_pos_bits;
        pg2.g_cursor = ws->d_pool_cursor + 2 * CURSOR_STRIDE;
        hipLaunchKernelGGL(positions_global_kernel, dim3(g_grid), dim3(64 * G_WAVES), 0, s, pg2);
    }
    if (!fused_finalize)
        hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(256), 0, s, ws->d_counter_replicas, ws->d_counters, ws->d_list_counts,
                           ws->d_status_out, ws->d_pool_cursor, ws->d_slot_scale, ws->slot_scale_cap, ws->slot_scale_margin);
    if (timed) HIPCHK(hipEventRecord(ev[4], s));
    HIPCHK(hipGetLastError());
    if (split) {
        HIPCHK(hipEventRecord(ws->ev_count, s));
        ws->split_pending = true;
    }
    ws->clean = true;  // everything up to finalize is enqueued
    ws->last_was_merge = false;
    if (timed) ws->n_timed++;

    out->n_queries_cap = ws->q_cap;
    out->d_n_queries = ws->d_nq;
    out->d_q = ws->d_q;
    fill_result_hits(ws, out);
    out->d_orf_aa = nucl ? ws->d_orf_aa : nullptr;
    out->d_starts_alt = nucl ? ws->d_starts_alt : nullptr;
    out->d_counters = ws->d_counters;
    out->d_pos_off = ws->want_positions ? ws->d_pos_off : nullptr;
    out->d_pos_bits = ws->want_positions ? (const uint64_t *)ws->d_pos_bits : nullptr;
    out->d_pos_base = ws->want_positions ? ws->d_pos_base : nullptr;
    return KAAMER_OK;
}

// n_queries / n_entries are host-side bounds when d_n_queries (a device scalar) gives the actual number of queries
static int merge_device_impl(kaamer_workspace *ws, const uint64_t *d_ent_off, const uint32_t *d_pid, const uint32_t *d_km,
                             const uint32_t *d_fp, uint32_t n_queries, const uint32_t *d_n_queries, uint64_t n_entries, void *stream,
                             kaamer_device_result *out)
{
    if (!ws || !out || (n_queries && !d_ent_off) || (n_entries && (!d_pid || !d_km || !d_fp)))
        return kaamer_fail(KAAMER_E_ARG, "merge_device: bad argument");
    if (n_queries > ws->q_cap) return kaamer_fail(KAAMER_E_CAPACITY, "merge of %u queries exceeds the workspace (%u)", n_queries, ws->q_cap);
    if (n_entries > ws->hit_cap) return kaamer_fail(KAAMER_E_CAPACITY, "merge of %llu entries exceeds workspace max_hits", (unsigned long long)n_entries);
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipSetDevice(ws->device));
    if (!ws->clean) {
        HIPCHK(hipMemsetAsync(ws->d_pool_cursor, 0, (size_t)3 * CURSOR_STRIDE * sizeof(unsigned long long), s));
        HIPCHK(hipMemsetAsync(ws->d_list_counts, 0, N_SMALL_SLOTS * sizeof(uint32_t), s));
        HIPCHK(hipMemsetAsync(ws->d_counter_replicas, 0, sizeof(unsigned long long) * CTR_REPLICAS * CTR_N, s));
        HIPCHK(hipMemsetAsync(ws->d_valid, 0, (size_t)(ws->pos_cap / 64 + 2) * sizeof(unsigned long long), s));
    }
    ws->clean = false;
    uint32_t *status = ws->d_list_counts + SLOT_STATUS;
    const uint32_t nq_bound = n_queries;
    const int pb = 256;
    hipLaunchKernelGGL(prep_merge_kernel, dim3((n_queries + pb - 1) / pb > 0 ? (n_queries + pb - 1) / pb : 1), dim3(pb), 0, s, d_ent_off,
                       n_queries, d_n_queries, ws->d_qinfo, ws->d_slots, ws->d_nq, ws->d_hit_off, ws->d_q_cnt);
    // (the merge runs on count_group_kernel<., 2>: units of two group windows, as the search of protein batches)
    launch_layout(ws, nq_bound, status, s, GRP_SHIFT + 1u);
    CountParams p;
    memset(&p, 0, sizeof p);
    p.qinfo = ws->d_qinfo;
    p.slot_off = ws->d_slot_off;
    p.group_first = ws->d_group_first;
    p.sched = ws->d_sched;
    p.d_n_sched = ws->d_n_sched;
    p.d_n_groups = ws->d_n_groups;
    p.d_nq = ws->d_nq;
    p.last_group_pass = 1u;
    p.pack_shift = GRP_SHIFT + 1u;
    p.group_queue = ws->d_list_counts + SLOT_GROUP_QUEUE;
    p.m_pid = d_pid; p.m_km = d_km; p.m_fp = d_fp; p.merge_fp = ws->firstpos ? 1u : 0u;
    p.list_cap = ws->q_cap;
    p.hit_off = ws->d_hit_off;
    p.q_cnt = ws->d_q_cnt;
    p.hit_pid = ws->d_hit_pid; p.hit_km = ws->d_hit_km; p.hit_fp = ws->d_hit_fp;
    p.hit_cap = ws->sparse_cap;
    p.tail_cursor = ws->d_pool_cursor;
    p.g_keys = ws->d_g_keys;
    p.g_slots = ws->g_slots;
    p.g_cursor = ws->d_pool_cursor + CURSOR_STRIDE;
    p.counters = ws->d_counter_replicas;
    p.status = status;
    auto list_ptr = [&](int which) { return ws->d_lists + (size_t)which * ws->q_cap; };
    p.ovf_list = list_ptr(LIST_G); p.ovf_count = ws->d_list_counts + LIST_G;
    {
        uint64_t gb = ((uint64_t)GRP_MIN_TABLE * nq_bound + 3 * n_entries) / GRP_BUDGET + 1;
        if (gb > (uint64_t)ws->grp_grid) gb = ws->grp_grid;
        launch_group_merge(p, (int)gb, ws->firstpos, s);
    }
    CountParams pg = p;
    pg.list = list_ptr(LIST_G); pg.list_count = ws->d_list_counts + LIST_G;
    int g_grid = ws->g_grid;
    if ((uint32_t)g_grid > nq_bound) g_grid = nq_bound > 0 ? (int)nq_bound : 1;
    hipLaunchKernelGGL(merge_global_kernel, dim3(g_grid), dim3(256), 0, s, pg);
    if (ws->compact) launch_compaction(ws, nq_bound, status, s);
    hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(256), 0, s, ws->d_counter_replicas, ws->d_counters, ws->d_list_counts,
                       ws->d_status_out, ws->d_pool_cursor, (uint32_t *)nullptr, 16u, 16u);   // (a merge looks nothing up: the scale stays)
    HIPCHK(hipGetLastError());
    ws->clean = true;
    ws->last_was_merge = true;
    memset(out, 0, sizeof *out);
    out->n_queries_cap = ws->q_cap;
    out->d_n_queries = ws->d_nq;
    fill_result_hits(ws, out);
    out->d_counters = ws->d_counters;
    return KAAMER_OK;
}

int kaamer_merge_device(kaamer_workspace *ws, const uint64_t *d_ent_off, const uint32_t *d_pid, const uint32_t *d_km,
                        const uint32_t *d_fp, uint32_t n_queries, uint64_t n_entries, void *stream, kaamer_device_result *out)
{
    return merge_device_impl(ws, d_ent_off, d_pid, d_km, d_fp, n_queries, nullptr, n_entries, stream, out);
}

// ---- the exchange step of the sharded index (exchange.hip.inc) --------------------------------------
int kaamer_exchange_layout_init(uint32_t world, uint32_t rank, uint32_t max_queries, uint64_t max_entries_per_peer,
                                kaamer_exchange_layout *out)
{
    if (!out || world == 0 || world > 64 || rank >= world || max_entries_per_peer == 0 || max_entries_per_peer > 0xFFFFFFF0ull)
        return kaamer_fail(KAAMER_E_ARG, "exchange_layout_init: bad argument (1 <= world <= 64, rank < world, 0 < entries < 2^32)");
    memset(out, 0, sizeof *out);
    out->world = world;
    out->rank = rank;
    out->q_cap = (max_queries + world - 1) / world + 1;
    out->arrays = 3;
    out->e_cap = (max_entries_per_peer + 3) & ~3ull;
    out->block_words = (X_HDR + (uint64_t)out->q_cap + 3 * out->e_cap + 3) & ~3ull;
    return KAAMER_OK;
}

// scratch of the tiled scans: (world + 1) rows of tile sums
static int x_tiles(kaamer_workspace *ws, const kaamer_exchange_layout *L, XParams *x)
{
    const uint32_t n_tiles = (uint32_t)(((uint64_t)L->q_cap + X_TILE - 1) / X_TILE);
    const size_t need = (size_t)(L->world + 1) * n_tiles;
    if (ws->x_tiles_cap < need) {
        if (ws->d_x_tiles) (void)hipFree(ws->d_x_tiles);
        ws->d_x_tiles = nullptr; ws->x_tiles_cap = 0;
        const int rc = dev_alloc(&ws->d_x_tiles, need);
        if (rc) return rc;
        ws->x_tiles_cap = need;
    }
    x->tile_sum = ws->d_x_tiles;
    x->n_tiles = n_tiles;
    return KAAMER_OK;
}

static void x_fill(XParams &x, const kaamer_exchange_layout *L)
{
    memset(&x, 0, sizeof x);
    x.world = L->world; x.rank = L->rank; x.q_cap = L->q_cap; x.e_cap = L->e_cap; x.block_words = L->block_words;
}

int kaamer_exchange_pack(kaamer_workspace *ws, const kaamer_exchange_layout *L, uint32_t *d_send, void *stream)
{
    if (!ws || !L || !d_send || L->world == 0) return kaamer_fail(KAAMER_E_ARG, "exchange_pack: bad argument");
    if (L->arrays == 2 && ws->firstpos) return kaamer_fail(KAAMER_E_ARG, "exchange_pack: the layout has no room for the first positions this workspace computes");
    // (a layout fitted to fewer queries than the batch turns out to have is an overflow like any other: flagged in the
    // block headers by the scan, reported by every owner)
    HIPCHK(hipSetDevice(ws->device));
    { const int jrc = ws_join(ws, (hipStream_t)stream); if (jrc) return jrc; }
    if (ws->x_dst_cap < (size_t)L->world * L->q_cap) {
        HIPCHK(hipStreamSynchronize((hipStream_t)stream));
        if (ws->d_x_dst_off) (void)hipFree(ws->d_x_dst_off);
        ws->d_x_dst_off = nullptr; ws->x_dst_cap = 0;
        const int rc = dev_alloc(&ws->d_x_dst_off, (size_t)L->world * L->q_cap);
        if (rc) return rc;
        ws->x_dst_cap = (size_t)L->world * L->q_cap;
    }
    XParams x;
    x_fill(x, L);
    x.d_nq = ws->d_nq;
    x.src_status = ws->d_status_out;  // written by the search's finalize step, earlier on this stream
    x.hit_off = ws->compact ? ws->d_csr_off : ws->d_hit_off;
    x.hit_cnt = ws->d_q_cnt;
    x.pid = ws->compact ? ws->d_c_pid : ws->d_hit_pid;
    x.km = ws->compact ? ws->d_c_km : ws->d_hit_km;
    x.fp = ws->compact ? ws->d_c_fp : ws->d_hit_fp;
    x.with_fp = ws->firstpos ? 1u : 0u;
    x.send = d_send;
    x.dst_off = ws->d_x_dst_off;
    hipStream_t s = (hipStream_t)stream;
    {
        const int rc = x_tiles(ws, L, &x);
        if (rc) return rc;
    }
    if (x.n_tiles <= X_SMALL_TILES) {
        hipLaunchKernelGGL(x_pack_scan_small_kernel, dim3(L->world), dim3(X_BLOCK), 0, s, x);
    } else {
        hipLaunchKernelGGL(x_scan_sums_kernel<0>, dim3(x.n_tiles, L->world), dim3(X_BLOCK), 0, s, x);
        hipLaunchKernelGGL(x_scan_top_kernel<0>, dim3(1), dim3(X_BLOCK), 0, s, x);
        hipLaunchKernelGGL(x_scan_apply_kernel<0>, dim3(x.n_tiles, L->world), dim3(X_BLOCK), 0, s, x);
    }
    const bool wide = ws->q_cap <= X_WIDE_QUERIES;   // few queries with long lists: a wave per query
    uint32_t gb = wide ? (ws->q_cap + 3) / 4 : (ws->q_cap + 15) / 16;  // 4 / 16 queries per 256-thread block
    if (gb > (uint32_t)ws->n_cu * 8) gb = (uint32_t)ws->n_cu * 8;
    if (gb < 1) gb = 1;
    if (wide) hipLaunchKernelGGL(x_pack_copy_kernel<X_GROUP_WIDE>, dim3(gb), dim3(256), 0, s, x);
    else hipLaunchKernelGGL(x_pack_copy_kernel<X_GROUP>, dim3(gb), dim3(256), 0, s, x);
    HIPCHK(hipGetLastError());
    return KAAMER_OK;
}

int kaamer_exchange_merge(kaamer_workspace *ws, const kaamer_exchange_layout *L, const uint32_t *d_recv, void *stream,
                          kaamer_device_result *out)
{
    if (!ws || !L || !d_recv || !out || L->world == 0) return kaamer_fail(KAAMER_E_ARG, "exchange_merge: bad argument");
    if (L->arrays == 2 && ws->firstpos) return kaamer_fail(KAAMER_E_ARG, "exchange_merge: the layout has no room for the first positions this workspace wants");
    if (L->q_cap > ws->q_cap) return kaamer_fail(KAAMER_E_CAPACITY, "exchange_merge: %u owned queries exceed the merge workspace (%u)", L->q_cap, ws->q_cap);
    const uint64_t m_cap = (uint64_t)L->world * L->e_cap;
    if (m_cap > ws->hit_cap) return kaamer_fail(KAAMER_E_CAPACITY, "exchange_merge: %llu entries exceed the merge workspace's max_hits (%llu)",
                                                 (unsigned long long)m_cap, (unsigned long long)ws->hit_cap);
    HIPCHK(hipSetDevice(ws->device));
    if (ws->x_src_cap < (size_t)L->world * L->q_cap || ws->x_ent_cap < (size_t)L->q_cap + 1 || ws->x_m_cap < (size_t)m_cap || !ws->d_x_nq_owned) {
        HIPCHK(hipStreamSynchronize((hipStream_t)stream));
        void *bufs[] = { ws->d_x_src_off, ws->d_x_nq_owned, ws->d_x_pid, ws->d_x_km, ws->d_x_fp, ws->d_x_ent_off, ws->d_x_stats };
        for (void *b : bufs) if (b) (void)hipFree(b);
        ws->d_x_src_off = ws->d_x_nq_owned = ws->d_x_pid = ws->d_x_km = ws->d_x_fp = nullptr;
        ws->d_x_ent_off = nullptr; ws->d_x_stats = nullptr;
        ws->x_src_cap = ws->x_ent_cap = ws->x_m_cap = 0;
        int rc = dev_alloc(&ws->d_x_src_off, (size_t)L->world * L->q_cap);
        if (!rc) rc = dev_alloc(&ws->d_x_nq_owned, 1);
        if (!rc) rc = dev_alloc(&ws->d_x_ent_off, (size_t)L->q_cap + 1);
        if (!rc) rc = dev_alloc(&ws->d_x_pid, (size_t)m_cap);
        if (!rc) rc = dev_alloc(&ws->d_x_km, (size_t)m_cap);
        if (!rc) rc = dev_alloc(&ws->d_x_fp, (size_t)m_cap);
        if (!rc) rc = dev_alloc(&ws->d_x_stats, 4);
        if (rc) return rc;
        ws->x_src_cap = (size_t)L->world * L->q_cap; ws->x_ent_cap = (size_t)L->q_cap + 1; ws->x_m_cap = (size_t)m_cap;
    }
    if (!ws->h_x_stats) {
        if (hipHostMalloc((void **)&ws->h_x_stats, 8 * sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess)
            return kaamer_fail(KAAMER_E_NOMEM, "exchange_merge: pinned statistics");
        memset(ws->h_x_stats, 0, 8 * sizeof(unsigned long long));
        for (int i = 0; i < 2; i++) HIPCHK(hipEventCreateWithFlags(&ws->ev_x_stats[i], hipEventDisableTiming));
    }
    hipStream_t s = (hipStream_t)stream;
    if (!ws->clean) {  // the status word the unpack kernels may set must start from zero
        HIPCHK(hipMemsetAsync(ws->d_pool_cursor, 0, (size_t)3 * CURSOR_STRIDE * sizeof(unsigned long long), s));
        HIPCHK(hipMemsetAsync(ws->d_list_counts, 0, N_SMALL_SLOTS * sizeof(uint32_t), s));
        HIPCHK(hipMemsetAsync(ws->d_counter_replicas, 0, sizeof(unsigned long long) * CTR_REPLICAS * CTR_N, s));
        HIPCHK(hipMemsetAsync(ws->d_valid, 0, (size_t)(ws->pos_cap / 64 + 2) * sizeof(unsigned long long), s));
        ws->clean = true;
    }
    XParams x;
    x_fill(x, L);
    x.recv = d_recv;
    x.src_off = ws->d_x_src_off;
    x.ent_off = ws->d_x_ent_off;
    x.d_nq_owned = ws->d_x_nq_owned;
    x.m_pid = ws->d_x_pid; x.m_km = ws->d_x_km; x.m_fp = ws->d_x_fp;
    x.with_fp = ws->firstpos ? 1u : 0u;
    x.m_cap = m_cap;
    x.stats = ws->d_x_stats;
    x.status = ws->d_list_counts + SLOT_STATUS;
    {
        const int rc = x_tiles(ws, L, &x);
        if (rc) return rc;
    }
    if (x.n_tiles <= X_SMALL_TILES) {
        hipLaunchKernelGGL(x_unpack_scan_small_kernel, dim3(L->world + 1), dim3(X_BLOCK), 0, s, x);
    } else {
        hipLaunchKernelGGL(x_scan_sums_kernel<1>, dim3(x.n_tiles, L->world + 1), dim3(X_BLOCK), 0, s, x);
        hipLaunchKernelGGL(x_scan_top_kernel<1>, dim3(1), dim3(X_BLOCK), 0, s, x);
        hipLaunchKernelGGL(x_scan_apply_kernel<1>, dim3(x.n_tiles, L->world + 1), dim3(X_BLOCK), 0, s, x);
    }
    const bool wide = (uint64_t)L->q_cap * L->world <= X_WIDE_QUERIES;
    uint32_t gb = wide ? (L->q_cap + 3) / 4 : (L->q_cap + 15) / 16;
    if (gb > (uint32_t)ws->n_cu * 8) gb = (uint32_t)ws->n_cu * 8;
    if (gb < 1) gb = 1;
    if (wide) hipLaunchKernelGGL(x_unpack_copy_kernel<X_GROUP_WIDE>, dim3(gb), dim3(256), 0, s, x);
    else hipLaunchKernelGGL(x_unpack_copy_kernel<X_GROUP>, dim3(gb), dim3(256), 0, s, x);
    HIPCHK(hipGetLastError());
    {   // what the W headers said, to the host: read by kaamer_exchange_stats without waiting for the merge itself
        const int slot = (int)(ws->x_merge_seq & 1u);
        HIPCHK(hipMemcpyAsync(ws->h_x_stats + 4 * slot, ws->d_x_stats, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        HIPCHK(hipEventRecord(ws->ev_x_stats[slot], s));
        ws->x_merge_seq++;
    }
    return merge_device_impl(ws, ws->d_x_ent_off, ws->d_x_pid, ws->d_x_km, ws->d_x_fp, L->q_cap, ws->d_x_nq_owned, m_cap, stream, out);
}

uint32_t kaamer_workspace_query_capacity(const kaamer_workspace *ws) { return ws ? ws->q_cap : 0u; }

int kaamer_exchange_layout_fit(const kaamer_exchange_layout *cap, uint32_t n_queries, uint64_t entries_per_block, int32_t with_first_pos,
                               kaamer_exchange_layout *out)
{
    if (!cap || !out || cap->world == 0) return kaamer_fail(KAAMER_E_ARG, "exchange_layout_fit: bad argument");
    kaamer_exchange_layout L = *cap;
    L.arrays = with_first_pos ? 3u : 2u;   // a protein search without -pos sends no first positions: no room for them either
    const uint64_t q = ((uint64_t)n_queries + cap->world - 1) / cap->world + 1;
    if (q < L.q_cap) L.q_cap = (uint32_t)q;
    uint64_t e = (entries_per_block + 3) & ~3ull;
    if (e < 4) e = 4;
    if (e < L.e_cap) L.e_cap = e;
    L.block_words = (X_HDR + (uint64_t)L.q_cap + (uint64_t)L.arrays * L.e_cap + 3) & ~3ull;
    if (L.block_words > cap->block_words) L = *cap;   // never beyond what the buffers hold
    *out = L;
    return KAAMER_OK;
}

int kaamer_exchange_stats(kaamer_workspace *ws, uint32_t back, uint64_t out[4])
{
    if (!ws || !out || back > 1) return kaamer_fail(KAAMER_E_ARG, "exchange_stats: bad argument (back is 0 or 1)");
    if (ws->x_merge_seq <= back || !ws->h_x_stats) return kaamer_fail(KAAMER_E_ARG, "exchange_stats: no such merge yet");
    const uint64_t seq = ws->x_merge_seq - 1 - back;
    const int slot = (int)(seq & 1u);
    HIPCHK(hipSetDevice(ws->device));
    HIPCHK(hipEventSynchronize(ws->ev_x_stats[slot]));
    out[0] = seq;
    out[1] = ws->h_x_stats[4 * slot + 0];
    out[2] = ws->h_x_stats[4 * slot + 1];
    out[3] = ws->h_x_stats[4 * slot + 2];
    return KAAMER_OK;
}

// grouped ncclSend / ncclRecv of equal blocks, for hosts that own an RCCL communicator and nothing else to drive it
// (the library does not link RCCL: the symbols are looked up in the process at first use)
int kaamer_rccl_alltoall(void *nccl_comm, const void *d_send, void *d_recv, uint64_t bytes_per_peer, uint32_t world, void *stream)
{
    typedef int (*grp_t)(void);
    typedef int (*send_t)(const void *, size_t, int, int, void *, void *);
    typedef int (*recv_t)(void *, size_t, int, int, void *, void *);
    static grp_t g_start = nullptr, g_end = nullptr;
    static send_t g_send = nullptr;
    static recv_t g_recv = nullptr;
    if (!nccl_comm || !d_send || !d_recv || world == 0) return kaamer_fail(KAAMER_E_ARG, "rccl_alltoall: bad argument");
    if (!g_send) {
        void *h = dlopen(nullptr, RTLD_NOW);
        void *sym = h ? dlsym(h, "ncclSend") : nullptr;
        if (!sym) { h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL); if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL); }
        if (!h) return kaamer_fail(KAAMER_E_HIP, "rccl_alltoall: librccl is not loaded and cannot be opened");
        g_start = (grp_t)dlsym(h, "ncclGroupStart"); g_end = (grp_t)dlsym(h, "ncclGroupEnd");
        g_recv = (recv_t)dlsym(h, "ncclRecv");
        g_send = (send_t)dlsym(h, "ncclSend");
        if (!g_start || !g_end || !g_send || !g_recv) { g_send = nullptr; return kaamer_fail(KAAMER_E_HIP, "rccl_alltoall: RCCL symbols not found"); }
    }
    const int nccl_uint8 = 1;  // ncclUint8 (nccl.h: ncclInt8 = 0, ncclUint8 = 1)
    int rc = g_start();
    for (uint32_t peer = 0; peer < world && rc == 0; peer++) {
        rc = g_send((const char *)d_send + (size_t)peer * bytes_per_peer, (size_t)bytes_per_peer, nccl_uint8, (int)peer, nccl_comm, stream);
        if (rc == 0) rc = g_recv((char *)d_recv + (size_t)peer * bytes_per_peer, (size_t)bytes_per_peer, nccl_uint8, (int)peer, nccl_comm, stream);
    }
    const int rc2 = g_end();
    if (rc || rc2) return kaamer_fail(KAAMER_E_HIP, "rccl_alltoall: RCCL error %d", rc ? rc : rc2);
    return KAAMER_OK;
}

// FilterResults / top-N (and SetBestStartCodon for nucleotide input) of the workspace's last
// search or merge, on the device; see topn.hip.inc.
int kaamer_topn_device(kaamer_workspace *ws, const kaamer_topn_opts *opts, void *stream, kaamer_topn_result *out)
{
    if (!ws || !opts || !out || opts->max_results < 1) return kaamer_fail(KAAMER_E_ARG, "topn_device: bad argument");
    const bool best_start = opts->best_start_codon != 0;
    const kaamer_workspace *src = opts->orf_source ? opts->orf_source : ws;  // whose queries (ORFs) the results belong to
    if (best_start && (!src->nucleotide || (ws->last_was_merge && !opts->orf_source)))
        return kaamer_fail(KAAMER_E_ARG, "topn_device: SetBestStartCodon needs the ORFs of a nucleotide/reads search");
    if (opts->orf_source && opts->orf_source->device != ws->device) return kaamer_fail(KAAMER_E_ARG, "topn_device: orf_source lives on another device");
    if (ws->last_was_merge && !opts->d_size_in_kmer && !opts->orf_source)
        return kaamer_fail(KAAMER_E_ARG, "topn_device: merged results need d_size_in_kmer or orf_source (the owner's queries)");
    if (best_start && !ws->firstpos) return kaamer_fail(KAAMER_E_ARG, "topn_device: SetBestStartCodon needs first positions (first_pos != 2)");
    HIPCHK(hipSetDevice(ws->device));
    if (ws->topn_k < opts->max_results) {
        uint32_t **bufs[] = { &ws->d_top_pid, &ws->d_top_km, &ws->d_top_fp };
        for (uint32_t **b : bufs) { if (*b) (void)hipFree(*b); *b = nullptr; }
        int rc = 0;
        const size_t n = (size_t)ws->q_cap * opts->max_results;
        rc = dev_alloc(&ws->d_top_pid, n);
        if (!rc) rc = dev_alloc(&ws->d_top_km, n);
        if (!rc) rc = dev_alloc(&ws->d_top_fp, n);
        if (!rc && !ws->d_top_cnt) {
            rc = dev_alloc(&ws->d_top_cnt, ws->q_cap);
            if (!rc) rc = dev_alloc(&ws->d_top_trim, ws->q_cap);
            if (!rc) rc = dev_alloc(&ws->d_top_start, ws->q_cap);
            if (!rc) rc = dev_alloc(&ws->d_top_size, ws->q_cap);
        }
        if (rc) { ws->topn_k = 0; return rc; }
        ws->topn_k = opts->max_results;
    }
    TopnParams p;
    memset(&p, 0, sizeof p);
    p.d_nq = ws->d_nq;
    p.q = src->d_q;
    p.src_nq = src->d_nq;
    p.q_first = opts->orf_source ? opts->q_first : 0u;
    p.q_stride = opts->orf_source ? (opts->q_stride ? opts->q_stride : 1u) : 1u;
    p.size_in = opts->d_size_in_kmer;
    p.hit_cnt = ws->d_q_cnt;
    p.hit_off = ws->compact ? ws->d_csr_off : ws->d_hit_off;
    p.pid = ws->compact ? ws->d_c_pid : ws->d_hit_pid;
    p.km = ws->compact ? ws->d_c_km : ws->d_hit_km;
    p.fp = ws->compact ? ws->d_c_fp : ws->d_hit_fp;
    p.orf_aa = src->d_orf_aa;
    p.starts_alt = src->d_starts_alt;
    p.min_k_ratio = opts->min_k_ratio;
    p.min_k_match = opts->min_k_match;
    p.K = opts->max_results;
    p.best_start = best_start ? 1 : 0;
    p.top_cnt = ws->d_top_cnt; p.top_pid = ws->d_top_pid; p.top_km = ws->d_top_km; p.top_fp = ws->d_top_fp;
    p.trim = ws->d_top_trim; p.start_pos = ws->d_top_start; p.size_out = ws->d_top_size;
    // lanes per query: a protein query has ~190 hits (a wave), an ORF ~15 (16 lanes: four ORFs per wave)
    // a search whose counting stage ran on the count stream: the post-steps follow it there (the caller's stream is the probe
    // stream of the next batches), and consumers keep waiting for ev_count
    const bool on_count_stream = ws->split_pending && ws->count_stream;
    hipStream_t ts = on_count_stream ? ws->count_stream : (hipStream_t)stream;
    if (src->nucleotide) hipLaunchKernelGGL(topn_kernel<16>, dim3(ws->n_cu * 8), dim3(256), 0, ts, p);
    else hipLaunchKernelGGL(topn_kernel<64>, dim3(ws->n_cu * 8), dim3(256), 0, ts, p);
    if (on_count_stream) HIPCHK(hipEventRecord(ws->ev_count, ts));
    HIPCHK(hipGetLastError());
    out->max_results = opts->max_results;
    out->d_top_cnt = ws->d_top_cnt;
    out->d_top_pid = ws->d_top_pid;
    out->d_top_kmatch = ws->d_top_km;
    out->d_top_first_pos = ws->d_top_fp;
    out->d_trim = ws->d_top_trim;
    out->d_start_position = ws->d_top_start;
    out->d_size_in_kmer = ws->d_top_size;
    return KAAMER_OK;
}

// What the device's finalize step left as the next batch's table scale, worked out again on the host from the same
// counters: from a scale of 2 on, ORF batches take the pack kernel with the larger arena (tables of 200-500 slots).
static void ws_note_density(kaamer_workspace *ws, const kaamer_counters &c)
{
    if (!c.n_lookup || getenv("KAAMER_PACK_LONG")) return;
    unsigned long long t = (c.n_hits * ws->slot_scale_margin + c.n_lookup - 1ull) / c.n_lookup;
    if (t > ws->slot_scale_cap) t = ws->slot_scale_cap;
    ws->dense_tables = t >= 2u * SLOT_SCALE_ONE;
}

int kaamer_workspace_finish(kaamer_workspace *ws, void *stream, kaamer_counters *out)
{
    if (!ws) return kaamer_fail(KAAMER_E_ARG, "workspace_finish: bad argument");
    HIPCHK(hipSetDevice(ws->device));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    if (ws->split_pending) HIPCHK(hipEventSynchronize(ws->ev_count));   // (the counting stage ran on the count stream)
    uint32_t status = 0;
    HIPCHK(hipMemcpy(&status, ws->d_status_out, sizeof status, hipMemcpyDeviceToHost));
    kaamer_counters c;
    HIPCHK(hipMemcpy(&c, ws->d_counters, sizeof c, hipMemcpyDeviceToHost));
    if (out) *out = c;
    if (!ws->last_was_merge) ws_note_density(ws, c);
    if (status) ws->clean = false;  // an aborted batch may leave per-batch state behind
    if (status & ST_POOL_FULL) return kaamer_fail(KAAMER_E_CAPACITY, "hit pool exhausted: raise workspace max_hits (now %llu)", (unsigned long long)ws->hit_cap);
    if (status & ST_LIST_FULL) return kaamer_fail(KAAMER_E_CAPACITY, "tier work list exhausted");
    if (status & ST_CHAIN_TIMEOUT) return kaamer_fail(KAAMER_E_HIP, "table layout: a tile never published its total");
    if (status & (ST_QUERY_CAP | ST_AA_CAP))
        return kaamer_fail(KAAMER_E_CAPACITY, "more ORFs than the workspace holds: raise workspace max_queries (now %u)", ws->q_cap);
    if (status & ST_EXCHANGE_CAP) return kaamer_fail(KAAMER_E_CAPACITY, "exchange block capacity exceeded (or the blocks do not describe one batch): raise max_entries_per_peer / max_queries of the exchange layout");
    if (status & ST_PEER_FAILED) return kaamer_fail(KAAMER_E_CAPACITY, "a peer's search of this batch exceeded one of its workspace bounds: nothing was merged (that rank's kaamer_workspace_finish says which bound)");
    if (status & ST_POS_CAP) return kaamer_fail(KAAMER_E_CAPACITY, "position bitmaps exceed the workspace: raise max_pos_words (now %llu)", (unsigned long long)ws->bits_cap);
    if (status & ST_G_ARENA_FULL)
        return kaamer_fail(KAAMER_E_CAPACITY, "global counting arena exhausted: raise workspace g_tier_slots (now %llu)", (unsigned long long)ws->g_slots);
    if (status) return kaamer_fail(KAAMER_E_CAPACITY, "device status 0x%x", status);
    return KAAMER_OK;
}

int kaamer_workspace_kernel_ms_sum(kaamer_workspace *ws, double *probe_ms, double *count_ms, double *total_ms,
                                   uint32_t *n_calls)
{
    if (!ws) return kaamer_fail(KAAMER_E_ARG, "kernel_ms_sum: bad argument");
    HIPCHK(hipSetDevice(ws->device));
    double a = 0, b = 0, c = 0;
    for (uint32_t i = 0; i < ws->n_timed; i++) {
        hipEvent_t *ev = ws->ev->data() + (size_t)i * EV_PER_CALL;
        float x = 0, y = 0, z = 0;
        HIPCHK(hipEventElapsedTime(&x, ev[1], ev[2]));
        HIPCHK(hipEventElapsedTime(&y, ev[2], ev[3]));
        HIPCHK(hipEventElapsedTime(&z, ev[0], ev[4]));
        a += x;
        b += y;
        c += z;
    }
    if (probe_ms) *probe_ms = a;
    if (count_ms) *count_ms = b;
    if (total_ms) *total_ms = c;
    if (n_calls) *n_calls = ws->n_timed;
    return KAAMER_OK;
}

void kaamer_workspace_reset_timers(kaamer_workspace *ws)
{
    if (ws) { ws->n_timed = 0; ws->call_no = 0; }
}

void kaamer_workspace_set_timing(kaamer_workspace *ws, uint32_t every)
{
    if (ws) { ws->time_every = every; ws->call_no = 0; }
}

// ------------------------------------------------------------------------------------
// host-buffer form
// ------------------------------------------------------------------------------------
// The hit arrays of kaamer_search_batch (tens of MB) land in pinned host memory: a D2H copy into fresh
// pageable memory runs at 2-3 GB/s here.  Pinned buffers are kept in a small process-wide cache.
extern "C++" {
namespace {
struct PinnedCache {
    std::mutex mu;
    std::vector<std::pair<void *, size_t>> free_list;
};
PinnedCache g_pinned;

void *pinned_get(size_t bytes, size_t *cap)
{
    {
        std::lock_guard<std::mutex> lock(g_pinned.mu);
        size_t best = (size_t)-1;
        for (size_t i = 0; i < g_pinned.free_list.size(); i++)
            if (g_pinned.free_list[i].second >= bytes && (best == (size_t)-1 || g_pinned.free_list[i].second < g_pinned.free_list[best].second)) best = i;
        if (best != (size_t)-1) {
            void *p = g_pinned.free_list[best].first;
            *cap = g_pinned.free_list[best].second;
            g_pinned.free_list.erase(g_pinned.free_list.begin() + (long)best);
            return p;
        }
    }
    void *p = nullptr;
    const size_t want = bytes + bytes / 4 + 4096;
    if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) return nullptr;
    *cap = want;
    return p;
}

void pinned_put(void *p, size_t cap)
{
    std::lock_guard<std::mutex> lock(g_pinned.mu);
    if (g_pinned.free_list.size() < 32) g_pinned.free_list.emplace_back(p, cap);   // (four full-list callers hold twelve arrays at once)
    else (void)hipHostFree(p);
}

template <class T> struct PinnedArr {
    T *p = nullptr;
    size_t cap = 0;
    PinnedArr() = default;
    PinnedArr(const PinnedArr &) = delete;
    PinnedArr &operator=(const PinnedArr &) = delete;
    ~PinnedArr() { if (p) pinned_put(p, cap); }
    bool resize(size_t n)
    {
        if (p) { pinned_put(p, cap); p = nullptr; }
        p = (T *)pinned_get((n ? n : 1) * sizeof(T), &cap);
        return p != nullptr;
    }
    T *data() { return p; }
};
}  // namespace
}  // extern "C++"

struct batch_out_owner {
    kaamer_batch_out pub;
    std::vector<kaamer_query_meta> q;
    std::vector<uint64_t> hit_off;
    std::vector<uint32_t> hit_cnt;
    PinnedArr<uint32_t> pid, km, fp;
    std::vector<uint8_t> orf_aa;
    std::vector<int32_t> starts_alt;
    std::vector<uint64_t> pos_off, pos_bits;
};

// workspace + staging buffers of a host-buffer call: reused while they are large enough
static int host_slot_acquire(kaamer_index *ix, HostSlot &h, const kaamer_workspace_opts &need, uint64_t seq_bytes, uint32_t n_seqs)
{
    const kaamer_workspace_opts &o = h.opts;
    const bool fits = h.ws && o.seq_type == need.seq_type && o.first_pos == need.first_pos && o.want_positions == need.want_positions &&
                      o.compact == need.compact && o.max_seq_bytes >= need.max_seq_bytes && o.max_seqs >= need.max_seqs &&
                      o.max_hits >= need.max_hits && o.g_tier_slots >= need.g_tier_slots && o.max_queries >= need.max_queries &&
                      o.max_pos_words >= need.max_pos_words && (need.max_hits != 0 || o.max_hits == 0);
    if (!fits) {
        if (h.ws) { kaamer_workspace_free(h.ws); h.ws = nullptr; }
        kaamer_workspace_opts g = need;  // some headroom so that slightly larger batches do not rebuild it
        g.max_seq_bytes = need.max_seq_bytes + need.max_seq_bytes / 4 + 4096;
        g.max_seqs = need.max_seqs + need.max_seqs / 4 + 16;
        if (need.max_hits) g.max_hits = need.max_hits + need.max_hits / 4;
        if (need.max_pos_words) g.max_pos_words = need.max_pos_words + need.max_pos_words / 4;
        const int rc = kaamer_workspace_create(ix, &g, &h.ws);
        if (rc) { h.ws = nullptr; return rc; }
        h.opts = g;
    }
    if (h.seq_cap < seq_bytes + 16) {
        if (h.d_seqs) (void)hipFree(h.d_seqs);
        h.d_seqs = nullptr; h.seq_cap = 0;
        const size_t cap = (size_t)seq_bytes + (size_t)seq_bytes / 4 + 4096;
        const int rc = dev_alloc(&h.d_seqs, cap);
        if (rc) return rc;
        h.seq_cap = cap;
    }
    if (h.off_cap < (size_t)n_seqs + 1) {
        if (h.d_off) (void)hipFree(h.d_off);
        h.d_off = nullptr; h.off_cap = 0;
        const size_t cap = (size_t)n_seqs + (size_t)n_seqs / 4 + 16;
        const int rc = dev_alloc(&h.d_off, cap);
        if (rc) return rc;
        h.off_cap = cap;
    }
    return KAAMER_OK;
}

// host buffers -> device, search enqueued on the slot's stream; nothing is waited for
static int search_batch_enqueue(kaamer_index *ix, HostSlot &slot, const kaamer_batch_in *in, uint64_t max_hits, uint64_t g_slots,
                                uint32_t max_queries, kaamer_device_result *dr)
{
    const uint64_t seq_bytes = in->offsets[in->n_seqs];
    kaamer_workspace_opts o;
    memset(&o, 0, sizeof o);
    o.max_seq_bytes = seq_bytes;
    o.max_seqs = in->n_seqs ? in->n_seqs : 1;
    o.max_hits = max_hits;
    o.g_tier_slots = g_slots;
    o.seq_type = in->seq_type;
    o.first_pos = 1;  // kaamer_batch_out always carries hit_first_pos
    o.want_positions = in->want_positions ? 1u : 0u;
    o.compact = 1u;  // the host form is CSR
    o.max_pos_words = max_hits * 8;
    o.max_queries = max_queries;
    int rc = host_slot_acquire(ix, slot, o, seq_bytes, in->n_seqs);
    if (rc) return rc;
    if (!slot.stream) HIPCHK(hipStreamCreateWithFlags(&slot.stream, hipStreamNonBlocking));
    hipStream_t s = slot.stream;
    hipError_t e = hipSuccess;
    if (seq_bytes) e = hipMemcpyAsync(slot.d_seqs, in->seqs, (size_t)seq_bytes, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(slot.d_off, in->offsets, ((size_t)in->n_seqs + 1) * 8, hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return kaamer_fail(KAAMER_E_HIP, "H2D: %s", hipGetErrorString(e));
    return kaamer_search_device(ix, slot.ws, slot.d_seqs, slot.d_off, in->n_seqs, seq_bytes, in->seq_type, s, dr);
}

// waits for the slot's stream and brings the full hit lists to the host
static int search_batch_collect(HostSlot &slot, const kaamer_batch_in *in, const kaamer_device_result &dr, kaamer_batch_out **out)
{
    kaamer_workspace *ws = slot.ws;
    batch_out_owner *bo = nullptr;
    hipStream_t s = slot.stream;
    kaamer_counters c;
    uint32_t nq = 0;
    uint64_t n_hits = 0, n_words = 0, n_sa = 0;
    unsigned long long n_aa = 0;
    hipError_t e;
    int rc = kaamer_workspace_finish(ws, s, &c);
    if (rc) goto done;
    bo = new (std::nothrow) batch_out_owner();
    if (!bo) { rc = kaamer_fail(KAAMER_E_NOMEM, "batch_out"); goto done; }
    // every copy on the slot's own stream (a plain hipMemcpy goes through the null stream, where concurrent callers
    // would queue behind each other); the host waits where it needs a value to size the next copy
    e = hipMemcpyAsync(&nq, dr.d_n_queries, 4, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e == hipSuccess) { bo->q.resize(nq); bo->hit_off.resize((size_t)nq + 1); bo->hit_cnt.resize((size_t)nq + 1); }
    if (e == hipSuccess && nq) e = hipMemcpyAsync(bo->hit_cnt.data(), dr.d_hit_cnt, (size_t)nq * 4, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess && nq) e = hipMemcpyAsync(bo->q.data(), dr.d_q, (size_t)nq * sizeof(kaamer_query_meta), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(bo->hit_off.data(), dr.d_hit_off, ((size_t)nq + 1) * 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess && ws->want_positions) e = hipMemcpyAsync(&n_words, ws->d_pos_base + nq, sizeof n_words, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess && ws->nucleotide) {
        const size_t cap6 = (size_t)ws->max_seqs * 6;
        e = hipMemcpyAsync(&n_aa, ws->d_n_pos, sizeof n_aa, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipMemcpyAsync(&n_sa, ws->d_off3 + 2 * (cap6 + 1) + (size_t)in->n_seqs * 6, sizeof n_sa, hipMemcpyDeviceToHost, s);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e == hipSuccess) {
        n_hits = bo->hit_off[nq];
        if (!bo->pid.resize(n_hits) || !bo->km.resize(n_hits) || !bo->fp.resize(n_hits)) e = hipErrorOutOfMemory;
        if (e == hipSuccess && n_hits) {
            e = hipMemcpyAsync(bo->pid.data(), dr.d_hit_pid, n_hits * 4, hipMemcpyDeviceToHost, s);
            if (e == hipSuccess) e = hipMemcpyAsync(bo->km.data(), dr.d_hit_kmatch, n_hits * 4, hipMemcpyDeviceToHost, s);
            if (e == hipSuccess) e = hipMemcpyAsync(bo->fp.data(), dr.d_hit_first_pos, n_hits * 4, hipMemcpyDeviceToHost, s);
        }
    }
    if (e == hipSuccess && ws->want_positions) {
        bo->pos_off.resize(n_hits + 1); bo->pos_bits.resize(n_words + 1);
        if (n_hits) e = hipMemcpyAsync(bo->pos_off.data(), dr.d_pos_off, n_hits * 8, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess && n_words) e = hipMemcpyAsync(bo->pos_bits.data(), dr.d_pos_bits, n_words * 8, hipMemcpyDeviceToHost, s);
    }
    if (e == hipSuccess && ws->nucleotide) {
        bo->orf_aa.resize(n_aa + 1); bo->starts_alt.resize(n_sa + 1);
        if (n_aa) e = hipMemcpyAsync(bo->orf_aa.data(), dr.d_orf_aa, n_aa, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess && n_sa) e = hipMemcpyAsync(bo->starts_alt.data(), dr.d_starts_alt, n_sa * sizeof(int32_t), hipMemcpyDeviceToHost, s);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) { rc = kaamer_fail(KAAMER_E_HIP, "D2H: %s", hipGetErrorString(e)); goto done; }
    memset(&bo->pub, 0, sizeof bo->pub);
    bo->pub.n_queries = nq;
    bo->pub.q = bo->q.data();
    bo->pub.hit_off = bo->hit_off.data();
    bo->pub.hit_cnt = bo->hit_cnt.data();
    bo->pub.hit_pid = bo->pid.data();
    bo->pub.hit_kmatch = bo->km.data();
    bo->pub.hit_first_pos = bo->fp.data();
    if (ws->nucleotide) { bo->pub.orf_aa = bo->orf_aa.data(); bo->pub.starts_alt = bo->starts_alt.data(); }
    if (ws->want_positions) { bo->pub.pos_off = bo->pos_off.data(); bo->pub.pos_bits = bo->pos_bits.data(); }
    bo->pub.counters = c;
    *out = &bo->pub;
    bo = nullptr;
done:
    delete bo;
    return rc;
}

static int search_batch_once(kaamer_index *ix, HostSlot &slot, const kaamer_batch_in *in, uint64_t max_hits, uint64_t g_slots,
                             uint32_t max_queries, kaamer_batch_out **out)
{
    kaamer_device_result dr;
    const int rc = search_batch_enqueue(ix, slot, in, max_hits, g_slots, max_queries, &dr);
    if (rc) return rc;
    return search_batch_collect(slot, in, dr, out);
}

// ---- the same call in two halves (the worker pool of search_protein.go:58-118 without a blocked thread per batch when
// the full hit lists are wanted: -pos, PositionHits).  submit copies the caller's buffers (they are borrowed for the call
// only), takes a slot, enqueues; wait collects, repeating the batch from the copy when a bound was too small.
struct kaamer_full_ticket {
    kaamer_index *ix;
    HostSlot *slot;
    std::vector<uint8_t> *seqs;
    std::vector<uint64_t> *offs;
    kaamer_batch_in in;
    kaamer_device_result dr;
    uint64_t max_hits, g_slots;
    uint32_t max_queries;
    int attempt;
};

static HostSlot *host_slot_take(kaamer_index *ix, int32_t seq_type)
{
    HostSlot *slot = nullptr;
    std::unique_lock<std::mutex> lock(ix->pool_mu);
    for (;;) {
        for (HostSlot &h : ix->host)
            if (!h.busy && (!slot || (h.ws && h.opts.seq_type == seq_type && !(slot->ws && slot->opts.seq_type == seq_type)))) slot = &h;
        if (slot) break;
        ix->pool_cv.wait(lock);
    }
    slot->busy = true;
    return slot;
}

static void host_slot_give(kaamer_index *ix, HostSlot *h)
{
    { std::lock_guard<std::mutex> lock(ix->pool_mu); h->busy = false; }
    ix->pool_cv.notify_all();
}

int kaamer_submit_batch_flat(kaamer_index *ix, const uint8_t *seqs, const uint64_t *offsets, uint32_t n_seqs, int32_t seq_type,
                             int32_t want_positions, kaamer_full_ticket **ticket)
{
    if (!ix || !ticket || !offsets || (n_seqs && !seqs)) return kaamer_fail(KAAMER_E_ARG, "submit_batch: bad argument");
    *ticket = nullptr;
    HIPCHK(hipSetDevice(ix->device));
    kaamer_full_ticket *t = new (std::nothrow) kaamer_full_ticket();
    if (!t) return kaamer_fail(KAAMER_E_NOMEM, "ticket");
    memset(t, 0, sizeof *t);
    t->ix = ix;
    t->seqs = new (std::nothrow) std::vector<uint8_t>(seqs, seqs + offsets[n_seqs]);
    t->offs = new (std::nothrow) std::vector<uint64_t>(offsets, offsets + n_seqs + 1);
    if (!t->seqs || !t->offs) { delete t->seqs; delete t->offs; delete t; return kaamer_fail(KAAMER_E_NOMEM, "ticket"); }
    t->in.seqs = t->seqs->data(); t->in.offsets = t->offs->data(); t->in.n_seqs = n_seqs; t->in.seq_type = seq_type;
    t->in.want_positions = want_positions;
    t->max_hits = offsets[n_seqs] * 8 + 65536;
    t->slot = host_slot_take(ix, seq_type);
    const int rc = search_batch_enqueue(ix, *t->slot, &t->in, t->max_hits, t->g_slots, t->max_queries, &t->dr);
    if (rc && rc != KAAMER_E_CAPACITY) {
        if (t->slot->stream) (void)hipStreamSynchronize(t->slot->stream);
        host_slot_give(ix, t->slot);
        delete t->seqs; delete t->offs; delete t;
        return rc;
    }
    if (rc) t->attempt = -1;   // (a bound refused on the host already: wait starts with the retry)
    *ticket = t;
    return KAAMER_OK;
}

int kaamer_wait_batch(kaamer_full_ticket *t, kaamer_batch_out **out)
{
    if (!t || !out) return kaamer_fail(KAAMER_E_ARG, "wait_batch: bad argument");
    *out = nullptr;
    kaamer_index *ix = t->ix;
    int rc = hipSetDevice(ix->device) == hipSuccess ? KAAMER_OK : kaamer_fail(KAAMER_E_HIP, "hipSetDevice");
    const bool nucl = t->in.seq_type == KAAMER_NUCLEOTIDE || t->in.seq_type == KAAMER_READS;
    while (!rc) {
        rc = t->attempt < 0 ? KAAMER_E_CAPACITY : search_batch_collect(*t->slot, &t->in, t->dr, out);
        if (t->attempt < 0) t->attempt = 0;
        if (rc != KAAMER_E_CAPACITY || t->attempt >= 6) break;
        t->attempt++;
        t->max_hits *= 4;
        t->g_slots = t->g_slots ? t->g_slots * 4 : (128ull << 20);
        if (nucl) {
            const uint64_t hard = t->in.offsets[t->in.n_seqs] / 10 + (uint64_t)t->in.n_seqs * 6 + 64;
            t->max_queries = (uint32_t)(hard > 0xFFFFFFF0ull ? 0xFFFFFFF0ull : hard);
        }
        rc = search_batch_enqueue(ix, *t->slot, &t->in, t->max_hits, t->g_slots, t->max_queries, &t->dr);
    }
    if (rc && t->slot->stream) (void)hipStreamSynchronize(t->slot->stream);
    host_slot_give(ix, t->slot);
    delete t->seqs; delete t->offs; delete t;
    return rc;
}

void kaamer_full_ticket_discard(kaamer_full_ticket *t)
{
    if (!t) return;
    (void)hipSetDevice(t->ix->device);
    if (t->slot->stream) (void)hipStreamSynchronize(t->slot->stream);
    if (t->slot->ws) t->slot->ws->clean = false;
    host_slot_give(t->ix, t->slot);
    delete t->seqs; delete t->offs; delete t;
}

int kaamer_search_batch(kaamer_index *ix, const kaamer_batch_in *in, kaamer_batch_out **out)
{
    if (!ix || !in || !out || !in->offsets || (in->n_seqs && !in->seqs)) return kaamer_fail(KAAMER_E_ARG, "search_batch: bad argument");
    *out = nullptr;
    HIPCHK(hipSetDevice(ix->device));
    // a free slot (one whose workspace already serves this kind of batch, if there is one); callers beyond the slots wait
    HostSlot *slot = nullptr;
    {
        std::unique_lock<std::mutex> lock(ix->pool_mu);
        for (;;) {
            for (HostSlot &h : ix->host)
                if (!h.busy && (!slot || (h.ws && h.opts.seq_type == in->seq_type && !(slot->ws && slot->opts.seq_type == in->seq_type)))) slot = &h;
            if (slot) break;
            ix->pool_cv.wait(lock);
        }
        slot->busy = true;
    }
    struct Release {
        kaamer_index *ix; HostSlot *h;
        ~Release() { { std::lock_guard<std::mutex> lock(ix->pool_mu); h->busy = false; } ix->pool_cv.notify_all(); }
    } release{ ix, slot };
    // The hit count of a batch is data dependent: start from a generous estimate and
    // enlarge on KAAMER_E_CAPACITY (the device reports it; results are never partial).
    uint64_t max_hits = in->offsets[in->n_seqs] * 8 + 65536, g_slots = 0;
    uint32_t max_queries = 0;
    const bool nucl = in->seq_type == KAAMER_NUCLEOTIDE || in->seq_type == KAAMER_READS;
    for (int attempt = 0;; attempt++) {
        const int rc = search_batch_once(ix, *slot, in, max_hits, g_slots, max_queries, out);
        if (rc != KAAMER_E_CAPACITY || attempt >= 6) return rc;
        max_hits *= 4;
        g_slots = g_slots ? g_slots * 4 : (128ull << 20);
        if (nucl) {  // hard bound: a frame of n codons holds at most n/21 + 1 ORFs
            const uint64_t hard = in->offsets[in->n_seqs] / 10 + (uint64_t)in->n_seqs * 6 + 64;
            max_queries = (uint32_t)(hard > 0xFFFFFFF0ull ? 0xFFFFFFF0ull : hard);
        }
    }
}

#include "host_top.hip.inc"
#include "host_sharded.hip.inc"
#include "host_replicas.hip.inc"

void kaamer_batch_free(kaamer_batch_out *out)
{
    if (!out) return;
    delete reinterpret_cast<batch_out_owner *>(out);  // pub is the first member
}

}  // extern "C"
