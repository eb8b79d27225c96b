// builder.cpp — host-side emitter of the GPU hash-table layout (kaamer_layout.h).
//
// Semantics follow the reference's offline side:
//   emit   : pkg/makedb/inputFASTA.go:245-248, inputTSV.go:236-239 (every 7-mer
//            window of every protein with len >= 7 -> (key, proteinId))
//   collapse: pkg/indexdb/indexdb.go:68-132 (KeyToList gathers all versions of a
//            key) + pkg/kvstore/kv_store.go:284-305 (RemoveDuplicatesFromSlice)
//            => key -> set<proteinId>
//   sharing: pkg/kvstore/kcomb_store.go:42-85 (identical sets stored once)
// The LSM stores themselves are replaced by the bucketised table.
#include "kaamer_internal.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace {

unsigned n_threads()
{
    unsigned n = std::thread::hardware_concurrency();
    if (const char *e = getenv("KAAMER_BUILD_THREADS")) n = (unsigned)atoi(e);
    if (n < 1) n = 1;
    if (n > 64) n = 64;
    return n;
}

template <class F> void parallel_for(size_t n, unsigned nt, F f)
{
    if (nt <= 1 || n < 2) { f(0, n, 0); return; }
    std::vector<std::thread> th;
    size_t per = (n + nt - 1) / nt;
    for (unsigned t = 0; t < nt; t++) {
        size_t b = std::min(n, per * t), e = std::min(n, per * (t + 1));
        if (b >= e) break;
        th.emplace_back([=] { f(b, e, t); });
    }
    for (auto &x : th) x.join();
}

// sort u64 ascending: partition on the top 11 bits, then std::sort the parts in parallel
void sort_u64(uint64_t *a, size_t n)
{
    unsigned nt = n_threads();
    if (n < (1u << 16) || nt == 1) { std::sort(a, a + n); return; }
    const int B = 11, NB = 1 << B;
    std::vector<size_t> cnt(NB + 1, 0);
    for (size_t i = 0; i < n; i++) cnt[(a[i] >> (64 - B)) + 1]++;
    for (int i = 0; i < NB; i++) cnt[i + 1] += cnt[i];
    uint64_t *tmp = (uint64_t *)malloc(n * sizeof(uint64_t));
    if (!tmp) { std::sort(a, a + n); return; }
    {
        std::vector<size_t> pos(cnt.begin(), cnt.end() - 1);
        for (size_t i = 0; i < n; i++) tmp[pos[a[i] >> (64 - B)]++] = a[i];
    }
    std::atomic<int> next(0);
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++)
        th.emplace_back([&] {
            for (;;) {
                int b = next.fetch_add(1);
                if (b >= NB) break;
                std::sort(tmp + cnt[b], tmp + cnt[b + 1]);
            }
        });
    for (auto &x : th) x.join();
    memcpy(a, tmp, n * sizeof(uint64_t));
    free(tmp);
}

// KAAMER_BUILD_TRACE=1: phase times of a build on stderr
struct Trace {
    bool on = getenv("KAAMER_BUILD_TRACE") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(const char *what, uint64_t n)
    {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[kaamer build] %-22s %8.2f s  (%llu)\n", what, std::chrono::duration<double>(now - t).count(), (unsigned long long)n);
        t = now;
    }
};

uint64_t hash_list(const uint32_t *ids, uint32_t n)
{
    uint64_t h = 0x9E3779B97F4A7C15ull ^ n;
    for (uint32_t i = 0; i < n; i++) {
        h ^= ids[i] + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
        h *= 0xff51afd7ed558ccdull;
        h ^= h >> 33;
    }
    return h ? h : 1;
}

}  // namespace

int kaamer_image_alloc(kaamer_image *img, uint64_t n_buckets, uint64_t arena_words)
{
    img->buckets = nullptr;
    img->arena = nullptr;
    if (posix_memalign((void **)&img->buckets, 4096, (size_t)n_buckets * sizeof(kh_bucket)) != 0)
        return KAAMER_E_NOMEM;
    if (posix_memalign((void **)&img->arena, 4096, (size_t)std::max<uint64_t>(arena_words, 4) * 4) != 0)
        return KAAMER_E_NOMEM;
    img->hdr.n_buckets = n_buckets;
    img->hdr.arena_words = arena_words;
    return KAAMER_OK;
}

// pairs64: key<<32|id, any order, possibly with duplicates; consumed (sorted in place)
static int build_from_sorted_input(uint64_t *pairs64, uint64_t n, uint32_t shard, uint32_t n_shards,
                                   double load, kaamer_image **out)
{
    if (!(load > 0.05 && load <= 0.95)) load = 0.5;
    Trace tr;
    sort_u64(pairs64, n);
    tr.lap("sort pairs", n);
    n = (uint64_t)(std::unique(pairs64, pairs64 + n) - pairs64);
    tr.lap("unique", n);

    // distinct keys
    uint64_t n_keys = 0;
    for (uint64_t i = 0; i < n; i++)
        if (i == 0 || (pairs64[i] >> 32) != (pairs64[i - 1] >> 32)) n_keys++;

    kaamer_image *img = new (std::nothrow) kaamer_image();
    if (!img) return kaamer_fail(KAAMER_E_NOMEM, "image alloc");
    memset(&img->hdr, 0, sizeof img->hdr);
    img->hdr.magic = KH_IMAGE_MAGIC;
    img->hdr.version = KH_IMAGE_VERSION;
    img->hdr.kmer_size = KAAMER_KMER_SIZE;
    img->hdr.shard = shard;
    img->hdr.n_shards = n_shards;
    img->hdr.load_factor = load;
    img->hdr.n_pairs = n;
    img->hdr.n_keys = n_keys;

    uint64_t n_buckets = (uint64_t)((double)n_keys / (KH_SLOTS_PER_BUCKET * load)) + 1;
    if (n_buckets >= (1ull << 32)) { delete img; return kaamer_fail(KAAMER_E_ARG, "too many buckets"); }

    // ---- postings arena with set sharing ------------------------------------
    // pass 1: upper bound of arena words (no sharing) to size the buffer
    uint64_t ub_words = 4;  // offset 0 is reserved (val 0 never used)
    {
        uint64_t i = 0;
        while (i < n) {
            uint64_t j = i + 1;
            while (j < n && (pairs64[j] >> 32) == (pairs64[i] >> 32)) j++;
            uint64_t c = j - i;
            if (!(c == 1 && (uint32_t)pairs64[i] < KH_INLINE_BIT)) ub_words += ((1 + c + 3) / 4) * 4;
            i = j;
        }
    }
    if (ub_words / 4 >= KH_INLINE_BIT) { delete img; return kaamer_fail(KAAMER_E_ARG, "arena exceeds 32 GiB per shard"); }
    int rc = kaamer_image_alloc(img, n_buckets, ub_words);
    if (rc) { kaamer_image_free(img); return kaamer_fail(rc, "image buffers"); }
    memset(img->buckets, 0xFF, (size_t)n_buckets * sizeof(kh_bucket));
    memset(img->arena, 0, 16);

    // dedupe table: open addressing on the 64-bit content hash -> arena offset (16-B units)
    uint64_t n_lists_ub = 0;
    {
        uint64_t i = 0;
        while (i < n) {
            uint64_t j = i + 1;
            while (j < n && (pairs64[j] >> 32) == (pairs64[i] >> 32)) j++;
            if (!(j - i == 1 && (uint32_t)pairs64[i] < KH_INLINE_BIT)) n_lists_ub++;
            i = j;
        }
    }
    uint64_t dcap = 16;
    while (dcap < n_lists_ub * 2) dcap <<= 1;
    std::vector<uint64_t> dh(dcap, 0);
    std::vector<uint32_t> doff(dcap, 0);

    uint64_t words = 4, n_inline = 0, n_lists = 0, max_list = 0, n_displaced = 0;
    uint32_t max_pid = 0;
    std::vector<uint32_t> tmp;
    uint64_t i = 0;
    while (i < n) {
        uint64_t j = i + 1;
        while (j < n && (pairs64[j] >> 32) == (pairs64[i] >> 32)) j++;
        uint32_t key = (uint32_t)(pairs64[i] >> 32);
        uint32_t c = (uint32_t)(j - i);
        uint32_t val;
        max_pid = std::max(max_pid, (uint32_t)pairs64[j - 1]);
        if (c == 1 && (uint32_t)pairs64[i] < KH_INLINE_BIT) {
            val = KH_INLINE_BIT | (uint32_t)pairs64[i];
            n_inline++;
        } else {
            tmp.resize(c);
            for (uint32_t t = 0; t < c; t++) tmp[t] = (uint32_t)pairs64[i + t];  // ascending
            uint64_t h = hash_list(tmp.data(), c);
            uint64_t s = h & (dcap - 1);
            val = 0;
            for (;;) {
                if (dh[s] == 0) break;
                if (dh[s] == h) {
                    const uint32_t *l = img->arena + (uint64_t)doff[s] * 4;
                    if (l[0] == c && memcmp(l + 1, tmp.data(), (size_t)c * 4) == 0) { val = doff[s]; break; }
                }
                s = (s + 1) & (dcap - 1);
            }
            if (!val) {
                val = (uint32_t)(words / 4);
                uint32_t *l = img->arena + words;
                l[0] = c;
                memcpy(l + 1, tmp.data(), (size_t)c * 4);
                uint64_t used = 1 + (uint64_t)c, padded = ((used + 3) / 4) * 4;
                for (uint64_t p = used; p < padded; p++) l[p] = KH_EMPTY_PID;
                words += padded;
                dh[s] = h;
                doff[s] = val;
                n_lists++;
                max_list = std::max<uint64_t>(max_list, c);
            }
        }
        // ---- insert into the bucket table ------------------------------------
        uint64_t b = kh_home_bucket(key, n_shards, n_buckets);
        bool placed = false, home = true;
        for (uint64_t tries = 0; tries < n_buckets && !placed; tries++) {
            kh_bucket &bk = img->buckets[b];
            for (int sidx = 0; sidx < KH_SLOTS_PER_BUCKET; sidx++) {
                if (bk.s[sidx].key == KH_EMPTY_KEY) {
                    bk.s[sidx].key = key;
                    bk.s[sidx].val = val;
                    placed = true;
                    break;
                }
            }
            if (!placed) { b = (b + 1 == n_buckets) ? 0 : b + 1; home = false; }
        }
        if (!placed) { kaamer_image_free(img); return kaamer_fail(KAAMER_E_CAPACITY, "table full"); }
        if (!home) n_displaced++;
        i = j;
    }
    tr.lap("lists + placement", n_keys);
    img->hdr.arena_words = words;
    img->hdr.n_inline = n_inline;
    img->hdr.n_lists = n_lists;
    img->hdr.max_list = max_list;
    img->hdr.n_displaced = n_displaced;
    img->hdr.max_protein_id = max_pid;
    *out = img;
    return KAAMER_OK;
}

extern "C" {

uint32_t kaamer_shard_of(uint32_t key, uint32_t n_shards)
{
    return n_shards <= 1 ? 0 : kh_shard_of(key, n_shards);
}

uint32_t kaamer_encode_kmer(const uint8_t kmer[KAAMER_KMER_SIZE])
{
    return kh_key_from_codes(kh_residue_code(kmer[0]), kh_residue_code(kmer[1]), kh_residue_code(kmer[2]),
                             kh_residue_code(kmer[3]), kh_residue_code(kmer[4]), kh_residue_code(kmer[5]),
                             kh_residue_code(kmer[6]));
}

int kaamer_image_build_pairs(const kaamer_pair *pairs, uint64_t n, uint32_t shard, uint32_t n_shards,
                             double load_factor, kaamer_image **out)
{
    if (!out || (!pairs && n) || n_shards == 0 || shard >= n_shards) return kaamer_fail(KAAMER_E_ARG, "build_pairs: bad argument");
    *out = nullptr;
    uint64_t *p64 = (uint64_t *)malloc((n ? n : 1) * sizeof(uint64_t));
    if (!p64) return kaamer_fail(KAAMER_E_NOMEM, "pairs buffer");
    uint64_t m = 0;
    for (uint64_t i = 0; i < n; i++) {
        if (pairs[i].key == KH_EMPTY_KEY || pairs[i].protein_id == KH_EMPTY_PID) {
            free(p64);
            return kaamer_fail(KAAMER_E_ARG, "key/id 0xFFFFFFFF is reserved");
        }
        if (n_shards > 1 && kh_shard_of(pairs[i].key, n_shards) != shard) continue;
        p64[m++] = ((uint64_t)pairs[i].key << 32) | pairs[i].protein_id;
    }
    int rc = build_from_sorted_input(p64, m, shard, n_shards, load_factor, out);
    free(p64);
    return rc;
}

int kaamer_image_build_proteins(const uint8_t *seqs, const uint64_t *offsets, const uint32_t *ids,
                                uint32_t n_proteins, uint32_t shard, uint32_t n_shards,
                                double load_factor, kaamer_image **out)
{
    if (!out || !offsets || (!seqs && n_proteins) || n_shards == 0 || shard >= n_shards)
        return kaamer_fail(KAAMER_E_ARG, "build_proteins: bad argument");
    *out = nullptr;
    // Two passes over the proteins, both parallel: (1) windows per protein that belong to THIS shard,
    // (2) emit exactly those.  The pair buffer is therefore O(shard), not O(database): a database of
    // W windows in S shards needs 8 W / S bytes here (+ the same again inside the sort).
    auto for_windows = [&](size_t p, auto &&f) {
        const uint8_t *s = seqs + offsets[p];
        const uint64_t len = offsets[p + 1] - offsets[p];
        if (len < KAAMER_KMER_SIZE) return;                                  // inputFASTA.go:228, inputTSV.go:139
        const uint64_t nw = len - KAAMER_KMER_SIZE + 1;                      // inputFASTA.go:245, inputTSV.go:236
        uint32_t c[7];
        for (int t = 0; t < 6; t++) c[t + 1] = kh_residue_code(s[t]);
        for (uint64_t i = 0; i < nw; i++) {
            for (int t = 0; t < 6; t++) c[t] = c[t + 1];
            c[6] = kh_residue_code(s[i + 6]);
            const uint32_t key = kh_key_from_codes(c[0], c[1], c[2], c[3], c[4], c[5], c[6]);
            if (n_shards > 1 && kh_shard_of(key, n_shards) != shard) continue;
            f(key);
        }
    };
    std::vector<uint64_t> woff((size_t)n_proteins + 1, 0);
    std::atomic<int> bad(0);
    parallel_for(n_proteins, n_threads(), [&](size_t b, size_t e, unsigned) {
        for (size_t p = b; p < e; p++) {
            const uint32_t id = ids ? ids[p] : (uint32_t)p;  // inputTSV.go:141-142
            if (id == KH_EMPTY_PID) { bad = 1; continue; }
            if (n_shards == 1) {
                const uint64_t len = offsets[p + 1] - offsets[p];
                woff[p + 1] = len >= KAAMER_KMER_SIZE ? len - KAAMER_KMER_SIZE + 1 : 0;
            } else {
                uint64_t k = 0;
                for_windows(p, [&](uint32_t) { k++; });
                woff[p + 1] = k;
            }
        }
    });
    if (bad) return kaamer_fail(KAAMER_E_ARG, "protein id 0xFFFFFFFF is reserved");
    for (uint32_t p = 0; p < n_proteins; p++) woff[p + 1] += woff[p];
    const uint64_t total = woff[n_proteins];
    uint64_t *p64 = (uint64_t *)malloc((total ? total : 1) * sizeof(uint64_t));
    if (!p64) return kaamer_fail(KAAMER_E_NOMEM, "pairs buffer (%llu pairs)", (unsigned long long)total);
    parallel_for(n_proteins, n_threads(), [&](size_t b, size_t e, unsigned) {
        for (size_t p = b; p < e; p++) {
            const uint32_t id = ids ? ids[p] : (uint32_t)p;
            uint64_t *w = p64 + woff[p];
            for_windows(p, [&](uint32_t key) { *w++ = ((uint64_t)key << 32) | id; });
        }
    });
    int rc = build_from_sorted_input(p64, total, shard, n_shards, load_factor, out);
    free(p64);
    return rc;
}

int kaamer_image_save(const kaamer_image *img, const char *path)
{
    if (!img || !path) return kaamer_fail(KAAMER_E_ARG, "image_save: bad argument");
    FILE *f = fopen(path, "wb");
    if (!f) return kaamer_fail(KAAMER_E_IO, "cannot open for writing");
    bool ok = fwrite(&img->hdr, sizeof img->hdr, 1, f) == 1;
    ok = ok && fwrite(img->buckets, sizeof(kh_bucket), (size_t)img->hdr.n_buckets, f) == img->hdr.n_buckets;
    ok = ok && fwrite(img->arena, 4, (size_t)img->hdr.arena_words, f) == img->hdr.arena_words;
    ok = (fclose(f) == 0) && ok;
    return ok ? KAAMER_OK : kaamer_fail(KAAMER_E_IO, "short write");
}

int kaamer_image_load(const char *path, kaamer_image **out)
{
    if (!path || !out) return kaamer_fail(KAAMER_E_ARG, "image_load: bad argument");
    *out = nullptr;
    FILE *f = fopen(path, "rb");
    if (!f) return kaamer_fail(KAAMER_E_IO, "cannot open index image");
    kaamer_image *img = new (std::nothrow) kaamer_image();
    if (!img) { fclose(f); return kaamer_fail(KAAMER_E_NOMEM, "image alloc"); }
    img->buckets = nullptr;
    img->arena = nullptr;
    if (fread(&img->hdr, sizeof img->hdr, 1, f) != 1) { fclose(f); delete img; return kaamer_fail(KAAMER_E_IO, "short header"); }
    if (img->hdr.magic != KH_IMAGE_MAGIC || img->hdr.version != KH_IMAGE_VERSION || img->hdr.kmer_size != KAAMER_KMER_SIZE) {
        fclose(f); delete img;
        return kaamer_fail(KAAMER_E_FORMAT, "not a kaamer GPU index image (magic/version)");
    }
    uint64_t nb = img->hdr.n_buckets, aw = img->hdr.arena_words;
    // the header drives allocations and device-side indexing: validate it against the file
    {
        const long here = ftell(f);
        bool sane = nb >= 1 && nb < (1ull << 32) && aw >= 4 && aw / 4 < KH_INLINE_BIT && img->hdr.n_shards >= 1 &&
                    img->hdr.shard < img->hdr.n_shards && here == (long)sizeof img->hdr;
        if (sane && fseek(f, 0, SEEK_END) == 0) {
            const long long size = ftell(f);
            sane = size == (long long)(sizeof img->hdr + nb * sizeof(kh_bucket) + aw * 4);
            fseek(f, here, SEEK_SET);
        } else sane = false;
        if (!sane) { fclose(f); delete img; return kaamer_fail(KAAMER_E_FORMAT, "index image header does not match the file"); }
    }
    int rc = kaamer_image_alloc(img, nb, aw);
    if (rc) { fclose(f); kaamer_image_free(img); return kaamer_fail(rc, "image buffers"); }
    bool ok = fread(img->buckets, sizeof(kh_bucket), (size_t)nb, f) == nb;
    ok = ok && fread(img->arena, 4, (size_t)aw, f) == aw;
    fclose(f);
    if (!ok) { kaamer_image_free(img); return kaamer_fail(KAAMER_E_IO, "short read"); }
    // every list reference must stay inside the arena (a corrupted image must not make the kernels read
    // outside the device allocation)
    {
        std::atomic<int> bad_ref(0);
        parallel_for((size_t)nb, n_threads(), [&](size_t b0, size_t b1, unsigned) {
            for (size_t b = b0; b < b1; b++)
                for (int c = 0; c < KH_SLOTS_PER_BUCKET; c++) {
                    const kh_slot &sl = img->buckets[b].s[c];
                    if (sl.key == KH_EMPTY_KEY || (sl.val & KH_INLINE_BIT)) continue;
                    const uint64_t off = (uint64_t)sl.val * 4;
                    if (sl.val == 0 || off + 1 > aw || off + 1 + img->arena[off] > aw) bad_ref = 1;
                }
        });
        if (bad_ref) { kaamer_image_free(img); return kaamer_fail(KAAMER_E_FORMAT, "index image: postings reference out of range"); }
    }
    *out = img;
    return KAAMER_OK;
}

static void fill_stats(const kh_image_header &h, kaamer_image_stats *s)
{
    memset(s, 0, sizeof *s);
    s->n_pairs = h.n_pairs; s->n_keys = h.n_keys; s->n_buckets = h.n_buckets;
    s->arena_words = h.arena_words; s->n_inline = h.n_inline; s->n_lists = h.n_lists;
    s->max_list = h.max_list; s->n_displaced = h.n_displaced; s->shard = h.shard;
    s->n_shards = h.n_shards; s->max_protein_id = h.max_protein_id;
}

int kaamer_image_get_stats(const kaamer_image *img, kaamer_image_stats *out)
{
    if (!img || !out) return kaamer_fail(KAAMER_E_ARG, "image_get_stats: bad argument");
    fill_stats(img->hdr, out);
    return KAAMER_OK;
}

void kaamer_stats_from_header(const kh_image_header *h, kaamer_image_stats *out) { fill_stats(*h, out); }

void kaamer_image_free(kaamer_image *img)
{
    if (!img) return;
    free(img->buckets);
    free(img->arena);
    delete img;
}

uint32_t kaamer_image_get(const kaamer_image *img, uint32_t key, uint32_t *ids, uint32_t cap)
{
    if (!img || img->hdr.n_buckets == 0) return 0;
    if (img->hdr.n_shards > 1 && kh_shard_of(key, img->hdr.n_shards) != img->hdr.shard) return 0;
    uint64_t nb = img->hdr.n_buckets, b = kh_home_bucket(key, img->hdr.n_shards, nb);
    for (uint64_t tries = 0; tries < nb; tries++) {
        const kh_bucket &bk = img->buckets[b];
        bool has_empty = false;
        for (int s = 0; s < KH_SLOTS_PER_BUCKET; s++) {
            if (bk.s[s].key == key) {
                uint32_t v = bk.s[s].val;
                if (v & KH_INLINE_BIT) { if (ids && cap) ids[0] = v & ~KH_INLINE_BIT; return 1; }
                const uint32_t *l = img->arena + (uint64_t)v * 4;
                for (uint32_t t = 0; t < l[0] && t < cap && ids; t++) ids[t] = l[1 + t];
                return l[0];
            }
            if (bk.s[s].key == KH_EMPTY_KEY) has_empty = true;
        }
        if (has_empty) return 0;
        b = (b + 1 == nb) ? 0 : b + 1;
    }
    return 0;
}

// EXPERIMENT (tools/r4_arena_order.sh, not declared in the public header): re-orders the postings lists of `img` by FIRST TOUCH
// -- the order in which a walk over the proteins (input order) and their windows (ascending position) meets them -- so that
// the lists behind consecutive windows of a protein sit in consecutive 16-byte units (four list heads per 64-byte sector).
int kaamer_exp_relayout_first_touch(kaamer_image *img, const uint8_t *seqs, const uint64_t *offsets, uint32_t n_proteins)
{
    if (!img || !offsets || (!seqs && n_proteins)) return kaamer_fail(KAAMER_E_ARG, "exp_relayout: bad argument");
    const uint64_t units = img->hdr.arena_words / 4;
    std::vector<uint32_t> new_off((size_t)units + 1, 0u);
    std::vector<uint32_t> out((size_t)img->hdr.arena_words, 0u);
    uint32_t cur = 1;
    const uint64_t nb = img->hdr.n_buckets;
    auto place = [&](uint32_t old) {
        if (new_off[old]) return;
        const uint32_t cnt = img->arena[(uint64_t)old * 4], u = (1u + cnt + 3u) / 4u;
        memcpy(&out[(size_t)cur * 4], &img->arena[(size_t)old * 4], (size_t)u * 16);
        new_off[old] = cur;
        cur += u;
    };
    for (uint32_t p = 0; p < n_proteins; p++) {
        const uint8_t *s = seqs + offsets[p];
        const uint64_t len = offsets[p + 1] - offsets[p];
        for (uint64_t i = 0; i + KAAMER_KMER_SIZE <= len; i++) {
            const uint32_t key = kaamer_encode_kmer(s + i);
            if (img->hdr.n_shards > 1 && kh_shard_of(key, img->hdr.n_shards) != img->hdr.shard) continue;
            uint64_t b = kh_home_bucket(key, img->hdr.n_shards, nb);
            for (uint64_t tries = 0; tries < nb; tries++) {
                const kh_bucket &bk = img->buckets[b];
                bool done = false, has_empty = false;
                for (int t = 0; t < KH_SLOTS_PER_BUCKET; t++) {
                    if (bk.s[t].key == key) { if (!(bk.s[t].val & KH_INLINE_BIT) && bk.s[t].val) place(bk.s[t].val); done = true; break; }
                    if (bk.s[t].key == KH_EMPTY_KEY) has_empty = true;
                }
                if (done || has_empty) break;
                b = (b + 1 == nb) ? 0 : b + 1;
            }
        }
    }
    for (uint64_t b = 0; b < nb; b++)   // (lists no window reached: none, every key came from a window; kept for safety)
        for (int t = 0; t < KH_SLOTS_PER_BUCKET; t++) {
            kh_slot &sl = img->buckets[b].s[t];
            if (sl.key != KH_EMPTY_KEY && !(sl.val & KH_INLINE_BIT) && sl.val) { place(sl.val); sl.val = new_off[sl.val]; }
        }
    memcpy(img->arena, out.data(), (size_t)img->hdr.arena_words * 4);
    return KAAMER_OK;
}

}  // extern "C"
