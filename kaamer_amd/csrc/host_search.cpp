// host_search.cpp — host-side pieces of pkg/search kept next to the kernels so that
// non-Go callers get the reference's behaviour bit for bit:
//
//   kaamer_parse_fasta / kaamer_parse_fastq   GetQueriesFasta / GetQueriesFastq
//                                             (search.go:222-412), from a text buffer
//   kaamer_set_best_start_codon               SetBestStartCodon (dna.go:198-272), driven by
//                                             the per-hit lowest matching position
//
// In the Go integration these stay the reference's own Go code.
#include "kaamer_internal.h"

#include <zlib.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

struct kaamer_reads {
    std::vector<uint8_t> seqs;
    std::vector<uint64_t> offsets;  // n + 1
    std::vector<int32_t> size_in_kmer;
    std::vector<char> names;
    std::vector<uint64_t> name_off;  // n + 1
    std::vector<int32_t> plus_strand;  // Location.PlusStrand as the reference's reader leaves it
};

namespace {

// bufio.Scanner with ScanLines: split at '\n', drop one trailing '\r'
struct LineReader {
    const char *p, *end;
    bool next(const char *&b, const char *&e)
    {
        if (p >= end) return false;
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        b = p;
        e = nl ? nl : end;
        p = nl ? nl + 1 : end;
        if (e > b && e[-1] == '\r') e--;
        return true;
    }
};

inline bool is_space(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\v' || c == '\f' || c == '\r'; }

void push_record(kaamer_reads *r, const std::string &seq, const std::string &name, bool upper, bool star_rule)
{
    int32_t size = (int32_t)seq.size() - KAAMER_KMER_SIZE + 1;          // search.go:290,314,395,407
    if (star_rule && !seq.empty() && seq.back() == '*') size--;         // search.go:291-293,315-317
    for (char c : seq) r->seqs.push_back((uint8_t)((upper && c >= 'a' && c <= 'z') ? c - 32 : c));  // strings.ToUpper (:295)
    r->offsets.push_back(r->seqs.size());
    r->size_in_kmer.push_back(size);
    r->names.insert(r->names.end(), name.begin(), name.end());
    r->name_off.push_back(r->names.size());
    // Both readers build the first Query with Location{PlusStrand: true} and every following one as
    // Query{Sequence: "", ...} (search.go:297,399): Location is then Go's zero value, PlusStrand false.  Protein results
    // report that field as it is (search_protein.go never sets it); for nucleotide input GetORFs overwrites it.
    r->plus_strand.push_back(r->plus_strand.empty() ? 1 : 0);
}

}  // namespace

bool kaamer_is_gzip(const char *text, uint64_t len)
{
    return len >= 3 && (uint8_t)text[0] == 0x1F && (uint8_t)text[1] == 0x8B && (uint8_t)text[2] == 0x08;
}

int kaamer_gunzip(const char *text, uint64_t len, std::string *out)
{
    out->clear();
    z_stream z;
    memset(&z, 0, sizeof z);
    if (inflateInit2(&z, 16 + MAX_WBITS) != Z_OK) return kaamer_fail(KAAMER_E_NOMEM, "gzip: inflateInit2");
    z.next_in = (Bytef *)const_cast<char *>(text);
    uint64_t left = len;
    bool first_header_seen = false;
    std::string buf(1u << 20, '\0');
    for (;;) {
        if (z.avail_in == 0 && left) {
            const uInt take = left > (1u << 30) ? (1u << 30) : (uInt)left;
            z.avail_in = take;
            left -= take;
        }
        z.next_out = (Bytef *)&buf[0];
        z.avail_out = (uInt)buf.size();
        const int rc = inflate(&z, Z_NO_FLUSH);
        const size_t got = buf.size() - z.avail_out;
        if (got) { out->append(buf.data(), got); first_header_seen = true; }
        if (rc == Z_STREAM_END) {
            first_header_seen = true;
            if (z.avail_in == 0 && left == 0) break;          // the last member ended with the input
            // another member follows (gzip.Reader multistream); bytes that are not a gzip header end the text here
            const Bytef *next = z.next_in;
            const uInt avail = z.avail_in;
            if (inflateReset(&z) != Z_OK) break;
            z.next_in = const_cast<Bytef *>(next);
            z.avail_in = avail;
            continue;
        }
        if (rc == Z_OK) continue;
        if (rc == Z_BUF_ERROR && z.avail_in == 0 && left == 0) break;   // the stream breaks off: what was read stays
        break;                                                           // damaged data: the same
    }
    inflateEnd(&z);
    if (!first_header_seen && out->empty()) {
        // (an empty member is fine; a first header that is no gzip header is gzip.NewReader's error)
        z_stream t;
        memset(&t, 0, sizeof t);
        bool ok = false;
        if (inflateInit2(&t, 16 + MAX_WBITS) == Z_OK) {
            Bytef sink[64];
            t.next_in = (Bytef *)const_cast<char *>(text);
            t.avail_in = len > 4096 ? 4096u : (uInt)len;
            t.next_out = sink; t.avail_out = sizeof sink;
            const int rc = inflate(&t, Z_NO_FLUSH);
            ok = rc == Z_OK || rc == Z_STREAM_END || rc == Z_BUF_ERROR;
            inflateEnd(&t);
        }
        if (!ok) return kaamer_fail(KAAMER_E_FORMAT, "gzip: invalid header");
    }
    return KAAMER_OK;
}

extern "C" {

// GetQueriesFasta, search.go:222-322, on decompressed text.  Reference behaviours kept:
// blank lines are skipped (:285-287); sequence lines are TrimSpace'd and concatenated
// (:309); every record but the LAST is upper-cased (:295 vs :313-320); SizeInKmer = len-6,
// minus 1 when the sequence ends in '*'.
int kaamer_parse_fasta(const char *text, uint64_t len, kaamer_reads **out)
{
    if (!out || (!text && len)) return kaamer_fail(KAAMER_E_ARG, "parse_fasta: bad argument");
    std::string inflated;   // (search.go:259-263, 361-366: gzipped input)
    if (kaamer_is_gzip(text, len)) {
        const int zrc = kaamer_gunzip(text, len, &inflated);
        if (zrc) return zrc;
        text = inflated.data(); len = inflated.size();
    }
    kaamer_reads *r = new (std::nothrow) kaamer_reads();
    if (!r) return kaamer_fail(KAAMER_E_NOMEM, "parse_fasta");
    r->offsets.push_back(0);
    r->name_off.push_back(0);
    LineReader lr{ text, text + len };
    const char *b, *e;
    std::string seq, name;
    while (lr.next(b, e)) {
        if (e - b < 1) continue;
        if (*b == '>') {
            if (!seq.empty()) {
                push_record(r, seq, name, /*upper=*/true, /*star_rule=*/true);
                seq.clear();
            }
            name.assign(b + 1, e);
        } else {
            while (b < e && is_space(*b)) b++;
            while (e > b && is_space(e[-1])) e--;
            seq.append(b, e);
        }
    }
    if (!seq.empty()) push_record(r, seq, name, /*upper=*/false, /*star_rule=*/true);
    *out = r;
    return KAAMER_OK;
}

// GetQueriesFastq, search.go:324-412: a line starting with '@' opens a record (also a
// quality line that happens to start with '@'); a line matching ^[ATGCNatgcn]+$ REPLACES
// the record's sequence; no case change; SizeInKmer = len-6.
int kaamer_parse_fastq(const char *text, uint64_t len, kaamer_reads **out)
{
    if (!out || (!text && len)) return kaamer_fail(KAAMER_E_ARG, "parse_fastq: bad argument");
    std::string inflated;   // (search.go:259-263, 361-366: gzipped input)
    if (kaamer_is_gzip(text, len)) {
        const int zrc = kaamer_gunzip(text, len, &inflated);
        if (zrc) return zrc;
        text = inflated.data(); len = inflated.size();
    }
    kaamer_reads *r = new (std::nothrow) kaamer_reads();
    if (!r) return kaamer_fail(KAAMER_E_NOMEM, "parse_fastq");
    r->offsets.push_back(0);
    r->name_off.push_back(0);
    LineReader lr{ text, text + len };
    const char *b, *e;
    std::string seq, name;
    while (lr.next(b, e)) {
        if (e - b < 1) continue;
        if (*b == '@') {
            if (!seq.empty()) {
                push_record(r, seq, name, false, false);
                seq.clear();
                name.clear();
            }
            name.assign(b + 1, e);
        } else {
            bool is_seq = true;
            for (const char *c = b; c < e && is_seq; c++)
                is_seq = *c == 'A' || *c == 'T' || *c == 'G' || *c == 'C' || *c == 'N' || *c == 'a' || *c == 't' || *c == 'g' || *c == 'c' || *c == 'n';
            if (is_seq) seq.assign(b, e);
        }
    }
    if (!seq.empty()) push_record(r, seq, name, false, false);
    *out = r;
    return KAAMER_OK;
}

uint32_t kaamer_reads_count(const kaamer_reads *r) { return r ? (uint32_t)r->size_in_kmer.size() : 0; }
const uint8_t *kaamer_reads_seqs(const kaamer_reads *r) { return r ? r->seqs.data() : nullptr; }
const uint64_t *kaamer_reads_offsets(const kaamer_reads *r) { return r ? r->offsets.data() : nullptr; }
const int32_t *kaamer_reads_size_in_kmer(const kaamer_reads *r) { return r ? r->size_in_kmer.data() : nullptr; }
const char *kaamer_reads_names(const kaamer_reads *r) { return r ? r->names.data() : nullptr; }
const uint64_t *kaamer_reads_name_offsets(const kaamer_reads *r) { return r ? r->name_off.data() : nullptr; }
const int32_t *kaamer_reads_plus_strand(const kaamer_reads *r) { return r ? r->plus_strand.data() : nullptr; }
void kaamer_reads_free(kaamer_reads *r) { delete r; }

// SetBestStartCodon, dna.go:198-272.  hits must be in sortMapByValue order (Kmatch
// descending; kaamer_sort_hits).  The reference scans PositionHits of the best hits; the only
// thing it reads from them is the lowest matching position of the first best hit and whether
// position 0 matches for the following ones (its `exit` flag is never reset, dna.go:224-237),
// which is exactly hit_first_pos.  Returns the number of residues trimmed from the ORF head
// (0: unchanged) and updates start_position / size_in_kmer like dna.go:252-267.
int32_t kaamer_set_best_start_codon(const uint32_t *kmatch_sorted, const uint32_t *first_pos_sorted, int64_t n_hits,
                                    const int32_t *starts_alt, int32_t n_starts, int32_t plus_strand,
                                    const uint8_t *orf_aa, uint32_t aa_len, int32_t *start_position, int32_t *size_in_kmer)
{
    if (n_starts < 1 || !starts_alt || !start_position || !size_in_kmer) return 0;  // dna.go:210-212
    int64_t best_hit_score = 0;
    int64_t first_best_hit_pos = 999999999;                                        // dna.go:219
    bool exit_ = false;
    for (int64_t h = 0; h < n_hits; h++) {
        if ((int64_t)kmatch_sorted[h] < best_hit_score) continue;                  // dna.go:203-208
        best_hit_score = kmatch_sorted[h];
        // dna.go:225-237 on this best hit's PositionHits
        if (!exit_) {
            if ((int64_t)first_pos_sorted[h] < first_best_hit_pos) first_best_hit_pos = first_pos_sorted[h];
            exit_ = true;
        } else if (first_pos_sorted[h] == 0) {
            first_best_hit_pos = 0;  // only position 0 is examined once exit is set
        }
    }
    int32_t best_start = starts_alt[0];
    const int32_t first_start = starts_alt[0];
    for (int32_t s = 0; s < n_starts; s++) {                                       // dna.go:240-249
        if ((int64_t)starts_alt[s] <= first_best_hit_pos) best_start = starts_alt[s];
        else break;
    }
    if (best_start == first_start) return 0;
    *start_position = plus_strand ? *start_position + 3 * best_start : *start_position - 3 * best_start;  // dna.go:254-258
    const int64_t new_len = (int64_t)aa_len - best_start;
    int32_t s = (int32_t)(new_len - KAAMER_KMER_SIZE + 1);                         // dna.go:263
    if (new_len > 0 && orf_aa && orf_aa[aa_len - 1] == '*') s--;                   // dna.go:264-266
    *size_in_kmer = s;
    return best_start;
}

}  // extern "C"

// ---- host post-steps kept bit-compatible with the reference (moved here from search.hip: pure host code, part of
// the sanitized CPU build, tools/asan) -------------------------------------------------------------------------
extern "C" {

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
