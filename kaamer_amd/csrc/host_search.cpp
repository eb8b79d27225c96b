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

#include <errno.h>
#include <fcntl.h>
#include <unistd.h>

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

// GetQueriesFasta / GetQueriesFastq (search.go:222-412) as a state machine over lines, so that a whole buffer and a file
// read in pieces go through ONE statement of the rules:
//   FASTA  blank lines are skipped (:285-287); a '>' line closes the pending record and names the next; sequence lines
//          are TrimSpace'd and concatenated (:309); every record but the LAST of the input is upper-cased (:295 vs
//          :313-320); SizeInKmer = len - 6, minus 1 when the sequence ends in '*' (:290-293)
//   FASTQ  a line starting with '@' opens a record (also a quality line that happens to start with '@'); a line
//          matching ^[ATGCNatgcn]+$ REPLACES the record's sequence; no case change; SizeInKmer = len - 6 (:395,407)
//   both   the first Query of the input is built with Location{PlusStrand: true}, every following one as
//          Query{Sequence: "", ...} (search.go:297,399): Location is then Go's zero value, PlusStrand false
struct RecordParser {
    bool fastq = false;
    bool first_done = false;
    std::string seq, name;

    void push(kaamer_reads *r, bool upper, bool star_rule)
    {
        int32_t size = (int32_t)seq.size() - KAAMER_KMER_SIZE + 1;          // search.go:290,314,395,407
        if (star_rule && !seq.empty() && seq.back() == '*') size--;         // search.go:291-293,315-317
        const size_t at = r->seqs.size();
        r->seqs.resize(at + seq.size());
        if (!upper) {
            memcpy(r->seqs.data() + at, seq.data(), seq.size());
        } else {
            for (size_t i = 0; i < seq.size(); i++) {
                const char c = seq[i];
                r->seqs[at + i] = (uint8_t)((c >= 'a' && c <= 'z') ? c - 32 : c);  // strings.ToUpper (:295)
            }
        }
        r->offsets.push_back(r->seqs.size());
        r->size_in_kmer.push_back(size);
        r->names.insert(r->names.end(), name.begin(), name.end());
        r->name_off.push_back(r->names.size());
        r->plus_strand.push_back(first_done ? 0 : 1);
        first_done = true;
    }

    void line(kaamer_reads *r, const char *b, const char *e)
    {
        if (e - b < 1) return;
        if (!fastq) {
            if (*b == '>') {
                if (!seq.empty()) { push(r, /*upper=*/true, /*star_rule=*/true); seq.clear(); }
                name.assign(b + 1, e);
            } else {
                while (b < e && is_space(*b)) b++;
                while (e > b && is_space(e[-1])) e--;
                seq.append(b, e);
            }
            return;
        }
        if (*b == '@') {
            if (!seq.empty()) { push(r, false, false); seq.clear(); name.clear(); }
            name.assign(b + 1, e);
        } else {
            static const struct NtTable {   // ^[ATGCNatgcn]+$ (search.go:340,386)
                bool ok[256];
                NtTable() { memset(ok, 0, sizeof ok); for (const char *c = "ATGCNatgcn"; *c; c++) ok[(uint8_t)*c] = true; }
            } nt;
            bool is_seq = true;
            for (const char *c = b; c < e && is_seq; c++) is_seq = nt.ok[(uint8_t)*c];
            if (is_seq) seq.assign(b, e);
        }
    }

    // end of the input: the pending record is the LAST one (FASTA: not upper-cased)
    void finish(kaamer_reads *r)
    {
        if (!seq.empty()) { push(r, false, !fastq); seq.clear(); }
    }
};

kaamer_reads *new_reads()
{
    kaamer_reads *r = new (std::nothrow) kaamer_reads();
    if (r) { r->offsets.push_back(0); r->name_off.push_back(0); }
    return r;
}

// bufio.Scanner with Buffer(buf, 1 MiB) (search.go:273-274, inputFASTA.go:88-89): a line of 1 048 576 bytes or more
// (its '\r' included, its '\n' not) fills the scanner's buffer without a token; Scan() returns false and the reference
// goes on as if the input had ended there
const size_t SCANNER_MAX = 1024u * 1024u;

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

static int parse_buffer(const char *text, uint64_t len, bool fastq, kaamer_reads **out)
{
    std::string inflated;   // (search.go:259-263, 361-366: gzipped input)
    if (kaamer_is_gzip(text, len)) {
        const int zrc = kaamer_gunzip(text, len, &inflated);
        // gzip.NewReader fails on a first header that is no gzip header: the reference prints the error and returns
        // without a query (search.go:259-263) -- an empty read set, not an error of the call
        if (zrc == KAAMER_E_FORMAT) inflated.clear();
        else if (zrc) return zrc;
        text = inflated.data(); len = inflated.size();
    }
    kaamer_reads *r = new_reads();
    if (!r) return kaamer_fail(KAAMER_E_NOMEM, "parse");
    RecordParser ps;
    ps.fastq = fastq;
    LineReader lr{ text, text + len };
    const char *b, *e;
    while (lr.next(b, e)) ps.line(r, b, e);
    ps.finish(r);
    *out = r;
    return KAAMER_OK;
}

// GetQueriesFasta, search.go:222-322 (rules: RecordParser above)
int kaamer_parse_fasta(const char *text, uint64_t len, kaamer_reads **out)
{
    if (!out || (!text && len)) return kaamer_fail(KAAMER_E_ARG, "parse_fasta: bad argument");
    return parse_buffer(text, len, false, out);
}

// GetQueriesFastq, search.go:324-412
int kaamer_parse_fastq(const char *text, uint64_t len, kaamer_reads **out)
{
    if (!out || (!text && len)) return kaamer_fail(KAAMER_E_ARG, "parse_fastq: bad argument");
    return parse_buffer(text, len, true, out);
}

// ---- the same readers over a FILE, in chunks: search.go:240-283 opens the file, sniffs the first 32 bytes, wraps a
// gzip.Reader around it when they carry the gzip signature and scans line by line; a 100 M-read file is never in
// memory as a whole (configs[4]).  kaamer_reader_next hands out up to max_seqs records / about max_bytes of sequence
// as a kaamer_reads (the accessors below), which goes straight into kaamer_stream_push.
struct kaamer_reader {
    int fd = -1;
    bool own_fd = false;
    bool gz = false, src_eof = false, ended = false, strict = false;
    z_stream z;
    bool z_live = false;
    std::vector<uint8_t> raw;     // bytes as read from the file (gzip only)
    size_t raw_pos = 0, raw_len = 0;
    std::vector<char> text;       // decoded text not yet split into lines
    size_t text_pos = 0, text_len = 0;
    std::string carry;            // a line that began in an earlier block
    RecordParser ps;
    uint64_t n_records = 0;
};

static ssize_t fd_read(int fd, void *dst, size_t n)
{
    for (;;) {
        const ssize_t k = read(fd, dst, n);
        if (k < 0 && errno == EINTR) continue;
        return k;
    }
}

// fills rd->text with the next block of decoded text; false: the input is over (EOF, a gzip stream that broke off or is
// damaged: what was read before stays -- gzip.Reader returns the error and the scanner stops)
static bool reader_fill(kaamer_reader *rd)
{
    rd->text_pos = rd->text_len = 0;
    if (rd->src_eof) return false;
    if (!rd->gz) {
        const ssize_t k = fd_read(rd->fd, rd->text.data(), rd->text.size());
        if (k <= 0) { rd->src_eof = true; return false; }
        rd->text_len = (size_t)k;
        return true;
    }
    for (;;) {
        if (rd->raw_pos == rd->raw_len) {
            const ssize_t k = fd_read(rd->fd, rd->raw.data(), rd->raw.size());
            rd->raw_pos = 0;
            rd->raw_len = k > 0 ? (size_t)k : 0;
        }
        const bool file_over = rd->raw_len == 0;
        rd->z.next_in = rd->raw.data() + rd->raw_pos;
        rd->z.avail_in = (uInt)(rd->raw_len - rd->raw_pos);
        rd->z.next_out = (Bytef *)rd->text.data();
        rd->z.avail_out = (uInt)rd->text.size();
        const int rc = inflate(&rd->z, Z_NO_FLUSH);
        rd->raw_pos = rd->raw_len - rd->z.avail_in;
        const size_t got = rd->text.size() - rd->z.avail_out;
        if (rc == Z_STREAM_END) {
            // another member may follow (gzip.Reader is multistream); bytes that are no gzip header end the text here
            if (rd->raw_pos == rd->raw_len) {
                const ssize_t k = fd_read(rd->fd, rd->raw.data(), rd->raw.size());
                rd->raw_pos = 0;
                rd->raw_len = k > 0 ? (size_t)k : 0;
            }
            if (rd->raw_len == rd->raw_pos || inflateReset(&rd->z) != Z_OK) rd->src_eof = true;
            if (got) { rd->text_len = got; return true; }
            if (rd->src_eof) return false;
            continue;
        }
        if (rc != Z_OK && !(rc == Z_BUF_ERROR && !file_over && got == 0 && rd->z.avail_in == 0)) rd->src_eof = true;  // broke off / damaged
        if (file_over && got == 0) rd->src_eof = true;
        if (got) { rd->text_len = got; return true; }
        if (rd->src_eof) return false;
    }
}

int kaamer_reader_open_fd(int fd, int format, int strict_scanner, kaamer_reader **out)
{
    if (!out || fd < 0 || (format != 0 && format != 1)) return kaamer_fail(KAAMER_E_ARG, "reader_open: bad argument (format 0 = FASTA, 1 = FASTQ)");
    *out = nullptr;
    kaamer_reader *rd = new (std::nothrow) kaamer_reader();
    if (!rd) return kaamer_fail(KAAMER_E_NOMEM, "reader");
    rd->fd = fd;
    rd->ps.fastq = format == 1;
    rd->strict = strict_scanner != 0;
    rd->text.resize(4u << 20);
    // the first 32 bytes decide how the file is read (http.DetectContentType, search.go:246-270)
    uint8_t head[32];
    size_t nh = 0;
    while (nh < sizeof head) {
        const ssize_t k = fd_read(fd, head + nh, sizeof head - nh);
        if (k <= 0) break;
        nh += (size_t)k;
    }
    if (nh == 0) { rd->src_eof = true; rd->ended = true; }   // (the reference exits on the read error of an empty file)
    if (kaamer_is_gzip((const char *)head, nh)) {
        rd->gz = true;
        rd->raw.resize(1u << 20);
        memcpy(rd->raw.data(), head, nh);
        rd->raw_len = nh;
        memset(&rd->z, 0, sizeof rd->z);
        if (inflateInit2(&rd->z, 16 + MAX_WBITS) != Z_OK) { delete rd; return kaamer_fail(KAAMER_E_NOMEM, "gzip: inflateInit2"); }
        rd->z_live = true;
    } else {
        if (rd->strict) {
            // anything DetectContentType does not call "text/plain; charset=utf-8" yields no query (search.go:266-270):
            // a UTF-16 / UTF-32 byte-order mark, or a byte the sniffer counts as binary among the first 32
            bool text_utf8 = !(nh >= 2 && ((head[0] == 0xFE && head[1] == 0xFF) || (head[0] == 0xFF && head[1] == 0xFE)));
            for (size_t i = 0; i < nh && text_utf8; i++) {
                const uint8_t c = head[i];
                if (c <= 0x08 || c == 0x0B || (c >= 0x0E && c <= 0x1A) || (c >= 0x1C && c <= 0x1F)) text_utf8 = false;
            }
            if (!text_utf8) { rd->src_eof = true; rd->ended = true; }
        }
        memcpy(rd->text.data(), head, nh);
        rd->text_len = nh;
    }
    *out = rd;
    return KAAMER_OK;
}

int kaamer_reader_open(const char *path, int format, int strict_scanner, kaamer_reader **out)
{
    if (!path || !out) return kaamer_fail(KAAMER_E_ARG, "reader_open: bad argument");
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return kaamer_fail(KAAMER_E_IO, "reader_open: cannot open %s", path);
    const int rc = kaamer_reader_open_fd(fd, format, strict_scanner, out);
    if (rc) { close(fd); return rc; }
    (*out)->own_fd = true;
    return KAAMER_OK;
}

int kaamer_reader_next(kaamer_reader *rd, uint32_t max_seqs, uint64_t max_bytes, kaamer_reads **out)
{
    if (!rd || !out || max_seqs == 0) return kaamer_fail(KAAMER_E_ARG, "reader_next: bad argument");
    *out = nullptr;
    kaamer_reads *r = new_reads();
    if (!r) return kaamer_fail(KAAMER_E_NOMEM, "reader_next");
    while (!rd->ended && r->size_in_kmer.size() < max_seqs && r->seqs.size() < max_bytes) {
        if (rd->text_pos == rd->text_len && !reader_fill(rd)) {
            // the end of the input: an unterminated last line, then the pending record as the LAST one
            if (!rd->carry.empty()) {
                if (!(rd->strict && rd->carry.size() >= SCANNER_MAX)) {
                    size_t n = rd->carry.size();
                    if (n && rd->carry[n - 1] == '\r') n--;
                    rd->ps.line(r, rd->carry.data(), rd->carry.data() + n);
                }
                rd->carry.clear();
            }
            rd->ps.finish(r);
            rd->ended = true;
            break;
        }
        const char *p = rd->text.data() + rd->text_pos, *end = rd->text.data() + rd->text_len;
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        if (!nl) {   // the line goes on in the next block
            rd->carry.append(p, end);
            rd->text_pos = rd->text_len;
            if (rd->strict && rd->carry.size() >= SCANNER_MAX) {   // the scanner gives up here: the pending record is the last
                rd->carry.clear();
                rd->ps.finish(r);
                rd->ended = true;
            }
            continue;
        }
        rd->text_pos = (size_t)(nl + 1 - rd->text.data());
        const char *b = p, *e = nl;
        if (!rd->carry.empty()) {
            rd->carry.append(p, nl);
            b = rd->carry.data();
            e = b + rd->carry.size();
        }
        if (rd->strict && (size_t)(e - b) >= SCANNER_MAX) {
            rd->carry.clear();
            rd->ps.finish(r);
            rd->ended = true;
            break;
        }
        if (e > b && e[-1] == '\r') e--;
        rd->ps.line(r, b, e);
        rd->carry.clear();
    }
    rd->n_records += r->size_in_kmer.size();
    *out = r;
    return KAAMER_OK;
}

int kaamer_reader_done(const kaamer_reader *rd) { return (!rd || rd->ended) ? 1 : 0; }
uint64_t kaamer_reader_records(const kaamer_reader *rd) { return rd ? rd->n_records : 0; }

void kaamer_reader_close(kaamer_reader *rd)
{
    if (!rd) return;
    if (rd->z_live) inflateEnd(&rd->z);
    if (rd->own_fd && rd->fd >= 0) close(rd->fd);
    delete rd;
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
