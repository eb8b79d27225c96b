// makedb.cpp — the makedb side of the builder: which proteins a database file contributes, under which ids
// (pkg/makedb/inputFASTA.go, inputTSV.go), and the protein table FetchHitsInformation reads
// (pkg/search/search.go:454-470; pkg/kvstore/protein.proto) — in place of the ProteinStore.
//
//   kaamer_makedb_fasta   runFASTA + processProteinInputFASTA (inputFASTA.go:95-124, 191-250):
//                         every '>' line bumps proteinNb and queues the PREVIOUS entry under the NEW number, so
//                         record k (1-based) gets id k+1 and the last record, queued at EOF, keeps id N: the last
//                         two records share id N (reference behaviour, reproduced); lines before the first header
//                         form an entry of their own (id 1); EntryId = header up to the first space,
//                         ProteinName = the rest; the sequence is upper-cased (ASCII); entries whose name contains
//                         ", partial" and sequences shorter than 7 are dropped (ids are not renumbered)
//   kaamer_makedb_tsv     runTSV (inputTSV.go:92-142): header row names the columns, "entryid" and "sequence"
//                         (any case) are required; a row is accepted when its sequence has >= 7 characters and its
//                         EntryId is not empty, and gets the next 0-based id; the sequence is NOT upper-cased;
//                         every other column is a feature (inputTSV.go:184-190)
// Divergences (documented): an empty line makes the reference's FASTA reader panic (line[0:1]) — skipped here; a TSV
// row with more columns than the header panics (features[i]) — the extra columns are ignored; bytes >= 0x80 are kept
// as they are (Go's ToUpper would rewrite invalid UTF-8); files are taken as text already decompressed.
#include "kaamer_internal.h"

#include <cstdio>
#include <algorithm>
#include <cstring>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

struct kaamer_proteins {
    std::vector<uint32_t> ids;
    std::vector<uint8_t> seqs;
    std::vector<uint64_t> offsets;            // n + 1
    std::vector<char> entry_ids;
    std::vector<uint64_t> entry_off;          // n + 1
    std::vector<std::string> feature_names;   // KStats.Features (inputFASTA.go:41, inputTSV.go:184-190)
    std::vector<char> features;               // values, protein-major, feature order
    std::vector<uint64_t> feature_off;        // n * n_features + 1
    uint64_t n_aa = 0, n_kmers = 0;           // KStats (inputFASTA.go:141-143)
    std::unordered_map<uint32_t, uint32_t> by_id;  // protein id -> record (a later record with the same id wins)
    // EMBL entries whose SQ line declares fewer residues than the entry holds: the reference indexes Sequence[:Length]
    // (inputEMBL.go:309-312) but STORES the whole string as Protein.Sequence (:293-305); `seqs` holds what is indexed,
    // this holds the whole string of those records (rare: real UniProt entries declare what they hold)
    std::unordered_map<uint32_t, std::string> full_seq;  // record -> Protein.Sequence
};

namespace {

// kaamer_makedb_text(strict_scanner = 1): the reference's scanners are bufio.Scanner with Buffer(buf, 1 MiB)
// (inputFASTA.go:88-89, inputTSV.go, inputEMBL.go, inputGBK.go alike): a line of 1 048 576 bytes or more ('\r' included,
// '\n' not) fills the buffer without a token, Scan() returns false and the reader goes on as if the input had ended there
thread_local bool g_strict_scanner = false;

struct Lines {
    const char *p, *end;
    bool next(const char *&b, const char *&e)
    {
        if (p >= end) return false;
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        b = p;
        e = nl ? nl : end;
        if (g_strict_scanner && (size_t)(e - b) >= 1024u * 1024u) { p = end; return false; }
        p = nl ? nl + 1 : end;
        if (e > b && e[-1] == '\r') e--;  // bufio.ScanLines drops one trailing '\r'
        return true;
    }
};

void add_protein(kaamer_proteins *r, uint32_t id, const std::string &entry_id, const std::string &seq, const std::vector<std::string> &feat)
{
    r->by_id[id] = (uint32_t)r->ids.size();
    r->ids.push_back(id);
    r->seqs.insert(r->seqs.end(), seq.begin(), seq.end());
    r->offsets.push_back(r->seqs.size());
    r->entry_ids.insert(r->entry_ids.end(), entry_id.begin(), entry_id.end());
    r->entry_off.push_back(r->entry_ids.size());
    for (size_t i = 0; i < r->feature_names.size(); i++) {
        if (i < feat.size()) r->features.insert(r->features.end(), feat[i].begin(), feat[i].end());
        r->feature_off.push_back(r->features.size());
    }
    r->n_aa += seq.size();
    r->n_kmers += seq.size() - KAAMER_KMER_SIZE + 1;
}

kaamer_proteins *new_proteins()
{
    kaamer_proteins *r = new (std::nothrow) kaamer_proteins();
    if (!r) return nullptr;
    r->offsets.push_back(0);
    r->entry_off.push_back(0);
    r->feature_off.push_back(0);
    return r;
}

std::string lower(const std::string &s)
{
    std::string o = s;
    for (char &c : o) if (c >= 'A' && c <= 'Z') c = (char)(c + 32);
    return o;
}

// processProteinInputFASTA, inputFASTA.go:191-250, on one queued entry
void process_fasta_entry(kaamer_proteins *r, uint32_t id, const std::vector<std::pair<const char *, const char *>> &lines)
{
    std::string entry_id, name, seq;
    for (auto &l : lines) {
        if (l.second - l.first < 1) continue;                                        // :199-201
        if (*l.first == '>') {                                                       // :204-208
            const char *sp = (const char *)memchr(l.first, ' ', (size_t)(l.second - l.first));
            entry_id.assign(l.first + 1, sp ? sp : l.second);
            name = sp ? std::string(sp + 1, l.second) : std::string();
        } else {
            for (const char *c = l.first; c < l.second; c++) seq.push_back((*c >= 'a' && *c <= 'z') ? (char)(*c - 32) : *c);  // :210 ToUpper
        }
    }
    if (name.find(", partial") != std::string::npos) return;                         // :215-217
    if (seq.size() < KAAMER_KMER_SIZE) return;                                        // :222-224
    add_protein(r, id, entry_id, seq, { name });
}

}  // namespace

extern "C" {

int kaamer_makedb_fasta(const char *text, uint64_t len, kaamer_proteins **out)
{
    if (!out || (!text && len)) return kaamer_fail(KAAMER_E_ARG, "makedb_fasta: bad argument");
    *out = nullptr;
    kaamer_proteins *r = new_proteins();
    if (!r) return kaamer_fail(KAAMER_E_NOMEM, "makedb_fasta");
    r->feature_names = { "ProteinName" };                                            // FASTA_DEF_FTS, inputFASTA.go:41
    Lines lr{ text, text + len };
    const char *b, *e;
    uint32_t protein_nb = 0;                                                         // inputFASTA.go:65
    std::vector<std::pair<const char *, const char *>> entry;
    while (lr.next(b, e)) {
        if (e - b < 1) continue;                                                     // (the reference panics on an empty line)
        if (*b == '>') {                                                             // :98-112
            protein_nb++;
            if (!entry.empty()) {
                process_fasta_entry(r, protein_nb, entry);                           // queued under the NEW number
                entry.clear();
            }
        }
        entry.emplace_back(b, e);                                                    // :114-117
    }
    if (!entry.empty()) process_fasta_entry(r, protein_nb, entry);                   // :120-124
    *out = r;
    return KAAMER_OK;
}

int kaamer_makedb_tsv(const char *text, uint64_t len, kaamer_proteins **out)
{
    if (!out || (!text && len)) return kaamer_fail(KAAMER_E_ARG, "makedb_tsv: bad argument");
    *out = nullptr;
    kaamer_proteins *r = new_proteins();
    if (!r) return kaamer_fail(KAAMER_E_NOMEM, "makedb_tsv");
    Lines lr{ text, text + len };
    const char *b, *e;
    auto split = [](const char *b, const char *e) {
        std::vector<std::string> cols;
        const char *s = b;
        for (const char *c = b;; c++)
            if (c == e || *c == '\t') { cols.emplace_back(s, c); s = c + 1; if (c == e) break; }
        return cols;
    };
    std::vector<std::string> header;
    std::vector<int> feat_slot;  // column -> feature index, -1 entryid, -2 sequence
    bool first = true;
    uint32_t protein_nb = 0;                                                         // inputTSV.go:62
    while (lr.next(b, e)) {
        if (first) {                                                                 // :94-115
            header = split(b, e);
            bool has_id = false, has_seq = false;
            for (auto &f : header) {
                const std::string lf = lower(f);
                if (lf == "entryid") { has_id = true; feat_slot.push_back(-1); }
                else if (lf == "sequence") { has_seq = true; feat_slot.push_back(-2); }
                else { feat_slot.push_back((int)r->feature_names.size()); r->feature_names.push_back(f); }
            }
            if (!has_id || !has_seq) {
                delete r;
                return kaamer_fail(KAAMER_E_FORMAT, has_id ? "TSV file doesn't contain 'Sequence' header" : "TSV file doesn't contain 'EntryID' header");
            }
            first = false;
            continue;
        }
        const std::vector<std::string> cols = split(b, e);                           // :118-119
        std::string entry_id, seq;
        std::vector<std::string> feat(r->feature_names.size());
        for (size_t i = 0; i < cols.size() && i < header.size(); i++) {              // :125-134
            if (feat_slot[i] == -1) entry_id = cols[i];
            else if (feat_slot[i] == -2) seq = cols[i];
            else feat[(size_t)feat_slot[i]] = cols[i];
        }
        if (seq.size() < KAAMER_KMER_SIZE || entry_id.empty()) continue;             // :137-139
        add_protein(r, protein_nb, entry_id, seq, feat);                             // :140-141: ids count the accepted rows
        protein_nb++;
    }
    if (first) { delete r; return kaamer_fail(KAAMER_E_FORMAT, "TSV file doesn't contain 'EntryID' header"); }
    *out = r;
    return KAAMER_OK;
}

}  // extern "C"

// ---- EMBL / GenBank flat files (pkg/makedb/inputEMBL.go, inputGBK.go) ---------------------------------------
// Both scan loops cut the file at lines that are exactly "//" (inputEMBL.go:95-113, inputGBK.go:94-112): every such
// line bumps proteinNb, and the text gathered since the previous one -- if there is any -- is queued under that
// number, so ids are the 1-based ordinal of the record's terminator (an empty record uses up a number).  Text after
// the last "//" is never queued.  (offset / length, the split-build options, are not reproduced: one file, all of it.)
namespace {

const std::string *find_feat(const std::vector<std::pair<std::string, std::string>> &f, const char *k)
{
    for (auto &kv : f) if (kv.first == k) return &kv.second;
    return nullptr;
}
std::string &feat_ref(std::vector<std::pair<std::string, std::string>> &f, const char *k)
{
    for (auto &kv : f) if (kv.first == k) return kv.second;
    f.emplace_back(k, std::string());
    return f.back().second;
}
std::string trim_right(std::string v, char c) { while (!v.empty() && v.back() == c) v.pop_back(); return v; }
std::vector<std::string> fields_of(const std::string &v)  // strings.Fields (ASCII white space)
{
    std::vector<std::string> out;
    size_t i = 0;
    while (i < v.size()) {
        while (i < v.size() && (v[i] == ' ' || (v[i] >= '\t' && v[i] <= '\r'))) i++;
        size_t j = i;
        while (j < v.size() && !(v[j] == ' ' || (v[j] >= '\t' && v[j] <= '\r'))) j++;
        if (j > i) out.emplace_back(v, i, j - i);
        i = j;
    }
    return out;
}
// regexp `<open>.*<close>` replaced by "": from the first `open` that has a `close` after it to the LAST `close`
std::string drop_greedy(const std::string &v, const char *open, const char *close)
{
    const size_t a = v.find(open);
    if (a == std::string::npos) return v;
    const size_t b = v.rfind(close);
    if (b == std::string::npos || b < a + strlen(open)) return v;
    return v.substr(0, a) + v.substr(b + strlen(close));
}
struct Skip {};  // where the Go code would panic (a slice past the end of a short line, [0] of an empty field list)
std::string from(const std::string &l, size_t a) { if (a > l.size()) throw Skip(); return l.substr(a); }

// processProteinInputEMBL, inputEMBL.go:189-314.  false: the entry contributes nothing
bool process_embl_entry(const std::string &entry, std::string &entry_id, std::string &seq, std::vector<std::pair<std::string, std::string>> &feat, size_t &declared)
{
    long long length = 0;
    size_t p0 = 0;
    while (p0 <= entry.size()) {
        size_t nl = entry.find('\n', p0);
        if (nl == std::string::npos) nl = entry.size();
        const std::string l = entry.substr(p0, nl - p0);
        p0 = nl + 1;
        if (l.size() < 2) continue;                                                   // :201-203
        const std::string tag = l.substr(0, 2);
        if (tag == "ID") {                                                            // :205-206
            const auto f = fields_of(from(l, 5));
            if (f.empty()) throw Skip();
            entry_id = f[0];
        } else if (tag == "GN") {                                                     // :207-211
            const std::string *g = find_feat(feat, "GeneName");
            if ((!g || g->empty()) && l.find("Name=") != std::string::npos) {
                const auto f = fields_of(from(l, 5));
                if (f.empty()) throw Skip();
                feat_ref(feat, "GeneName") = trim_right(from(f[0], 5), ';');
            }
        } else if (tag == "DE") {                                                     // :212-229
            const std::string r = from(l, 5);
            if (r.find("RecName") != std::string::npos) {
                feat_ref(feat, "ProteinName") = trim_right(drop_greedy(from(l, 19), " {", "};"), ';');
            } else if (r.find("SubName") != std::string::npos) {
                const std::string v = trim_right(drop_greedy(from(l, 19), " {", "};"), ';');
                std::string &pn = feat_ref(feat, "ProteinName");
                if (!pn.empty()) pn += ";;" + v; else pn = v;
            } else if (r.find("EC=") != std::string::npos) {
                feat_ref(feat, "EC") = trim_right(drop_greedy(from(l, 17), " {", "};"), ';');
            } else if (r.find("Flags: Fragment;") != std::string::npos) {
                return false;                                                         // protein fragments are skipped
            }
        } else if (tag == "OX") {                                                     // :230-232 (the [12:] eats the id's first digit: kept)
            const auto f = fields_of(from(l, 5));
            if (f.empty()) throw Skip();
            feat_ref(feat, "TaxId") = trim_right(from(f[0], 12), ';');
        } else if (tag == "OS") {                                                     // :233-239
            const bool had = find_feat(feat, "Organism") != nullptr;
            std::string &o = feat_ref(feat, "Organism");
            if (had) o += " ";
            o += trim_right(from(l, 5), '.');
        } else if (tag == "OC") {                                                     // :240-244
            std::string &t = feat_ref(feat, "FullTaxonomy");
            if (!t.empty()) t += " ";
            t += from(l, 5);
        } else if (tag == "DR") {                                                     // :245-281
            const auto f = fields_of(from(l, 5));
            if (f.empty()) throw Skip();
            const char *key = f[0] == "KEGG;" ? "KEGG_ID" : f[0] == "GO;" ? "GO" : f[0] == "BioCyc;" ? "BioCyc_ID" : f[0] == "HAMAP;" ? "HAMAP" : nullptr;
            if (key) {
                if (f.size() < 2) throw Skip();
                const bool had = find_feat(feat, key) != nullptr;
                std::string &v = feat_ref(feat, key);
                if (had) v += ";";
                v += trim_right(f[1], ';');
            }
        } else if (tag == "SQ") {                                                     // :282-285
            const auto f = fields_of(from(l, 5));
            if (f.size() < 2) throw Skip();
            length = 0;                                                               // strconv.Atoi, error ignored
            size_t i = 0;
            bool neg = false, ok = !f[1].empty();
            if (ok && (f[1][0] == '+' || f[1][0] == '-')) { neg = f[1][0] == '-'; i = 1; ok = f[1].size() > 1; }
            long long v = 0;
            for (; ok && i < f[1].size(); i++) {
                if (f[1][i] < '0' || f[1][i] > '9' || v > 100000000000ll) { ok = false; break; }
                v = v * 10 + (f[1][i] - '0');
            }
            if (ok) length = neg ? -v : v;
        } else if (tag == "  ") {                                                     // :287-288
            for (char c : from(l, 5)) if (c != ' ') seq.push_back(c);
        }
    }
    if (length < KAAMER_KMER_SIZE) return false;                                      // :293-295: the DECLARED length
    if ((unsigned long long)length > seq.size()) throw Skip();                       // (the k-mer loop :309-312 would slice past the end)
    declared = (size_t)length;                                                        // windows of Sequence[:Length]; the whole string is stored
    return true;
}

// processProteinInputGBK, inputGBK.go:186-301
bool process_gbk_entry(const std::string &entry, std::string &entry_id, std::string &seq, std::vector<std::pair<std::string, std::string>> &feat)
{
    int inside = 0;
    size_t p0 = 0;
    while (p0 <= entry.size()) {
        size_t nl = entry.find('\n', p0);
        if (nl == std::string::npos) nl = entry.size();
        const std::string l = entry.substr(p0, nl - p0);
        p0 = nl + 1;
        if (l.size() < 2) continue;                                                   // :206-208
        size_t a = 0, b = l.size();
        while (a < b && l[a] == ' ') a++;
        while (b > a && l[b - 1] == ' ') b--;
        const size_t sp = l.find(' ', a);
        const std::string tok = l.substr(a, (sp == std::string::npos || sp > b ? b : sp) - a);  // Split(Trim(l, " "), " ")[0]
        if (tok == "LOCUS" || tok == "ACCESSION" || tok == "KEYWORDS" || tok == "SOURCE" || tok == "COMMENT" || tok == "REFERENCE" ||
            tok == "DBLINK" || tok == "DBSOURCE") inside = 0;                         // :210-239
        else if (tok == "DEFINITION") inside = 1;
        else if (tok == "VERSION") inside = 2;
        else if (tok == "ORGANISM") inside = 3;
        else if (tok == "FEATURES") inside = 4;
        else if (tok == "ORIGIN") inside = 5;
        else if (tok == "//") inside = 6;
        if (inside == 1) {                                                            // :242-246
            std::string &pn = feat_ref(feat, "ProteinName");
            if (!pn.empty()) pn += " ";
            pn += from(l, 12);
        } else if (inside == 2) {                                                     // :247-248
            const auto f = fields_of(from(l, 12));
            if (f.empty()) throw Skip();
            entry_id = f[0];
        } else if (inside == 3) {                                                     // :249-257
            const std::string *o = find_feat(feat, "Organism");
            if (!o || o->empty()) feat_ref(feat, "Organism") = from(l, 12);
            else {
                std::string &t = feat_ref(feat, "FullTaxonomy");
                if (!t.empty()) t += " ";
                t += from(l, 12);
            }
        } else if (inside == 5) {                                                     // :260-263
            for (char c : from(l, 10)) if (c != ' ') seq.push_back((c >= 'a' && c <= 'z') ? (char)(c - 32) : c);
        }
    }
    const std::string *pn = find_feat(feat, "ProteinName");
    if (pn && pn->find(", partial") != std::string::npos) return false;               // :268-270
    if (seq.size() < KAAMER_KMER_SIZE) return false;                                  // :272-277
    feat_ref(feat, "ProteinName") = drop_greedy(feat_ref(feat, "ProteinName"), " [", "].");  // :279-280
    return true;
}

int makedb_flat(const char *text, uint64_t len, bool gbk, kaamer_proteins **out)
{
    if (!out || (!text && len)) return kaamer_fail(KAAMER_E_ARG, "makedb_%s: bad argument", gbk ? "gbk" : "embl");
    *out = nullptr;
    std::string inflated;   // (inputEMBL.go:76-84, inputGBK.go:75-83: gzipped input; the FASTA and TSV readers have no such branch)
    if (kaamer_is_gzip(text, len)) {
        const int zrc = kaamer_gunzip(text, len, &inflated);
        if (zrc) return zrc;
        text = inflated.data(); len = inflated.size();
    }
    kaamer_proteins *r = new_proteins();
    if (!r) return kaamer_fail(KAAMER_E_NOMEM, "makedb");
    if (gbk) r->feature_names = { "ProteinName", "Organism", "FullTaxonomy" };       // GBK_DEF_FTS, inputGBK.go:41
    else r->feature_names = { "ProteinName", "GeneName", "EC", "GO", "KEGG_ID", "BioCyc_ID", "HAMAP", "Organism", "TaxId", "FullTaxonomy" };  // inputEMBL.go:43
    Lines lr{ text, text + len };
    const char *b, *e;
    uint32_t protein_nb = 0;
    std::string entry;
    while (lr.next(b, e)) {
        if (e - b == 2 && b[0] == '/' && b[1] == '/') {
            protein_nb++;
            if (!entry.empty()) {
                std::string entry_id, seq;
                std::vector<std::pair<std::string, std::string>> feat;
                bool keep = false;
                size_t declared = std::string::npos;
                try { keep = gbk ? process_gbk_entry(entry, entry_id, seq, feat) : process_embl_entry(entry, entry_id, seq, feat, declared); }
                catch (const Skip &) { keep = false; }   // the reference process would have died here: the entry is dropped
                if (keep) {
                    std::vector<std::string> vals;
                    for (auto &n : r->feature_names) { const std::string *v = find_feat(feat, n.c_str()); vals.push_back(v ? *v : std::string()); }
                    if (declared != std::string::npos && declared < seq.size()) {   // indexed: Sequence[:Length]; stored: all of it
                        r->full_seq[(uint32_t)r->ids.size()] = seq;
                        seq.resize(declared);
                    }
                    add_protein(r, protein_nb, entry_id, seq, vals);
                }
                entry.clear();
            }
        } else {
            entry.append(b, e);
            entry.push_back('\n');
        }
    }
    *out = r;
    return KAAMER_OK;
}

}  // namespace

extern "C" {
int kaamer_makedb_embl(const char *text, uint64_t len, kaamer_proteins **out) { return makedb_flat(text, len, false, out); }
int kaamer_makedb_gbk(const char *text, uint64_t len, kaamer_proteins **out) { return makedb_flat(text, len, true, out); }

int kaamer_makedb_text(const char *text, uint64_t len, int32_t format, int32_t strict_scanner, kaamer_proteins **out)
{
    if (format < 0 || format > 3) return kaamer_fail(KAAMER_E_ARG, "makedb_text: format is 0 FASTA, 1 TSV, 2 EMBL, 3 GBK");
    g_strict_scanner = strict_scanner != 0;
    const int rc = format == 0 ? kaamer_makedb_fasta(text, len, out) : format == 1 ? kaamer_makedb_tsv(text, len, out)
                 : format == 2 ? kaamer_makedb_embl(text, len, out) : kaamer_makedb_gbk(text, len, out);
    g_strict_scanner = false;
    return rc;
}
}

extern "C" {
uint32_t kaamer_proteins_count(const kaamer_proteins *p) { return p ? (uint32_t)p->ids.size() : 0; }
const uint32_t *kaamer_proteins_ids(const kaamer_proteins *p) { return p ? p->ids.data() : nullptr; }
const uint8_t *kaamer_proteins_seqs(const kaamer_proteins *p) { return p ? p->seqs.data() : nullptr; }
const uint64_t *kaamer_proteins_offsets(const kaamer_proteins *p) { return p ? p->offsets.data() : nullptr; }
uint32_t kaamer_proteins_n_features(const kaamer_proteins *p) { return p ? (uint32_t)p->feature_names.size() : 0; }
const char *kaamer_proteins_feature_name(const kaamer_proteins *p, uint32_t i)
{
    return (p && i < p->feature_names.size()) ? p->feature_names[i].c_str() : nullptr;
}
void kaamer_proteins_stats(const kaamer_proteins *p, uint64_t out[3])
{
    if (!p || !out) return;
    out[0] = p->ids.size(); out[1] = p->n_aa; out[2] = p->n_kmers;                   // KStats: proteins, AA, k-mers
}
void kaamer_proteins_free(kaamer_proteins *p) { delete p; }

int kaamer_image_build_makedb(const kaamer_proteins *p, uint32_t shard, uint32_t n_shards, double load_factor, kaamer_image **out)
{
    if (!p || !out) return kaamer_fail(KAAMER_E_ARG, "image_build_makedb: bad argument");
    return kaamer_image_build_proteins(p->seqs.data(), p->offsets.data(), p->ids.data(), (uint32_t)p->ids.size(), shard, n_shards,
                                       load_factor, out);
}

// the packed proteins of a table, for the device builder (builder_device.hip)
void kaamer_proteins_raw(const kaamer_proteins *p, const uint8_t **seqs, const uint64_t **offsets, const uint32_t **ids, uint32_t *n)
{
    *seqs = p->seqs.data(); *offsets = p->offsets.data(); *ids = p->ids.data(); *n = (uint32_t)p->ids.size();
}

// FetchHitsInformation, search.go:454-470: the Protein entry of each hit (protein.proto: EntryId, Sequence,
// Length, Features), from the table instead of one ProteinStore point read per hit
int kaamer_fetch_hits(const kaamer_proteins *p, const uint32_t *ids, uint32_t n, kaamer_protein_entry *out)
{
    if (!p || (n && (!ids || !out))) return kaamer_fail(KAAMER_E_ARG, "fetch_hits: bad argument");
    const size_t nf = p->feature_names.size();
    for (uint32_t i = 0; i < n; i++) {
        kaamer_protein_entry &o = out[i];
        memset(&o, 0, sizeof o);
        const auto it = p->by_id.find(ids[i]);
        if (it == p->by_id.end()) continue;  // the reference stops at the first missing id (search.go:461-463); callers see found = 0
        const uint32_t r = it->second;
        o.found = 1;
        o.length = (uint32_t)(p->offsets[r + 1] - p->offsets[r]);
        o.entry_id = p->entry_ids.data() + p->entry_off[r];
        o.entry_id_len = (uint32_t)(p->entry_off[r + 1] - p->entry_off[r]);
        o.sequence = p->seqs.data() + p->offsets[r];
        o.sequence_len = o.length;
        const auto fs = p->full_seq.find(r);
        if (fs != p->full_seq.end()) {   // Protein.Sequence is the whole string, Protein.Length the declared one (inputEMBL.go:293-305)
            o.sequence = reinterpret_cast<const uint8_t *>(fs->second.data());
            o.sequence_len = (uint32_t)fs->second.size();
        }
        o.n_features = (uint32_t)nf;
        o.features = p->features.data();
        o.feature_off = p->feature_off.data() + (size_t)r * nf;
    }
    return KAAMER_OK;
}

// the protein table as a file next to the index image (what a server opens instead of the ProteinStore)
int kaamer_proteins_save(const kaamer_proteins *p, const char *path)
{
    if (!p || !path) return kaamer_fail(KAAMER_E_ARG, "proteins_save: bad argument");
    FILE *f = fopen(path, "wb");
    if (!f) return kaamer_fail(KAAMER_E_IO, "cannot open for writing");
    std::string names;
    for (auto &s : p->feature_names) { names += s; names.push_back('\0'); }
    // "AMRPROT1"; "AMRPROT2" when whole sequences follow (records whose stored Sequence is longer than what is indexed)
    const uint64_t hdr[8] = { p->full_seq.empty() ? 0x31544F5250524D41ull : 0x32544F5250524D41ull, p->ids.size(), p->seqs.size(), p->entry_ids.size(),
                              p->feature_names.size(), p->features.size(), names.size(), p->n_kmers };
    bool ok = fwrite(hdr, sizeof hdr, 1, f) == 1;
    auto put = [&](const void *d, size_t bytes) { if (ok && bytes) ok = fwrite(d, 1, bytes, f) == bytes; };
    put(p->ids.data(), p->ids.size() * 4);
    put(p->offsets.data(), p->offsets.size() * 8);
    put(p->seqs.data(), p->seqs.size());
    put(p->entry_off.data(), p->entry_off.size() * 8);
    put(p->entry_ids.data(), p->entry_ids.size());
    put(names.data(), names.size());
    put(p->feature_off.data(), p->feature_off.size() * 8);
    put(p->features.data(), p->features.size());
    if (!p->full_seq.empty()) {
        std::vector<uint32_t> recs;
        for (auto &kv : p->full_seq) recs.push_back(kv.first);
        std::sort(recs.begin(), recs.end());
        const uint64_t n_full = recs.size();
        put(&n_full, 8);
        for (uint32_t rr : recs) {
            const std::string &fs = p->full_seq.at(rr);
            const uint64_t rec_len[2] = { rr, fs.size() };
            put(rec_len, 16);
            put(fs.data(), fs.size());
        }
    }
    ok = (fclose(f) == 0) && ok;
    return ok ? KAAMER_OK : kaamer_fail(KAAMER_E_IO, "short write");
}

int kaamer_proteins_load(const char *path, kaamer_proteins **out)
{
    if (!path || !out) return kaamer_fail(KAAMER_E_ARG, "proteins_load: bad argument");
    *out = nullptr;
    FILE *f = fopen(path, "rb");
    if (!f) return kaamer_fail(KAAMER_E_IO, "cannot open protein table");
    uint64_t hdr[8];
    if (fread(hdr, sizeof hdr, 1, f) != 1 || (hdr[0] != 0x31544F5250524D41ull && hdr[0] != 0x32544F5250524D41ull)) { fclose(f); return kaamer_fail(KAAMER_E_FORMAT, "not a kaamer protein table"); }
    const bool v2 = hdr[0] == 0x32544F5250524D41ull;
    // sizes must add up to the file before anything is allocated from them
    fseek(f, 0, SEEK_END);
    const unsigned long long size = (unsigned long long)ftell(f);
    const unsigned long long n = hdr[1], nf = hdr[4];
    const unsigned long long want = sizeof hdr + n * 4 + (n + 1) * 8 + hdr[2] + (n + 1) * 8 + hdr[3] + hdr[6] + (n * nf + 1) * 8 + hdr[5];
    if (n > 0xFFFFFFFFull || nf > 65536 || hdr[2] > size || hdr[3] > size || hdr[5] > size || hdr[6] > size || (v2 ? want + 8 > size : want != size)) {
        fclose(f);
        return kaamer_fail(KAAMER_E_FORMAT, "protein table header does not match the file");
    }
    fseek(f, (long)sizeof hdr, SEEK_SET);
    kaamer_proteins *p = new (std::nothrow) kaamer_proteins();
    if (!p) { fclose(f); return kaamer_fail(KAAMER_E_NOMEM, "proteins_load"); }
    bool ok = true;
    auto get = [&](void *d, size_t bytes) { if (ok && bytes) ok = fread(d, 1, bytes, f) == bytes; };
    std::string names((size_t)hdr[6], '\0');
    p->ids.resize((size_t)n); p->offsets.resize((size_t)n + 1); p->seqs.resize((size_t)hdr[2]);
    p->entry_off.resize((size_t)n + 1); p->entry_ids.resize((size_t)hdr[3]);
    p->feature_off.resize((size_t)(n * nf + 1)); p->features.resize((size_t)hdr[5]);
    get(p->ids.data(), p->ids.size() * 4);
    get(p->offsets.data(), p->offsets.size() * 8);
    get(p->seqs.data(), p->seqs.size());
    get(p->entry_off.data(), p->entry_off.size() * 8);
    get(p->entry_ids.data(), p->entry_ids.size());
    get(&names[0], names.size());
    get(p->feature_off.data(), p->feature_off.size() * 8);
    get(p->features.data(), p->features.size());
    if (v2) {   // whole sequences: every length is checked against what is left of the file before it is allocated
        uint64_t n_full = 0, left = size - want - 8;
        get(&n_full, 8);
        for (uint64_t i = 0; ok && i < n_full; i++) {
            uint64_t rec_len[2] = { 0, 0 };
            if (left < 16) { ok = false; break; }
            get(rec_len, 16);
            left -= 16;
            if (!ok || rec_len[0] >= n || rec_len[1] > left) { ok = false; break; }
            std::string fs((size_t)rec_len[1], '\0');
            get(&fs[0], fs.size());
            left -= rec_len[1];
            p->full_seq[(uint32_t)rec_len[0]] = std::move(fs);
        }
        if (left != 0) ok = false;
    }
    fclose(f);
    // offsets must be monotone and end at the array sizes
    auto mono = [](const std::vector<uint64_t> &o, uint64_t end) {
        if (o.empty() || o[0] != 0 || o.back() != end) return false;
        for (size_t i = 1; i < o.size(); i++) if (o[i] < o[i - 1]) return false;
        return true;
    };
    ok = ok && mono(p->offsets, p->seqs.size()) && mono(p->entry_off, p->entry_ids.size()) && mono(p->feature_off, p->features.size());
    if (!ok) { delete p; return kaamer_fail(KAAMER_E_FORMAT, "protein table is truncated or inconsistent"); }
    for (size_t s = 0; s < names.size();) { const size_t e = names.find('\0', s); if (e == std::string::npos) break; p->feature_names.emplace_back(names.substr(s, e - s)); s = e + 1; }
    if (p->feature_names.size() != nf) { delete p; return kaamer_fail(KAAMER_E_FORMAT, "protein table: feature names"); }
    p->n_kmers = hdr[7];
    p->n_aa = p->seqs.size();
    for (uint32_t i = 0; i < (uint32_t)n; i++) p->by_id[p->ids[i]] = i;
    *out = p;
    return KAAMER_OK;
}

}  // extern "C"
