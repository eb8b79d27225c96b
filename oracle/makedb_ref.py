"""TEST INFRASTRUCTURE — not part of the product.  Literal restatement of the reference's makedb input
readers (pure-Python loops, small cases only), kept line for line with the Go so the id quirks survive:

  run_fasta   pkg/makedb/inputFASTA.go:60-126 (runFASTA's scan loop) + :191-250 (processProteinInputFASTA)
  run_tsv     pkg/makedb/inputTSV.go:60-142  (runTSV's scan loop)   + :184-241 (processProteinInputTSV)

Each returns [(protein_id, entry_id, sequence, {feature: value})] in the order the reference queues them;
k-mers of a protein are then stored under protein_id (inputFASTA.go:233-247, inputTSV.go:225-239) and the
Protein entry {EntryId, Sequence, Length, Features} under the same id (protein_store / protein.proto).
Parity unpinned by reference fixtures: the reference holds no tests or data files for makedb;
this follows the source text only."""

KMER_SIZE = 7


def _scan_lines(text):
    """bufio.ScanLines: split at '\\n', drop one trailing '\\r', no final empty token"""
    if not text:
        return []
    lines = text.split(b"\n")
    if lines[-1] == b"":
        lines.pop()
    return [l[:-1] if l.endswith(b"\r") else l for l in lines]


def run_fasta(text):
    out = []
    protein_nb = 0                                         # inputFASTA.go:65
    entry = []                                             # currentProtein.Entry
    jobs = []
    for line in _scan_lines(text):
        if len(line) < 1:
            continue                                       # (the reference indexes line[0:1] and panics: documented divergence)
        if line[0:1] == b">":                              # :98
            protein_nb += 1                                # :99
            if entry:                                      # :100
                jobs.append((protein_nb, entry))           # :101-106: the pending entry is sent with the NEW number
                entry = []
        entry.append(line)                                 # :114-117
    if entry:
        jobs.append((protein_nb, entry))                   # :120-124
    for pid, lines in jobs:                                # processProteinInputFASTA
        entry_id, name, seq = b"", b"", b""
        for l in lines:
            if len(l) < 1:
                continue
            if l[0:1] == b">":
                head = l[1:].split(b" ", 1)                # :205-207 SplitN(l[1:], " ", 2)
                entry_id = head[0]
                name = head[1] if len(head) > 1 else b""
            else:
                seq += l.upper()                           # :210 strings.ToUpper
        if b", partial" in name:                           # :215
            continue
        if len(seq) < KMER_SIZE:                           # :222
            continue
        out.append((pid, entry_id, seq, {b"ProteinName": name}))
    return out


def run_tsv(text):
    out = []
    lines = _scan_lines(text)
    if not lines:
        raise ValueError("TSV file doesn't contain 'EntryID' header")
    header = lines[0].split(b"\t")                         # inputTSV.go:96
    low = [h.lower() for h in header]
    if b"entryid" not in low:                              # :99-112
        raise ValueError("TSV file doesn't contain 'EntryID' header")
    if b"sequence" not in low:
        raise ValueError("TSV file doesn't contain 'Sequence' header")
    protein_nb = 0                                         # :62
    for line in lines[1:]:
        cols = line.split(b"\t")                           # :118
        entry_id, seq, feat = b"", b"", {}
        for i, c in enumerate(cols):                       # :125-134
            if i >= len(header):
                break                                      # (the reference indexes features[i] and panics: documented divergence)
            if low[i] == b"entryid":
                entry_id = c
            elif low[i] == b"sequence":
                seq = c
            else:
                feat[header[i]] = c
        if len(seq) < KMER_SIZE or seq == b"" or entry_id == b"":   # :137-139
            continue
        for h, l in zip(header, low):                      # columns missing from a short row read as ""
            if l not in (b"entryid", b"sequence"):
                feat.setdefault(h, b"")
        out.append((protein_nb, entry_id, seq, feat))      # :140
        protein_nb += 1                                    # :141
    return out


# ---- EMBL / GBK (round 3) ---------------------------------------------------------------------------------
# run_embl   pkg/makedb/inputEMBL.go:46-118 (runEMBL's scan loop) + :189-314 (processProteinInputEMBL)
# run_gbk    pkg/makedb/inputGBK.go:45-117  (runGBK's scan loop)  + :186-301 (processProteinInputGBK)
# Where the Go would panic (a slice past the end of a short line, Fields(...)[0] of a blank field list, a declared
# length beyond the sequence) the restatement raises RefPanic for that entry; the product skips what the panic
# would have taken down (documented divergence) and the tests keep such inputs apart.
import re as _re

EMBL_DEF_FTS = [b"ProteinName", b"GeneName", b"EC", b"GO", b"KEGG_ID", b"BioCyc_ID", b"HAMAP", b"Organism", b"TaxId", b"FullTaxonomy"]
GBK_DEF_FTS = [b"ProteinName", b"Organism", b"FullTaxonomy"]


class RefPanic(Exception):
    pass


def _sl(l, a):
    """Go l[a:]: panics when a > len(l)"""
    if a > len(l):
        raise RefPanic("slice bounds out of range")
    return l[a:]


def _fields(s):
    return s.split()       # strings.Fields: runs of white space (ASCII inputs)


def _first(fs):
    if not fs:
        raise RefPanic("index out of range")
    return fs[0]


def _entries(text):
    """the scan loops (inputEMBL.go:95-113, inputGBK.go:94-112) with offset 0 and the default length:
    -> [(protein id, entry text)]"""
    out = []
    protein_nb = 0
    entry = b""
    for line in _scan_lines(text):
        if line == b"//":
            protein_nb += 1
            if entry != b"":
                out.append((protein_nb, entry))
                entry = b""
        else:
            entry += line + b"\n"
    return out


_EMBL_REG = _re.compile(rb" \{.*\};")


def process_embl(entry):
    """-> None (skipped) | (entry_id, sequence, declared length, features)"""
    entry_id, seq, length = b"", b"", 0
    feat = {}
    for l in entry.split(b"\n"):
        if len(l) < 2:
            continue
        tag = l[0:2]
        if tag == b"ID":
            entry_id = _first(_fields(_sl(l, 5)))
        elif tag == b"GN":
            if feat.get(b"GeneName", b"") == b"" and b"Name=" in l:
                feat[b"GeneName"] = _sl(_first(_fields(_sl(l, 5))), 5).rstrip(b";")
        elif tag == b"DE":
            if b"RecName" in _sl(l, 5):
                feat[b"ProteinName"] = _EMBL_REG.sub(b"", _sl(l, 19)).rstrip(b";")
            elif b"SubName" in _sl(l, 5):
                v = _EMBL_REG.sub(b"", _sl(l, 19)).rstrip(b";")
                if feat.get(b"ProteinName", b"") != b"":
                    feat[b"ProteinName"] += b";;" + v
                else:
                    feat[b"ProteinName"] = v
            elif b"EC=" in _sl(l, 5):
                feat[b"EC"] = _EMBL_REG.sub(b"", _sl(l, 17)).rstrip(b";")
            elif b"Flags: Fragment;" in _sl(l, 5):
                return None
        elif tag == b"OX":
            feat[b"TaxId"] = _sl(_first(_fields(_sl(l, 5))), 12).rstrip(b";")
        elif tag == b"OS":
            if b"Organism" in feat:
                feat[b"Organism"] += b" " + _sl(l, 5).rstrip(b".")
            else:
                feat[b"Organism"] = _sl(l, 5).rstrip(b".")
        elif tag == b"OC":
            if feat.get(b"FullTaxonomy", b"") != b"":
                feat[b"FullTaxonomy"] += b" "
            feat[b"FullTaxonomy"] = feat.get(b"FullTaxonomy", b"") + _sl(l, 5)
        elif tag == b"DR":
            fields = _fields(_sl(l, 5))
            key = {b"KEGG;": b"KEGG_ID", b"GO;": b"GO", b"BioCyc;": b"BioCyc_ID", b"HAMAP;": b"HAMAP"}.get(_first(fields))
            if key is not None:
                if len(fields) < 2:
                    raise RefPanic("index out of range")
                if key in feat:
                    feat[key] += b";" + fields[1].rstrip(b";")
                else:
                    feat[key] = fields[1].rstrip(b";")
        elif tag == b"SQ":
            fields = _fields(_sl(l, 5))
            if len(fields) < 2:
                raise RefPanic("index out of range")
            try:
                length = int(fields[1]) if _re.fullmatch(rb"[+-]?[0-9]+", fields[1]) else 0   # strconv.Atoi, error ignored
            except ValueError:
                length = 0
            if not -2**31 <= length < 2**31:
                length = 0 if abs(length) >= 2**63 else ((length + 2**31) % 2**32) - 2**31     # Atoi range error -> clamp; int32() wraps
        elif tag == b"  ":
            seq += _sl(l, 5).replace(b" ", b"")
    if length < KMER_SIZE:
        return None
    if length > len(seq):
        raise RefPanic("sequence shorter than its declared length")
    return entry_id, seq, length, feat


def run_embl(text):
    """-> [(protein id, entry id, indexed residues = Sequence[:Length], stored sequence, features)]"""
    out = []
    for pid, entry in _entries(text):
        r = process_embl(entry)
        if r is not None:
            entry_id, seq, length, feat = r
            out.append((pid, entry_id, seq[:length], seq, feat))
    return out


_GBK_STATE = {b"LOCUS": 0, b"DEFINITION": 1, b"ACCESSION": 0, b"VERSION": 2, b"KEYWORDS": 0, b"SOURCE": 0, b"ORGANISM": 3,
              b"COMMENT": 0, b"FEATURES": 4, b"ORIGIN": 5, b"//": 6, b"REFERENCE": 0, b"DBLINK": 0, b"DBSOURCE": 0}
_GBK_REG = _re.compile(rb" \[.*\]\.")


def process_gbk(entry):
    entry_id, seq = b"", b""
    feat = {}
    inside = 0
    for l in entry.split(b"\n"):
        if len(l) < 2:
            continue
        tok = l.strip(b" ").split(b" ")[0]
        inside = _GBK_STATE.get(tok, inside)
        if inside == 1:
            if feat.get(b"ProteinName", b"") != b"":
                feat[b"ProteinName"] += b" "
            feat[b"ProteinName"] = feat.get(b"ProteinName", b"") + _sl(l, 12)
        elif inside == 2:
            entry_id = _first(_fields(_sl(l, 12)))
        elif inside == 3:
            if feat.get(b"Organism", b"") == b"":
                feat[b"Organism"] = _sl(l, 12)
            else:
                if feat.get(b"FullTaxonomy", b"") != b"":
                    feat[b"FullTaxonomy"] += b" "
                feat[b"FullTaxonomy"] = feat.get(b"FullTaxonomy", b"") + _sl(l, 12)
        elif inside == 5:
            if _sl(l, 10) != b"":
                seq += _sl(l, 10).replace(b" ", b"").upper()
    if b", partial" in feat.get(b"ProteinName", b""):
        return None
    if len(seq) < KMER_SIZE:
        return None
    feat[b"ProteinName"] = _GBK_REG.sub(b"", feat.get(b"ProteinName", b""))
    return entry_id, seq, len(seq), feat


def run_gbk(text):
    out = []
    for pid, entry in _entries(text):
        r = process_gbk(entry)
        if r is not None:
            entry_id, seq, length, feat = r
            out.append((pid, entry_id, seq, seq, feat))
    return out
