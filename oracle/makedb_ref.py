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
