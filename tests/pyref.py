"""Second, independent CPU restatement of the kaamer hot path, in pure Python.

Test infrastructure only.  It is written the way the Go code is written —
dict lookups with zero-value defaults, string concatenation, explicit loops —
so that it shares no code or data-structure choice with oracle/kaamer_oracle.c.
Small inputs only.  Citations are into /root/reference.
"""
from collections import defaultdict

KMER_SIZE = 7
MIN_LEN_CDS = 21

# --- pkg/kvstore/k_store.go:39-64 NewAATable ---------------------------------
_AA = ['A', 'C', 'D', 'E', 'F', 'G', 'H', 'I', 'K', 'L', 'M', 'N', 'P', 'Q', 'R', 'S', 'T', 'U', 'V', 'W', 'Y']


def new_aa_table():
    aa_table = {}
    i = 22
    for j, a in enumerate(_AA):
        aa_table[(a, '.')] = j
        for b in _AA:
            aa_table[(a, b)] = i
            i += 1
    return aa_table


_AA_TABLE = new_aa_table()


def encode_kmer(kmer):
    """k_store.go:91-117.  `kmer` is bytes or str (latin-1)."""
    if isinstance(kmer, bytes):
        kmer = kmer.decode('latin-1')
    kmer_int = 0
    i = 0
    shift_index = 1
    while (i + 2) < len(kmer):
        kmer_int |= (_AA_TABLE.get((kmer[i], kmer[i + 1]), 0) << (32 - ((shift_index * 9) & 0xFF))) & 0xFFFFFFFF
        shift_index += 1
        i += 2
    kmer_int |= _AA_TABLE.get((kmer[-1], '.'), 0)
    return kmer_int


# --- pkg/search/gcode.go:36-101 (transcribed through NCBI table 11 strings; the
# test suite checks every entry against tests/golden/gcode_bacteria.json, which
# was produced by parsing gcode.go as text) ------------------------------------
def _table11():
    base = "tcag"
    aas = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"
    starts = "---M---------------M------------MMMM---------------M------------"
    t = {}
    n = 0
    for a in base:
        for b in base:
            for c in base:
                t[a + b + c] = (aas[n], starts[n] == 'M', aas[n] == '*')
                n += 1
    return t


GCODE_BACTERIA = _table11()


def reverse_complement(dna):  # dna.go:55-63
    comp = {'a': 't', 't': 'a', 'g': 'c', 'c': 'g'}
    return ''.join(comp.get(ch, ch) for ch in reversed(dna.lower()))


def get_frame(frame_number, dna):  # dna.go:183-196
    if frame_number < 0:
        dna = reverse_complement(dna)
        frame_number = -frame_number
    start_pos = frame_number - 1
    len_frame = len(dna) - start_pos
    if len_frame < 0:
        return ""  # the reference panics here; see oracle header
    # Go's % truncates toward zero; len_frame >= 0 here so Python's agrees
    end_pos = len(dna) - (len_frame % 3)
    return dna[start_pos:end_pos]


def get_orfs(dna):  # dna.go:65-181
    frame_start_position = {0: 0, 1: 1, 2: 2, 3: 0, 4: 1, 5: 2}
    orfs = []
    dna = dna.lower()
    frames = [get_frame(f, dna) for f in (1, 2, 3, -1, -2, -3)]
    for frame_pos, frame_seq in enumerate(frames):
        start_pos = frame_start_position[frame_pos]
        plus = frame_pos <= 2
        abs_pos = frame_pos
        if not plus:
            abs_pos = len(dna) - start_pos - 1
        current_pos = 0
        orf = dict(seq="", start=abs_pos + 1, end=0, plus=plus, starts=[])
        inside = True
        cds = ""
        current_aa_pos = 0
        i = 0
        while i < len(frame_seq) - (len(frame_seq) % 3):
            current_pos = i
            aa, is_start, is_stop = GCODE_BACTERIA.get(frame_seq[i:i + 3], ("", False, False))
            if is_start:
                if not inside:
                    inside = True
                    current_aa_pos = 0
                    orf['start'] = frame_pos + i + 1
                    if not plus:
                        orf['start'] = len(dna) - (frame_pos + i) + 3
                    orf['starts'].append(current_aa_pos)
                else:
                    orf['starts'].append(current_aa_pos)
            if inside:
                cds += aa
            if is_stop:
                if inside and len(cds) >= MIN_LEN_CDS:
                    end_pos = i + 3 + frame_pos
                    if not plus:
                        end_pos = orf['start'] - (len(cds) * 3) + 1
                    orf['end'] = end_pos
                    orf['seq'] = cds
                    orfs.append(orf)
                orf = dict(seq="", start=0, end=0, plus=plus, starts=[])
                cds = ""
                inside = False
            current_aa_pos += 1
            i += 3
        if inside and len(cds) >= MIN_LEN_CDS:
            end_pos = current_pos + 3 + frame_pos
            if not plus:
                end_pos = orf['start'] - (len(cds) * 3) + 1
            orf['end'] = end_pos
            orf['seq'] = cds
            orfs.append(orf)
    orfs.sort(key=lambda o: o['end'] if o['plus'] else o['start'])  # stable
    return orfs


def size_in_kmer(seq):  # search.go:290-293
    s = len(seq) - KMER_SIZE + 1
    if len(seq) > 0 and seq[-1:] in ("*", b"*"):
        s -= 1
    return s


def build_index(proteins, ids=None):
    """makedb emit loops (inputTSV.go:236-239) + indexdb de-dup -> dict key -> set(ids)."""
    index = defaultdict(set)
    for n, seq in enumerate(proteins):
        if isinstance(seq, bytes):
            seq = seq.decode('latin-1')
        if len(seq) < KMER_SIZE:
            continue
        pid = n if ids is None else int(ids[n])
        for i in range(len(seq) - KMER_SIZE + 1):
            index[encode_kmer(seq[i:i + KMER_SIZE])].add(pid)
    return index


def kmer_search(index, seq, size=None):
    """search.go:414-452 -> (counter dict pid->Kmatch, positions dict pid->list[bool])"""
    if isinstance(seq, bytes):
        seq = seq.decode('latin-1')
    if size is None:
        size = size_in_kmer(seq)
    counter = defaultdict(int)
    positions = {}
    for k in range(size):
        ids = index.get(encode_kmer(seq[k:k + KMER_SIZE]))
        if not ids:
            continue
        for pid in ids:
            counter[pid] += 1
            if pid not in positions:
                positions[pid] = [False] * size
            positions[pid][k] = True
    return dict(counter), positions


def filter_results(hits, size, min_k_ratio=0.05, min_k_match=10, max_results=10):
    """search.go:189-220; hits = list of (pid, kmatch) sorted by kmatch desc."""
    last_good = len(hits) - 1
    for i, (_, km) in enumerate(hits):
        if (float(km) / float(size)) < min_k_ratio or km < min_k_match:
            if last_good == len(hits) - 1:
                last_good = i - 1
    if last_good >= max_results:
        last_good = max_results - 1
    if last_good < 0:
        return []
    return hits[0:last_good + 1]


def set_best_start_codon(hits, positions, query):
    """dna.go:198-272; query = dict(seq,start,plus,starts,size). Mutates a copy."""
    q = dict(query)
    best_hits = []
    best_score = 0
    for pid, km in hits:
        if km >= best_score:
            best_score = km
            best_hits.append(pid)
    if len(q['starts']) < 1:
        return q, 0
    best_start = q['starts'][0]
    first_start = q['starts'][0]
    first_best_hit_pos = 999999999
    exit_ = False
    for pid in best_hits:
        for i, is_match in enumerate(positions[pid]):
            if is_match:
                if i < first_best_hit_pos:
                    first_best_hit_pos = i
                exit_ = True
            if exit_:
                break
    for s in q['starts']:
        if s <= first_best_hit_pos:
            best_start = s
        else:
            break
    trimmed = 0
    if best_start != first_start:
        if q['plus']:
            q['start'] = q['start'] + 3 * best_start
        else:
            q['start'] = q['start'] - 3 * best_start
        q['seq'] = q['seq'][best_start:]
        q['size'] = size_in_kmer(q['seq'])
        trimmed = best_start
    q['starts'] = []
    return q, trimmed


# --- readers: search.go:222-412 ------------------------------------------------
def get_queries_fasta(text, is_protein=True):
    """search.go:222-322 on already-decompressed text. -> list of dict."""
    out = []
    seq = ""
    name = ""
    have = False
    for l in text.split("\n"):
        l = l.rstrip("\r") if False else l  # bufio.ScanLines strips a trailing \r
        if l.endswith("\r"):
            l = l[:-1]
        if len(l) < 1:
            continue
        if l[0] == '>':
            if seq != "":
                size = len(seq) - KMER_SIZE + 1
                if seq[-1:] == "*":
                    size -= 1
                # the first Query carries Location{PlusStrand: true} (search.go:224-229); the reader then rebuilds
                # query = Query{Sequence: "", ...} (search.go:297): every later record has Go's zero Location
                out.append(dict(seq=seq.upper(), name=name, size=size, end=len(seq), plus=len(out) == 0))
                seq = ""
            name = l[1:]
            have = True
        else:
            seq += l.strip()
    if seq != "":
        size = len(seq) - KMER_SIZE + 1
        if seq[-1:] == "*":
            size -= 1
        out.append(dict(seq=seq, name=name, size=size, end=len(seq), plus=len(out) == 0))  # NOT upper-cased (search.go:313-320)
    return out


def get_queries_fastq(text):
    """search.go:324-412."""
    import re
    is_seq = re.compile(r'^[ATGCNatgcn]+$')
    out = []
    seq = ""
    name = ""
    for l in text.split("\n"):
        if l.endswith("\r"):
            l = l[:-1]
        if len(l) < 1:
            continue
        if l[0] == '@':
            if seq != "":
                out.append(dict(seq=seq, name=name, size=len(seq) - KMER_SIZE + 1, end=len(seq)))
                seq = ""
                name = ""
            name = l[1:]
        elif is_seq.match(l):
            seq = l
    if seq != "":
        out.append(dict(seq=seq, name=name, size=len(seq) - KMER_SIZE + 1, end=len(seq)))
    return out
