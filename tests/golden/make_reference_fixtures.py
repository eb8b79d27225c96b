"""Generates the fixtures that pin the oracle to the reference's own text.

Run in the build container (needs /root/reference, read as TEXT only — the
reference is Go and cannot be executed here):

    python tests/golden/make_reference_fixtures.py

Outputs (data only: inputs and expected outputs, no reference source text):
  gcode_bacteria.json  codon -> [AA, Start, Stop], parsed from the map literal
                       gcodeBacteria in pkg/search/gcode.go:36-101
  docs_example.json    the worked protein-search example of docs/client.md:131-180
                       (query sequence, SizeInKmer, hit Kmatch, PositionHits length)
  matrix_scores.json   "<matrix>_<open>_<extend>" -> [lambda, K] and the letter -> index map AAPosInMatrix, parsed from
                       the literals of pkg/align/matrixScores.go:22-117
"""
import json
import os
import re

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def gcode():
    src = open(os.path.join(REF, "pkg/search/gcode.go")).read()
    m = re.search(r"var gcodeBacteria = map\[string\]AminoAcid\{(.*?)\n\}", src, re.S)
    body = m.group(1)
    table = {}
    for codon, aa, start, stop in re.findall(
            r'"([a-z]{3})":\s*AminoAcid\{AA:\s*"(.)",\s*Start:\s*(true|false),\s*Stop:\s*(true|false)\}', body):
        table[codon] = [aa, start == "true", stop == "true"]
    assert len(table) == 64, len(table)
    return table


def docs_example():
    md = open(os.path.join(REF, "docs/client.md")).read()
    m = re.search(r"```javascript\n(.*?)```", md, re.S)
    js = m.group(1)
    seq = re.search(r'"Query": \{\s*"Sequence": "([A-Z]+)"', js).group(1)
    size = int(re.search(r'"SizeInKmer": (\d+)', js).group(1))
    key = int(re.search(r'"Key": (\d+)', js).group(1))
    kmatch = int(re.search(r'"Kmatch": (\d+)', js).group(1))
    pos = re.search(r'"%d": \[([truefals,]+)\]' % key, js).group(1).split(",")
    db_seq = re.search(r'"Sequence": "([A-Z]+)",\s*"Length": (\d+)', js)
    return dict(query=seq, size_in_kmer=size, hit_key=key, kmatch=kmatch,
                n_positions=len(pos), all_positions_true=all(p == "true" for p in pos),
                db_sequence=db_seq.group(1), db_length=int(db_seq.group(2)),
                start_position=1, end_position=int(re.search(r'"EndPosition": (\d+)', js).group(1)))


def matrix_scores():
    src = open(os.path.join(REF, "pkg/align/matrixScores.go")).read()
    rows = re.findall(r'"([a-z0-9]+_\d+_\d+)":\s*MatrixScores\{SubMatrix: matrix\.([A-Z0-9]+), GapOpen: (\d+), GapExtend: (\d+), Lambda: ([0-9.]+), K: ([0-9.]+)\}', src)
    table = {}
    for key, mat, go, ge, lam, k in rows:
        assert key == "%s_%s_%s" % (mat.lower(), go, ge), key
        table[key] = [float(lam), float(k)]
    assert len(table) == len(rows) > 80
    pos = re.search(r"AAPosInMatrix = map\[rune\]int\{(.*?)\}", src, re.S).group(1)
    aa = {m[0]: int(m[1]) for m in re.findall(r"'(.)': (\d+)", pos)}
    assert len(aa) == 26
    return {"lambda_k": table, "aa_pos_in_matrix": aa}


if __name__ == "__main__":
    json.dump(matrix_scores(), open(os.path.join(HERE, "matrix_scores.json"), "w"), indent=0, sort_keys=True)
    json.dump(gcode(), open(os.path.join(HERE, "gcode_bacteria.json"), "w"), indent=0, sort_keys=True)
    json.dump(docs_example(), open(os.path.join(HERE, "docs_example.json"), "w"), indent=1, sort_keys=True)
    print("ok")
