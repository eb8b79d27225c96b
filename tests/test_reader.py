"""kaamer_reader_*: the query readers over a FILE in chunks (search.go:240-412), against the whole-buffer readers (which
tests/test_abi_host.py checks against the pure-Python restatement tests/pyref.py) -- every chunking must give the same
records, including the rules that span records (FASTA: the LAST record of the file is not upper-cased; PlusStrand is
true on the file's first record only), plain and gzipped (multi-member, broken), and the strict scanner mode."""
import gzip
import os

import numpy as np
import pytest

from kaamer_amd import abi, api


def _fasta(rng, n, lower=True):
    aa = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWYacdefghiklmnpqrstvwy*" if lower else b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)
    out = []
    for i in range(n):
        L = int(rng.integers(1, 400))
        s = bytes(aa[rng.integers(0, len(aa), L)]).decode()
        w = int(rng.integers(20, 90))
        lines = [s[k:k + w] for k in range(0, L, w)]
        if rng.random() < 0.2:
            lines.insert(int(rng.integers(0, len(lines) + 1)), "")        # blank lines are skipped
        if rng.random() < 0.2:
            lines = [("  " + x + "\t") for x in lines]                      # TrimSpace
        out.append(">rec%d some text\n%s\n" % (i, ("\r\n" if rng.random() < 0.1 else "\n").join(lines)))
    return "".join(out)


def _fastq(rng, n):
    nt = np.frombuffer(b"ACGTNacgtn", dtype=np.uint8)
    out = []
    for i in range(n):
        L = int(rng.integers(1, 300))
        s = bytes(nt[rng.integers(0, 10, L)]).decode()
        q = "".join(chr(int(c)) for c in rng.integers(33, 74, L))
        if rng.random() < 0.1:
            q = "@" + q[1:]            # a quality line that starts with '@' opens a record in the reference
        if rng.random() < 0.1:
            s = s[:L // 2] + "X" + s[L // 2 + 1:]   # not ^[ATGCNatgcn]+$: the record keeps an earlier sequence or none
        out.append("@read%d\n%s\n+\n%s\n" % (i, s, q))
    return "".join(out)


@pytest.mark.parametrize("fmt", ["fasta", "fastq"])
def test_chunked_reader_equals_whole_buffer(klib, tmp_path, fmt):
    rng = np.random.default_rng(5 if fmt == "fasta" else 6)
    text = (_fasta(rng, 700) if fmt == "fasta" else _fastq(rng, 900)).encode()
    if fmt == "fasta":
        text = text.rstrip(b"\n")          # an unterminated last line
    want = api.parse_reads(text, fmt)
    assert len(want) > 500
    z1 = gzip.compress(text)
    third = len(text) // 3
    zmulti = gzip.compress(text[:third]) + gzip.compress(text[third:2 * third]) + gzip.compress(b"") + gzip.compress(text[2 * third:])
    for name, blob in (("plain", text), ("gz", z1), ("multi.gz", zmulti)):
        p = tmp_path / name
        p.write_bytes(blob)
        for max_seqs, max_bytes in ((1 << 20, 1 << 30), (1, 1 << 30), (7, 1 << 30), (1000, 300), (64, 5000)):
            rd = api.Reader(p, fmt)
            got = rd.records_list(max_seqs, max_bytes)
            assert rd.done and rd.records == len(want)
            assert got == want, (name, max_seqs, max_bytes)
            rd.close()
        # by descriptor (the caller's fd stays open)
        fd = os.open(p, os.O_RDONLY)
        assert api.Reader(fmt=fmt, fd=fd).records_list(100, 1 << 20) == want
        os.close(fd)
    # a gzip stream that breaks off: what inflated before the break, as the whole-buffer reader has it
    for keep in (len(z1) - 5, len(z1) // 2, 30):
        p = tmp_path / "cut.gz"
        p.write_bytes(z1[:keep])
        assert api.Reader(p, fmt).records_list(50, 1 << 20) == api.parse_reads(z1[:keep], fmt), keep
    # the signature followed by no valid header, an empty file, a missing file
    p = tmp_path / "bad.gz"
    p.write_bytes(b"\x1f\x8b\x08" + b"\xff" * 40)
    assert api.Reader(p, fmt).records_list() == []
    p = tmp_path / "empty"
    p.write_bytes(b"")
    assert api.Reader(p, fmt).records_list() == []
    with pytest.raises(abi.KaamerError) as e:
        api.Reader(tmp_path / "missing", fmt)
    assert e.value.code == abi.E_IO


def test_chunks_stop_at_the_bounds(klib, tmp_path):
    rng = np.random.default_rng(9)
    p = tmp_path / "r.fq"
    p.write_bytes(_fastq(rng, 500).encode())
    rd = api.Reader(p, "fastq")
    n = 0
    while True:
        c = rd.next(max_seqs=37, max_bytes=1 << 30)
        if c is None:
            break
        assert 1 <= len(c[2]) <= 37
        n += len(c[2])
    assert n == rd.records > 400
    rd = api.Reader(p, "fastq")
    while True:
        c = rd.next(max_seqs=1 << 20, max_bytes=2000)
        if c is None:
            break
        assert len(c[0]) < 2000 + 300      # stops once the bound is reached: at most one record past it


def test_strict_scanner_mode(klib, tmp_path):
    """bufio.Scanner with a 1 MiB buffer (search.go:273-274): a line of 1 048 576 bytes or more ends the reference's scan;
    the record being read is then emitted as the LAST one (not upper-cased).  Default mode reads lines of any length."""
    MAX = 1024 * 1024
    head = ">a\nacdef\nGHIKL\n>b\nmnpqrstvwy\n"
    tail = "\n>c\nACDEFGHIKL\n"
    for n, cut in ((MAX - 1, False), (MAX, True), (MAX + 5, True)):
        text = (head + "k" * n + tail).encode()
        p = tmp_path / "long.fa"
        p.write_bytes(text)
        loose = api.Reader(p, "fasta", strict=False).records_list()
        assert [r["name"] for r in loose] == ["a", "b", "c"] and loose[1]["seq"] == "MNPQRSTVWY" + "K" * n
        strict = api.Reader(p, "fasta", strict=True).records_list()
        if not cut:
            assert strict == loose
        else:   # the scan ends inside record b: b is what it had before the long line, as the last record (lower case kept)
            assert [(r["name"], r["seq"]) for r in strict] == [("a", "ACDEFGHIKL"), ("b", "mnpqrstvwy")]
        # the same through gzip, and with the long line as the unterminated end of the file
        pz = tmp_path / "long.fa.gz"
        pz.write_bytes(gzip.compress(text, compresslevel=1))
        assert api.Reader(pz, "fasta", strict=True).records_list() == strict
        p.write_bytes((head + "k" * n).encode())
        s2 = api.Reader(p, "fasta", strict=True).records_list()
        assert len(s2) == 2 and s2[1]["seq"] == ("mnpqrstvwy" if cut else "mnpqrstvwy" + "k" * n)
    # http.DetectContentType: anything but "text/plain; charset=utf-8" yields no query (search.go:266-270)
    for blob, none in ((b"\xff\xfe>\x00a\x00\n\x00", True), (b">a\x01\nACDEFGHIKL\n", True), (b"\xef\xbb\xbf>a\nACDEFGHIKL\n", False),
                       (b">a\nACDEFGHIKL\n", False)):
        p = tmp_path / "sniff.fa"
        p.write_bytes(blob)
        assert (api.Reader(p, "fasta", strict=True).records_list() == []) == none
        assert api.Reader(p, "fasta", strict=False).records_list() != []
