"""Gzipped input (search.go:255-263, 361-366; inputEMBL.go:76-84, inputGBK.go:75-83): the readers take the bytes of a file,
and bytes that start with the gzip signature (what http.DetectContentType calls application/x-gzip) are inflated first.
Go's gzip.Reader is multistream and the reference's scanner stops silently at a read error, handing over what was read:
concatenated members read as one text, a stream that breaks off reads as its inflated prefix.  CPU only."""
import gzip
import zlib

import numpy as np
import pytest

from kaamer_amd import abi, api

FASTA = ">q1 first\nMKTAYIAKQRQISFVKSHFSRQ\nLEERLGLIEVQ\n>q2\nACDEFGHIKLMNPQRSTVWY\n>q3 last keeps its case\nmktayiakqr\n"
FASTQ = "@r1 desc\nATGGCTAAGCAACGTCAGATT\n+\nIIIIIIIIIIIIIIIIIIIII\n@r2\nTTGACCGGTAACNNACGT\n+r2\nIIIIIIIIIIIIIIIIII\n"


def _same_reads(a, b):
    assert [(x["name"], x["seq"], x["size"], x["plus"]) for x in a] == [(x["name"], x["seq"], x["size"], x["plus"]) for x in b]


@pytest.mark.parametrize("fmt,text", [("fasta", FASTA), ("fastq", FASTQ)])
def test_gzipped_queries_read_as_their_text(klib, fmt, text):
    plain = api.parse_reads(text, fmt)
    assert len(plain) >= 2
    raw = text.encode()
    _same_reads(api.parse_reads(gzip.compress(raw), fmt), plain)
    _same_reads(api.parse_reads(gzip.compress(raw, compresslevel=1, mtime=0), fmt), plain)
    # two members: gzip.Reader goes on with the next one (multistream)
    cut = raw.index(b"\n", len(raw) // 2) + 1
    _same_reads(api.parse_reads(gzip.compress(raw[:cut]) + gzip.compress(raw[cut:]), fmt), plain)
    # an empty member in front
    _same_reads(api.parse_reads(gzip.compress(b"") + gzip.compress(raw), fmt), plain)
    # bytes behind the last member that are no gzip header end the text there
    _same_reads(api.parse_reads(gzip.compress(raw) + b"\x00\x01garbage", fmt), plain)


def test_a_stream_that_breaks_off_reads_as_its_prefix(klib):
    rng = np.random.default_rng(3)
    aa = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)
    text = "".join(">s%d\n%s\n" % (i, bytes(aa[rng.integers(0, 20, 60 + i)]).decode()) for i in range(400)).encode()
    z = gzip.compress(text, compresslevel=6)
    for keep in (len(z) - 1, len(z) - 9, len(z) * 2 // 3, len(z) // 3, 40):
        d = zlib.decompressobj(16 + zlib.MAX_WBITS)
        prefix = d.decompress(z[:keep])          # what a reader gets before the stream ends early
        _same_reads(api.parse_reads(z[:keep], "fasta"), api.parse_reads(prefix, "fasta") if prefix else [])
    # damaged in the middle: whatever inflates before the decoder notices (the same bytes for every inflate implementation
    # fed the same stream), then the text ends
    bad = bytearray(z)
    bad[len(z) // 2] ^= 0xFF
    bad[len(z) // 2 + 1] ^= 0xFF
    # (the flipped bytes leave the deflate structure intact here: the decoder delivers everything and then finds the
    # checksum wrong -- Go's gzip.Reader returns those last bytes together with the error, the scanner keeps them;
    # raw inflate of the body = the same bytes without the check.  gzip.compress writes a 10-byte header.)
    prefix = zlib.decompressobj(-zlib.MAX_WBITS).decompress(bytes(bad[10:]))
    assert prefix != text
    assert 0 < len(prefix) < len(text) + 4096
    _same_reads(api.parse_reads(bytes(bad), "fasta"), api.parse_reads(prefix, "fasta"))


def test_not_gzip_after_all_reads_as_no_queries(klib):
    """the signature, then no valid header: gzip.NewReader fails, GetQueriesFasta prints the error and returns without a
    query (search.go:259-263) -- an empty read set, not an error; the makedb readers log.Fatal there: KAAMER_E_FORMAT"""
    bad = b"\x1f\x8b\x08" + b"\xff" * 40
    assert api.parse_reads(bad, "fasta") == [] and api.parse_reads(bad, "fastq") == []
    with pytest.raises(abi.KaamerError) as e:
        api.Proteins.from_embl(bad)
    assert e.value.code == abi.E_FORMAT


def test_gzipped_embl_and_gbk(klib):
    from test_makedb import EMBL, GBK
    for text, make in ((EMBL, api.Proteins.from_embl), (GBK, api.Proteins.from_gbk)):
        a, b = make(text), make(gzip.compress(text))
        assert len(a) == len(b) > 0
        assert a.ids.tolist() == b.ids.tolist()
        assert bytes(a.packed[0]) == bytes(b.packed[0]) and a.packed[1].tolist() == b.packed[1].tolist()
        assert a.fetch_hits(a.ids.tolist()) == b.fetch_hits(b.ids.tolist())
    # the FASTA and TSV makedb readers have no gzip branch in the reference: the bytes are taken as text
    p = api.Proteins.from_fasta(gzip.compress(b">a\nMKTAYIAKQRQISFVK\n>b\nMKTAYIAKQRQISFVR\n"))
    assert b"MKTAYIAKQRQISFVK" not in bytes(p.packed[0])          # (not inflated: whatever lines the bytes happen to form)


def test_random_bytes_behind_the_signature(klib):
    """fuzz: the gzip signature followed by random bytes, and valid streams with random damage -- an error or some text,
    never a crash (runs under ASan/UBSan in tools/asan)"""
    rng = np.random.default_rng(11)
    z = gzip.compress(FASTA.encode() * 50)
    for trial in range(300):
        if trial % 2:
            data = b"\x1f\x8b\x08" + bytes(rng.integers(0, 256, int(rng.integers(0, 200)), dtype=np.uint8))
        else:
            b = bytearray(z)
            for _ in range(int(rng.integers(1, 6))):
                b[int(rng.integers(3, len(b)))] = int(rng.integers(0, 256))
            data = bytes(b[:int(rng.integers(3, len(b) + 1))])
        for fmt in ("fasta", "fastq"):
            try:
                api.parse_reads(data, fmt)
            except abi.KaamerError as e:
                assert e.code == abi.E_FORMAT
        try:
            api.Proteins.from_embl(data)
        except abi.KaamerError as e:
            assert e.code == abi.E_FORMAT
