"""Byte-flipping fuzz of the two file loaders that face untrusted files (kaamer_image_load, kaamer_proteins_load) and of
the text readers: a damaged file is either refused (KAAMER_E_FORMAT / KAAMER_E_IO) or loads into an object every
accessor of which stays inside its buffers.  Runs in the CPU suite, and under AddressSanitizer + UBSan through
tools/asan (`make -C tools/asan test`), where any out-of-bounds read aborts the run."""
import numpy as np
import pytest

from kaamer_amd import abi, api


def _mutations(data, rng, n):
    for _ in range(n):
        b = bytearray(data)
        kind = int(rng.integers(0, 4))
        if kind == 0:                                   # flip bits, a few positions, biased to the header
            for _ in range(int(rng.integers(1, 6))):
                i = int(rng.integers(0, min(len(b), 4200))) if rng.random() < 0.6 else int(rng.integers(0, len(b)))
                b[i] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:                                 # overwrite 8 bytes with an extreme value
            i = int(rng.integers(0, max(1, len(b) - 8)))
            b[i:i + 8] = [b"\xff" * 8, b"\x00" * 8, (2 ** 63).to_bytes(8, "little"), (len(b) * 3).to_bytes(8, "little")][int(rng.integers(0, 4))]
        elif kind == 2:                                 # truncate
            b = b[:int(rng.integers(0, len(b)))]
        else:                                           # append garbage
            b += bytes(rng.integers(0, 256, int(rng.integers(1, 64)), dtype=np.uint8))
        yield bytes(b)


def test_damaged_image_files(klib, tmp_path):
    import ctypes as C
    from kaamer_amd import workload
    db = workload.make_db(40, seed=3)
    img = api.Image.from_proteins(packed=db)
    good = tmp_path / "good.kgi"
    img.save(good)
    data = good.read_bytes()
    keys = [int(klib.kaamer_encode_kmer(bytes(db[0][i:i + 7]))) for i in range(0, 60, 7)]
    rng = np.random.default_rng(5)
    n_loaded = n_refused = 0
    for i, mut in enumerate(_mutations(data, rng, 300)):
        path = tmp_path / "m.kgi"
        path.write_bytes(mut)
        h = C.c_void_p()
        rc = klib.kaamer_image_load(str(path).encode(), C.byref(h))
        if rc != 0:
            assert rc in (abi.E_FORMAT, abi.E_IO, abi.E_NOMEM), rc
            n_refused += 1
            continue
        n_loaded += 1
        im = api.Image(h.value)
        im.stats()
        for k in keys:                                  # point reads walk buckets and lists: must stay in bounds
            im.get(k)
        im.close()
    assert n_refused > 100 and n_loaded + n_refused == 300


def test_damaged_protein_tables(klib, tmp_path):
    import ctypes as C
    text = b">a first\nMKTAYIAKQRQISTFVKSHFSRQ\n>b second one\nACDEFGHIKLMNPQRSTVWY\n>c\nMKVLAAGTTEFGHIK\n"
    p = api.Proteins.from_fasta(text)
    good = tmp_path / "good.kpt"
    p.save(good)
    data = good.read_bytes()
    rng = np.random.default_rng(6)
    n_refused = 0
    for mut in _mutations(data, rng, 300):
        path = tmp_path / "m.kpt"
        path.write_bytes(mut)
        h = C.c_void_p()
        rc = klib.kaamer_proteins_load(str(path).encode(), C.byref(h))
        if rc != 0:
            assert rc in (abi.E_FORMAT, abi.E_IO, abi.E_NOMEM), rc
            n_refused += 1
            continue
        q = api.Proteins(h.value)
        q.ids, q.packed, q.feature_names, q.stats()
        q.fetch_hits([0, 1, 2, 3, 4, 0xFFFFFFFF])
        q.close()
    assert n_refused > 100


def test_random_bytes_through_the_text_readers(klib):
    """every reader on arbitrary bytes (NULs, high bytes, lone CRs, very long and empty lines): no crash, and the
    accessors of what comes back stay in bounds"""
    import ctypes as C
    rng = np.random.default_rng(7)
    alphabet = np.frombuffer(b">@+\n\r\t /;{}[].=ACGTNacgtnMKVLXYZ\x00\xff,", dtype=np.uint8)
    for trial in range(200):
        n = int(rng.integers(0, 600))
        text = bytes(alphabet[rng.integers(0, len(alphabet), n)])
        if trial % 7 == 0:
            text = b"//\n".join([text] * 3) + b"//\n"
        for make in (api.Proteins.from_fasta, api.Proteins.from_embl, api.Proteins.from_gbk):
            p = make(text)
            p.ids, p.packed, p.stats()
            p.fetch_hits(list(p.ids[:3]))
            p.close()
        try:
            p = api.Proteins.from_tsv(b"EntryID\tSequence\tX\n" + text)
            p.packed
            p.close()
        except abi.KaamerError:
            pass
        for fmt in ("fasta", "fastq"):
            api.parse_reads(text, fmt)
