"""The device builder (kaamer_image_build_proteins_device, builder_device.hip) against the host builder
(builder.cpp): the same image, byte for byte — header, buckets, arena — and the same search results.

The host builder walks the keys in ascending order (postings sets shared first-seen, keys into the first free
slot along the probe sequence: the offline side of pkg/indexdb/indexdb.go:68-132, pkg/kvstore/kcomb_store.go:42-85);
the device states both without the walk, so every case here is a check of that restatement: shards, explicit ids,
high load factors (long runs of full buckets, runs that wrap past the last bucket), long and shared postings sets,
hash collisions among sets (forced), empty and degenerate inputs."""
import os

import numpy as np
import pytest

from kaamer_amd import abi

pytestmark = pytest.mark.gpu


def _bytes(img, tmp_path, name):
    p = os.path.join(str(tmp_path), name)
    img.save(p)
    with open(p, "rb") as f:
        b = f.read()
    os.unlink(p)
    return b


def _same(api, tmp_path, dev, **kw):
    a = api.Image.from_proteins(**kw)
    b = api.Image.from_proteins(device=dev, **kw)
    sa, sb = a.stats(), b.stats()
    assert sa == sb, (sa, sb)
    ba, bb = _bytes(a, tmp_path, "host.kgi"), _bytes(b, tmp_path, "dev.kgi")
    if ba != bb:
        x, y = np.frombuffer(ba, dtype=np.uint8), np.frombuffer(bb, dtype=np.uint8)
        assert len(x) == len(y), (len(x), len(y))
        at = int(np.flatnonzero(x != y)[0])
        where = "header" if at < 4096 else ("buckets" if at < 4096 + sa["n_buckets"] * 64 else "arena")
        raise AssertionError("images differ first at byte %d (%s); %d bytes differ" % (at, where, int((x != y).sum())))
    return sa


def test_small_databases_shards_ids_and_loads(klib, gpu_device, tmp_path):
    from kaamer_amd import api, workload
    db = workload.make_db(3000, seed=31)
    n = len(db[1]) - 1
    st = _same(api, tmp_path, gpu_device, packed=db)
    assert st["n_lists"] > 0 and st["n_inline"] > 0 and st["n_displaced"] > 0
    for shard, n_shards in ((0, 2), (3, 8), (7, 8)):
        _same(api, tmp_path, gpu_device, packed=db, shard=shard, n_shards=n_shards)
    ids = np.random.default_rng(1).permutation(n).astype(np.uint32) * 7 + 5
    _same(api, tmp_path, gpu_device, packed=db, ids=ids)
    # ids above 2^31 cannot be stored inline (kaamer_layout.h): single-id keys become lists of one
    big = ids.copy()
    big[::3] |= 0x80000000
    st = _same(api, tmp_path, gpu_device, packed=db, ids=big)
    assert st["max_protein_id"] >= 0x80000000
    for load in (0.1, 0.75, 0.9, 0.95):
        st = _same(api, tmp_path, gpu_device, packed=db, load_factor=load)
    assert st["n_displaced"] > st["n_keys"] // 20        # at 0.95 the runs of full buckets are long


def test_runs_that_wrap_and_tiny_tables(klib, gpu_device, tmp_path):
    """Many small tables at high load: with few buckets a run of full buckets reaches the last one and goes on at 0."""
    from kaamer_amd import api, workload
    rng = np.random.default_rng(77)
    wrapped = 0
    for case in range(40):
        n = int(rng.integers(1, 60))
        db = workload.make_db(n, seed=1000 + case, family=int(rng.integers(1, 6)))
        load = float(rng.choice([0.5, 0.8, 0.95]))
        st = _same(api, tmp_path, gpu_device, packed=db, load_factor=load)
        img = api.Image.from_proteins(packed=db, load_factor=load, device=gpu_device)
        b = np.frombuffer(_bytes(img, tmp_path, "w.kgi"), dtype=np.uint32)
        keys = b[1024:1024 + st["n_buckets"] * 16:2].reshape(-1, 8)
        full = (keys != 0xFFFFFFFF).all(axis=1)
        wrapped += bool(full[-1] and full[0])
    assert wrapped >= 1


def test_long_and_shared_postings_sets(klib, gpu_device, tmp_path):
    from kaamer_amd import api, workload
    db = workload.make_db_zipf(20000, seed=5, n_motifs=400)
    st = _same(api, tmp_path, gpu_device, packed=db)
    assert st["max_list"] > 1000
    assert st["n_lists"] < st["n_keys"] - st["n_inline"]          # sets are shared
    _same(api, tmp_path, gpu_device, packed=db, shard=1, n_shards=4)


def test_hash_collisions_are_detected_and_retried(klib, gpu_device, tmp_path, capfd):
    """KAAMER_BUILD_WEAK_HASH makes the first attempt's content hash the set's size: unequal sets meet in the table,
    the word-by-word check sees it and the pass runs again under a real seed."""
    from kaamer_amd import api, workload
    db = workload.make_db(2000, seed=8)
    os.environ["KAAMER_BUILD_WEAK_HASH"] = "1"
    os.environ["KAAMER_BUILD_TRACE"] = "1"
    try:
        _same(api, tmp_path, gpu_device, packed=db)
    finally:
        del os.environ["KAAMER_BUILD_WEAK_HASH"]
        del os.environ["KAAMER_BUILD_TRACE"]
    assert "content-hash attempts: 2" in capfd.readouterr().err


def test_degenerate_inputs(klib, gpu_device, tmp_path):
    from kaamer_amd import api
    cases = [
        [],                                                   # no proteins
        [b"ACDEFG"],                                          # shorter than a k-mer (inputFASTA.go:228)
        [b"", b"ACDEFGH", b""],                               # one window, empty neighbours
        [b"ACDEFGH"] * 5,                                     # one key, five ids
        [b"ACDEFGHIKLMNPQRSTVWY" * 3, b"XXXXXXXXXX", b"ACD.FGHIK", b"acdefghik", b"ACDEFGHIKB*ZJO"],   # map misses
        [b"A" * 500, b"A" * 499 + b"C"],                      # a protein that repeats one key
    ]
    for seqs in cases:
        _same(api, tmp_path, gpu_device, seqs=seqs)
        if seqs:
            _same(api, tmp_path, gpu_device, seqs=seqs, shard=1, n_shards=2)
    # offsets that do not start at 0
    buf = np.frombuffer(b"JUNKJUNK" + b"ACDEFGHIKL" + b"MNPQRSTVWYAC", dtype=np.uint8)
    offs = np.array([8, 18, 30], dtype=np.uint64)
    _same(api, tmp_path, gpu_device, packed=(buf, offs))
    with pytest.raises(abi.KaamerError):
        api.Image.from_proteins(seqs=[b"ACDEFGHIK"], ids=np.array([0xFFFFFFFF], dtype=np.uint32), device=gpu_device)


def test_index_built_on_the_device_searches_like_the_image(klib, oracle, gpu_device):
    from kaamer_amd import api, workload
    db = workload.make_db(4000, seed=3)
    q = workload.make_protein_queries(db, 200, seed=4)
    ref = api.Index.from_image(api.Image.from_proteins(packed=db), gpu_device).search(packed=q)
    ix = api.Index.from_proteins(packed=db, device=gpu_device)
    assert ix.stats() == api.Image.from_proteins(packed=db).stats()
    res = ix.search(packed=q)
    oix = oracle.Index.from_proteins(None, packed=db)
    for i, s in enumerate(workload.unpack(q)):
        assert res.hits(i) == ref.hits(i)
        if i % 10 == 0 and oracle.size_in_kmer(s) >= 7:
            pid, km, _ = oix.search(s)
            assert res.hits(i) == dict(zip(pid.tolist(), km.tolist()))


def test_db_sp_size_is_byte_identical(klib, gpu_device, tmp_path):
    """The benchmark database (560 000 proteins, 2 x 10^8 residues): one shard of two, and the whole."""
    from kaamer_amd import api, workload
    db = workload.make_db(560000)
    st = _same(api, tmp_path, gpu_device, packed=db, shard=1, n_shards=2)
    assert st["n_keys"] > 5e7
