"""BASELINE configs[1] at full size (560 000-protein DB, 10 000 queries): the oracle would need
minutes here, so the checks are size-independent properties of the path:
  * a protein searched against a DB that contains it hits itself with Kmatch = len - 6
    (every window of the query is a window of the record; unknown letters alias consistently);
  * sum of Kmatch over all hits == postings expanded (n_post): every (position, id) is counted once;
  * n_lookup == sum of SizeInKmer over the searched queries;
  * the packed (CSR) and the in-place result forms hold the same hit sets, and a second run of the
    same batch is identical (per-batch device state is left clean);
  * the device top-N keeps a prefix of the (Kmatch desc, id asc) order and obeys MinKMatch / MinKRatio."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _from_ptr(ptr, n, dtype):
    from test_gpu_protein import _from_ptr as f
    return f(ptr, n, dtype)


def test_config1_properties(klib, gpu_device):
    import torch
    from kaamer_amd import api, workload
    db = workload.make_db(560000)
    ix = api.Index.from_image(api.Image.from_proteins(packed=db), gpu_device)
    qbuf, qoff = workload.make_protein_queries(db, 9900, seed=workload.SEED + 1)
    # + 100 unmodified DB records as queries (ids 0, 5600, 11200, ...)
    dbuf, doff = db
    own = list(range(0, 560000, 5600))
    extra = [bytes(dbuf[int(doff[i]):int(doff[i + 1])]) for i in own]
    seqs = workload.unpack((qbuf, qoff)) + extra
    buf, offs = api.pack_sequences(seqs)
    n = len(seqs)
    d_buf = torch.from_numpy(buf).cuda()
    d_off = torch.from_numpy(offs.view(np.int64)).cuda()
    st = torch.cuda.current_stream().cuda_stream
    lens = np.diff(offs.astype(np.int64))
    size = lens - 6 - np.array([1 if s.endswith(b"*") else 0 for s in seqs])

    def run(ws):
        r = ws.search_device(d_buf.data_ptr(), d_off.data_ptr(), n, len(buf), stream=st)
        c = ws.finish(st)
        off = _from_ptr(r.d_hit_off, n, np.uint64).astype(np.int64)
        cnt = _from_ptr(r.d_hit_cnt, n, np.uint32).astype(np.int64)
        pid = _from_ptr(r.d_hit_pid, int(r.hit_capacity), np.uint32)
        km = _from_ptr(r.d_hit_kmatch, int(r.hit_capacity), np.uint32)
        return c, off, cnt, pid, km

    ws = api.Workspace(ix, len(buf), n)
    c, off, cnt, pid, km = run(ws)
    assert c["n_queries"] == n and c["n_overflow"] >= 0
    assert c["n_lookup"] == int(size[size >= 7].sum())
    idx = np.concatenate([np.arange(o, o + k) for o, k in zip(off, cnt)]) if cnt.sum() else np.zeros(0, np.int64)
    assert int(km[idx].sum()) == c["n_post"]                      # every (position, id) counted exactly once
    assert int(cnt.sum()) == c["n_hits"]
    for j, i in enumerate(own):                                   # self hits
        q = 9900 + j
        a, b = int(off[q]), int(off[q]) + int(cnt[q])
        hits = dict(zip(pid[a:b].tolist(), km[a:b].tolist()))
        assert hits[i] == size[q] == max(hits.values()), (i, size[q])
    # second run, and the packed form
    c2, off2, cnt2, pid2, km2 = run(ws)
    assert c2 == c and (cnt2 == cnt).all()
    wsc = api.Workspace(ix, len(buf), n, compact=True)
    c3, off3, cnt3, pid3, km3 = run(wsc)
    assert c3 == c and (cnt3 == cnt).all()
    assert off3[0] == 0 and (np.diff(off3) == cnt3[:-1]).all()    # CSR
    for q in list(range(0, n, 97)) + list(range(9900, n)):
        a, b = int(off[q]), int(off[q]) + int(cnt[q])
        a3 = int(off3[q])
        assert dict(zip(pid[a:b].tolist(), km[a:b].tolist())) == dict(zip(pid3[a3:a3 + b - a].tolist(), km3[a3:a3 + b - a].tolist()))
    # device post-steps on the same result
    t = ws.topn_device(0.05, 10, 10, stream=st)
    ws.finish(st)
    tc = _from_ptr(t.d_top_cnt, n, np.uint32)
    tp = _from_ptr(t.d_top_pid, n * 10, np.uint32).reshape(n, 10)
    tk = _from_ptr(t.d_top_kmatch, n * 10, np.uint32).reshape(n, 10)
    for q in range(0, n, 53):
        a, b = int(off[q]), int(off[q]) + int(cnt[q])
        order = sorted(zip((-km[a:b].astype(np.int64)).tolist(), pid[a:b].tolist()))
        exp = [(p, -k) for k, p in order if -k >= 10 and (-k) / float(size[q]) >= 0.05][:10]
        assert tc[q] == len(exp)
        assert list(zip(tp[q, :len(exp)].tolist(), tk[q, :len(exp)].tolist())) == exp
