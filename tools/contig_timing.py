import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, numpy as np, torch
from kaamer_amd import api, abi, workload
db = workload.make_db(20000, seed=3)
ix = api.Index.from_image(api.Image.from_proteins(packed=db), 0)
rng = np.random.default_rng(1)
for n in (100_000, 1_000_000, 5_000_000):
    contig = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n)]
    buf = contig.copy(); offs = np.array([0, n], dtype=np.uint64)
    d_buf = torch.from_numpy(buf).cuda(); d_off = torch.from_numpy(offs.view(np.int64)).cuda()
    ws = api.Workspace(ix, n, 1, seq_type=abi.NUCLEOTIDE)
    st = torch.cuda.current_stream().cuda_stream
    ws.set_timing(1)
    for _ in range(2):
        ws.search_device(d_buf.data_ptr(), d_off.data_ptr(), 1, n, stream=st)
    c = ws.finish(st)
    ws.reset_timers()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        ws.search_device(d_buf.data_ptr(), d_off.data_ptr(), 1, n, stream=st)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    c = ws.finish(st); tm = ws.kernel_ms_sum()
    print("contig %d nt: %.2f ms per search (%d ORFs, %d lookups); probe %.3f ms count %.3f ms -> translate+rest %.2f ms" % (
        n, dt * 1e3, c["n_queries"], c["n_lookup"], tm["probe_ms"] / tm["calls"], tm["count_ms"] / tm["calls"], dt * 1e3 - (tm["probe_ms"] + tm["count_ms"]) / tm["calls"]))
