#!/usr/bin/env python3
"""bench.py — headline benchmark of the kaamer k-mer search path on MI355X.

Metric (BASELINE.json): k-mer lookups/sec (+ query seqs/sec) and the fraction of
the HBM-bandwidth roofline.  One "step" = one pass of the hot path (prep ->
probe/count kernel -> scan -> gather) over one batch of synthetic queries that
is already resident in HBM.

N = 1 workload: BASELINE.json configs[1] — Swiss-Prot-sized synthetic DB
(560 000 proteins, ~2e8 residues) resident in one MI355X, 10 000 protein
queries per batch (seed 20261003).
N > 1: the DB fits one GPU, so ranks are replicas (no data-path collective):
every rank holds the table and searches its own batch of 10 000 queries;
value = all ranks' lookups / max-over-ranks time ("weak").

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)


def _tame_malloc():
    # first touch of fresh pages is very slow on these VMs: keep freed memory in the heap
    try:
        libc = ctypes.CDLL("libc.so.6")
        libc.mallopt(-3, 2 ** 31 - 1)  # M_MMAP_THRESHOLD
        libc.mallopt(-1, 2 ** 31 - 1)  # M_TRIM_THRESHOLD
    except Exception:
        pass


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def cpu_baseline(oix, queries, seconds=12.0, threads=None, kind="protein"):
    """The CPU restatement of the reference algorithm (oracle/, kind "port"),
    multi-threaded across queries like the reference's nbOfThreads workers
    (search_protein.go:58), on a bounded sample of the same batch."""
    from concurrent.futures import ThreadPoolExecutor
    nq = len(queries[1]) - 1
    threads = threads or min(os.cpu_count() or 1, 16)
    block = 25
    t_start = time.time()

    def work(tid):
        # the batch is cycled until the time budget is spent (same workload, repeated)
        lookups = 0
        nqd = 0
        b = tid * block
        while time.time() - t_start < seconds:
            if b >= nq:
                b = tid * block
            e = min(nq, b + block)
            r = oix.batch(queries, kind, b, e)   # ctypes releases the GIL
            lookups += r["n_lookup"]
            nqd += e - b
            b += threads * block
        return lookups, nqd

    with ThreadPoolExecutor(threads) as ex:
        parts = list(ex.map(work, range(threads)))
    dt = time.time() - t_start
    lookups = sum(p[0] for p in parts)
    nqd = sum(p[1] for p in parts)
    return {"value": lookups / dt, "unit": "k-mer lookups/s", "cores": threads, "kind": "port",
            "sample": "%d %s (%d lookups) in %.1f s, cycling the timed batch of %d; oracle = sorted "
                      "(key,id) array + binary search, not Badger" % (nqd, "reads" if kind == "reads" else "queries", lookups, dt, nq),
            "queries_per_s": nqd / dt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--db-proteins", type=int, default=560000)
    ap.add_argument("--workload", choices=["protein", "reads"], default="protein",
                    help="protein = BASELINE configs[1] (default); reads = configs[2]: 150-nt reads, 6-frame path")
    ap.add_argument("--queries", type=int, default=0, help="sequences per batch (default 10000 proteins / 1000000 reads)")
    ap.add_argument("--load-factor", type=float, default=0.5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--check", type=int, default=50, help="queries checked against the oracle after timing")
    ap.add_argument("--mode", choices=["replicas", "sharded"], default="replicas",
                    help="N>1: replicas = every rank holds the table and its own batch (default; the DB fits one GPU); "
                         "sharded = the table is split by hash prefix, all ranks search one common batch, partial hit "
                         "lists are exchanged with one RCCL all-to-all and merged by the query's owner")
    ap.add_argument("--time-every", type=int, default=8,
                    help="bracket the kernels of every k-th timed step with HIP events (roofline.achieved is their average)")
    ap.add_argument("--post", type=int, default=0,
                    help="1: every step also runs the device post-steps (sortMapByValue, SetBestStartCodon for reads, "
                         "FilterResults: kaamer_topn_device with the reference's defaults)")
    ap.add_argument("--host-api", type=int, default=0,
                    help="1: also time the host-buffer calls once (PCIe-inclusive, informational, never `value`)")
    ap.add_argument("--pipelined-probe", type=int, default=0,
                    help="after the timed region also measure the same batches with three in flight (informational field)")
    ap.add_argument("--inflight", type=int, default=1,
                    help="batches in flight: step i runs on workspace/stream i %% inflight, so the probe kernel of one batch "
                         "(memory-request bound) overlaps the counting kernel of the previous one (LDS/latency bound)")
    ap.add_argument("--compact", type=int, default=0,
                    help="1: finish every batch with the hit lists packed in query order (one more scan + copy pass); "
                         "0 (default): each query's list stays where the counting kernel wrote it (offset + count per query)")
    args = ap.parse_args()

    if args.queries <= 0:
        args.queries = 10000 if args.workload == "protein" else 1000000
    _tame_malloc()
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    sharded_mode = args.mode == "sharded"
    if world > 1 or sharded_mode:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    assert world == args.gpus or world == 1, "launch with torch.distributed.run for --gpus > 1"
    torch.cuda.set_device(local_rank)

    from kaamer_amd import api, workload

    t0 = time.time()
    db = workload.make_db(args.db_proteins)
    log("DB: %d proteins, %d residues (%.1fs)" % (args.db_proteins, int(db[1][-1]), time.time() - t0))
    t0 = time.time()
    img = api.Image.from_proteins(packed=db, load_factor=args.load_factor,
                                  shard=rank if sharded_mode else 0, n_shards=world if sharded_mode else 1)
    st = img.stats()
    log("image built in %.1fs: %s" % (time.time() - t0, st))
    t0 = time.time()
    ix = api.Index.from_image(img, local_rank)
    img.close()
    log("index resident in HBM (%.2f GB) in %.1fs" % ((st["n_buckets"] * 64 + st["arena_words"] * 4) / 1e9, time.time() - t0))

    # every rank searches its own batch (replicas): same generator, rank-specific seed
    from kaamer_amd import abi
    reads = args.workload == "reads"
    if reads:
        q = workload.make_reads(db, args.queries, seed=workload.SEED + 2 + 1000 * rank)
    else:
        q = workload.make_protein_queries(db, args.queries, seed=workload.SEED + 1 + (0 if sharded_mode else 1000 * rank))
    qbuf, qoff = q
    d_buf = torch.from_numpy(qbuf).cuda()
    d_off = torch.from_numpy(qoff.view(np.int64)).cuda()
    ws = api.Workspace(ix, len(qbuf), args.queries,
                       seq_type=abi.READS if reads else abi.PROTEIN,
                       max_hits=(64 << 20) if reads else 0, first_pos=1 if sharded_mode else 0,
                       compact=bool(args.compact))
    stream = torch.cuda.current_stream().cuda_stream
    if sharded_mode:
        assert not reads, "sharded mode: protein workload only in this round"
        from kaamer_amd import sharded
        mws = api.Workspace(ix, len(qbuf), args.queries, first_pos=1, max_hits=16 << 20)
        searcher = sharded.ShardedSearcher(ix, ws, mws, rank, world)

        def step():
            return searcher.step(d_buf, d_off, args.queries, len(qbuf), stream)
    else:
        wss = [ws] + [api.Workspace(ix, len(qbuf), args.queries, seq_type=abi.READS if reads else abi.PROTEIN,
                                    max_hits=(64 << 20) if reads else 0, compact=bool(args.compact))
                      for _ in range(args.inflight - 1)]
        extra_streams = [torch.cuda.Stream() for _ in range(args.inflight - 1)]  # kept alive
        streams = [stream] + [x.cuda_stream for x in extra_streams]
        step_no = [0]

        def step():
            i = step_no[0] % args.inflight
            step_no[0] += 1
            r = wss[i].search_device(d_buf.data_ptr(), d_off.data_ptr(), args.queries, len(qbuf), stream=streams[i])
            if args.post:
                wss[i].topn_device(0.05, 10, 10, best_start_codon=reads, stream=streams[i])
            return r

    for _ in range(args.warmup):
        step()
    counters = (mws if sharded_mode else ws).finish(stream)  # also validates the batch (capacity / overflow)
    if not sharded_mode:
        for w_, s_ in zip(wss[1:], streams[1:]):
            w_.finish(s_)
    ws.set_timing(args.time_every)  # sampled: an event record idles the stream for a few microseconds
    ws.reset_timers()

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t_start
    if sharded_mode:
        mws.finish(stream)
        counters = step()[2]  # the local search's counters (one extra, untimed step)
        mws.finish(stream)
    else:
        for w_, s_ in zip(wss[1:], streams[1:]):
            w_.finish(s_)
        counters = ws.finish(stream)
    tm = ws.kernel_ms_sum()
    n_calls = max(tm["calls"], 1)

    # informational, after the timed region and never `value`: the same batches with three in flight on
    # three streams/workspaces (the next batch's probe kernel overlaps this batch's counting kernel)
    pipelined = None
    if not sharded_mode and world == 1 and args.inflight == 1 and args.pipelined_probe:
        pw = [ws] + [api.Workspace(ix, len(qbuf), args.queries, seq_type=abi.READS if reads else abi.PROTEIN,
                                   max_hits=(64 << 20) if reads else 0, compact=bool(args.compact)) for _ in range(2)]
        pstr_keep = [torch.cuda.Stream() for _ in range(2)]
        pstr = [stream] + [x.cuda_stream for x in pstr_keep]
        ws.set_timing(0)
        n_p = max(30, min(args.steps, 300)) if not reads else 9
        for i in range(6):
            pw[i % 3].search_device(d_buf.data_ptr(), d_off.data_ptr(), args.queries, len(qbuf), stream=pstr[i % 3])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n_p):
            pw[i % 3].search_device(d_buf.data_ptr(), d_off.data_ptr(), args.queries, len(qbuf), stream=pstr[i % 3])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        for w_, s_ in zip(pw, pstr):
            w_.finish(s_)
        pipelined = {"batches_in_flight": 3, "steps": n_p, "ms_per_step": dt / n_p * 1e3,
                     "lookups_per_s": float(counters["n_lookup"]) * n_p / dt}

    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    lk = torch.tensor([float(counters["n_lookup"]), float(args.queries) / (world if sharded_mode else 1)],
                      dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(lk, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    lookups_per_step = float(lk[0].item())
    queries_per_step = float(lk[1].item())

    # ---- roofline of the dominant kernel (probe_kernel), HBM bound ------------------------------
    # algorithmic bytes per launch (DESIGN.md "Measurement"), from exact kernel-side counters:
    #   probe_kernel : 1 B per residue position + 1 bit per position (k-mer-start bitmap)
    #                  + 64 B per bucket inspected + 4 B per position (the val it writes)
    #   count kernels: 4 B per position (val read back) + 4 B per arena word that must be read
    #                  (list header + ids) + 12 B per emitted hit (pid, kmatch, first_pos)
    c = counters
    n_pos = int(c["n_in"]) if reads else int(qoff[-1])  # residue positions kernel P walks (reads: ORF amino acids)
    probe_bytes = n_pos + n_pos // 8 + 64 * c["n_probe"] + 4 * n_pos
    count_bytes = 4 * n_pos + 4 * (c["n_lists"] + c["n_list_ids"]) + 12 * c["n_hits"]
    probe_s = tm["probe_ms"] / n_calls / 1e3
    count_s = tm["count_ms"] / n_calls / 1e3
    achieved = probe_bytes / probe_s / 1e9 if probe_s > 0 else 0.0
    # HBM traffic from the PMC counters is collected in separate rocprofv3 passes (profiles/); it is
    # reported here only when this run is the workload those passes measured
    traffic = None
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic_probe_kernel.json")))
        if (pm["config"]["db_proteins"], pm["config"]["queries"], pm["config"]["workload"]) == \
                (args.db_proteins, args.queries, args.workload):
            traffic = pm["probe_kernel"]["traffic_bytes_per_launch"]
    except Exception:
        pass
    roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                "random_request_ceiling": {"G_requests_per_s": 51.0, "achieved_G_probes_per_s": c["n_probe"] / probe_s / 1e9 if probe_s > 0 else 0.0,
                                           "source": "tools/random_read_bench.hip (profiles/r01_pmc_traffic_probe_kernel.json)"},
                "kernel": "probe_kernel", "kernel_ms": probe_s * 1e3, "timed_launches": n_calls,
                "algorithmic_bytes_per_launch": probe_bytes,
                "bytes_per_lookup": probe_bytes / max(c["n_lookup"], 1),
                "min_bytes_8B_slot": n_pos + n_pos // 8 + 8 * c["n_lookup"] + 4 * n_pos,
                "count_kernels": {"ms": count_s * 1e3, "algorithmic_bytes": count_bytes,
                                  "achieved_GBps": count_bytes / count_s / 1e9 if count_s > 0 else 0.0},
                "whole_batch": {"ms": tm["total_ms"] / n_calls,
                                "achieved_GBps": (probe_bytes + count_bytes) / (tm["total_ms"] / n_calls / 1e3) / 1e9
                                if tm["total_ms"] > 0 else 0.0}}

    out = {
        "metric": "k-mer lookups/sec", "value": lookups_per_step * args.steps / elapsed,
        "unit": "k-mer lookups/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if sharded_mode else "weak",
        "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": ("configs[2]: 6-frame path, %d synthetic 150-nt reads per GPU per step vs " if reads else
                                "configs[1]: %d protein queries per GPU per step vs ") % args.queries +
                               "Swiss-Prot-sized synthetic DB (%d proteins, %d residues, %d distinct 7-mers) resident in HBM"
                               % (args.db_proteins, int(db[1][-1]), st["n_keys"]),
                   "parallelism": ("hash-prefix shards x%d, one all-to-all of partial hit lists per batch" % world) if sharded_mode
                                  else ("replicas x%d (no collective)" % world if world > 1 else "single GPU"),
                   "result": "device-resident per-query hit lists (offset, count, protein ids, Kmatch)" +
                             (", packed in query order" if args.compact else ""),
                   "batches_in_flight": args.inflight, "device_post_steps": bool(args.post),
                   "seed": workload.SEED},
        "query_seqs_per_s": queries_per_step * args.steps / elapsed,
        "counters_per_step_rank0": c,
        "roofline": roofline,
    }
    if pipelined:
        out["pipelined_informational"] = pipelined

    if rank == 0:
        want_cpu = not args.no_cpu_baseline and world == 1 and not sharded_mode
        if sharded_mode:
            args.check = 0
        if args.check or want_cpu or args.host_api:
            from oracle import oracle as O  # the checker / the reported CPU baseline, never the product
            t0 = time.time()
            oix = O.Index.from_proteins(None, packed=db)
            log("oracle index built in %.1fs" % (time.time() - t0))
            if args.check:
                # the device-resident result of the last TIMED step, read back as it lies in HBM
                from kaamer_amd.sharded import dev_tensor
                sub = workload.unpack(q)[:args.check]
                nq_dev = int(counters["n_queries"])
                cap = int(last.hit_capacity)
                r_off = dev_tensor(last.d_hit_off, nq_dev, torch.int64).cpu().numpy()
                r_cnt = dev_tensor(last.d_hit_cnt, nq_dev, torch.int32).cpu().numpy()
                r_pid = dev_tensor(last.d_hit_pid, cap, torch.int32).cpu().numpy().view(np.uint32)
                r_km = dev_tensor(last.d_hit_kmatch, cap, torch.int32).cpu().numpy()

                class res:
                    n_queries = nq_dev

                    @staticmethod
                    def hits(i):
                        a = int(r_off[i])
                        return dict(zip(r_pid[a:a + int(r_cnt[i])].tolist(), r_km[a:a + int(r_cnt[i])].tolist()))
                if reads:
                    qi = 0
                    for s in sub:
                        for o in O.get_orfs(s):
                            pid, km, _ = oix.search(o["seq"])
                            assert res.hits(qi) == dict(zip(pid.tolist(), km.tolist())), "bench: ORF %d differs from the oracle" % qi
                            qi += 1
                    assert qi <= res.n_queries
                else:
                    for i, s in enumerate(sub):
                        exp = {}
                        if O.size_in_kmer(s) >= 7:
                            pid, km, _ = oix.search(s)
                            exp = dict(zip(pid.tolist(), km.tolist()))
                        assert res.hits(i) == exp, "bench: query %d differs from the oracle" % i
                out["parity_checked_queries"] = len(sub)
                log("parity: %d queries of the timed batch bit-exact vs oracle" % len(sub))
            if args.host_api:
                hb = {}
                for name, fn in (("search_batch_top", lambda: ix.search_top(packed=q, seq_type=abi.READS if reads else abi.PROTEIN)),
                                 ("search_batch", lambda: ix.search(packed=q, seq_type=abi.READS if reads else abi.PROTEIN))):
                    fn()
                    t0 = time.perf_counter()
                    fn()
                    hb[name] = {"ms": (time.perf_counter() - t0) * 1e3,
                                "lookups_per_s": c["n_lookup"] / (time.perf_counter() - t0)}
                out["host_buffer_calls_pcie_inclusive"] = hb
            if want_cpu:
                out["cpu_baseline"] = cpu_baseline(oix, q, seconds=args.cpu_seconds, kind="reads" if reads else "protein")
        print(json.dumps(out), flush=True)
    if world > 1 or sharded_mode:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
