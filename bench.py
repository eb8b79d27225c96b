#!/usr/bin/env python3
"""bench.py — headline benchmark of the kaamer k-mer search path on MI355X.

Metric (BASELINE.json): k-mer lookups/sec (+ query seqs/sec) and the fraction of
the HBM-bandwidth roofline.

N = 1 workload: BASELINE.json configs[1] — Swiss-Prot-sized synthetic DB
(560 000 proteins, ~2e8 residues) resident in one MI355X, batches of 10 000
protein queries (seed 20261003), inputs resident in HBM.

One STEP = `--batches-per-step` batches (default 200 x 10 000 protein queries, or
3 x 1 M reads) pushed back to back through the hot path (prep -> search kernel
[probe + count] -> G tier/finalize), rotating over `--distinct-batches` different
pre-generated batches so that consecutive batches touch different buckets (one
batch reads ~0.46 GB of buckets, the Infinity Cache holds 0.27 GB, the table is
4 GB).  A step therefore lasts ~25 ms and the driver's 20 steps give a 0.5 s
timed region; lookups/s does not depend on how batches are grouped into steps.

N > 1: the DB fits one GPU, so ranks are replicas (no data-path collective):
every rank holds the table and searches its own batches; value = all ranks'
lookups / max-over-ranks time ("weak").  `--mode sharded` runs the hash-prefix
sharded index of configs[3] instead (one exchange step per batch).

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

BUCKET_BYTES = 64      # kaamer_layout.h: 8 slots of {u32 key, u32 val}
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


def host_cores():
    """CPUs this process may actually use: the affinity mask, cut down to the cgroup's CPU quota when there is one
    (a GPU box shows all of the host's CPUs in the mask but schedules the container on a share of them)."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            f = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = f[0], float(f[1])
            else:
                quota, period = f[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                n = max(1, min(n, int(float(quota) / period + 0.5)))
            break
        except Exception:
            continue
    return n


def cpu_baseline(oix, queries, seconds=12.0, threads=None, kind="protein"):
    """The CPU restatement of the reference algorithm (oracle/, kind "port"),
    multi-threaded across queries like the reference's nbOfThreads workers
    (search_protein.go:58; the reference's default is runtime.NumCPU(), api/server.go:55-59),
    on a bounded sample of the same batch, on every host core this process may use."""
    from concurrent.futures import ThreadPoolExecutor
    nq = len(queries[1]) - 1
    threads = threads or host_cores()
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
                if b >= nq:
                    break
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
            "sample": "%d %s (%d lookups) in %.1f s, cycling one timed batch of %d; oracle = sorted "
                      "(key,id) array + binary search, not Badger" % (nqd, "reads" if kind == "reads" else "queries", lookups, dt, nq),
            "queries_per_s": nqd / dt}


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: this process -- which has not imported torch nor touched the GPU --
    starts N fresh children, one rank per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as torch.distributed.run sets
    them), relays rank 0's JSON line and returns non-zero if any rank did.  Nothing is re-executed in place."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    line = procs[0].stdout.read().decode()
    rcs = [p.wait() for p in procs]
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        print("[bench] ranks failed (rank, exit code): %s" % bad, file=sys.stderr, flush=True)
        if line.strip():
            print("[bench] rank 0 printed: %s" % line.strip()[:2000], file=sys.stderr, flush=True)
        return 1
    sys.stdout.write(line)
    sys.stdout.flush()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--db-proteins", type=int, default=560000)
    ap.add_argument("--db", choices=["sp", "zipf", "zipf-mid", "ur-lite"], default=None,
                    help="sp = DB-SP of SURVEY 8d (default); zipf = the same size with Zipf-distributed shared motifs "
                         "(postings lists of 1e4 and more), zipf-mid = a middle skew (longest list ~1e3): reported secondary "
                         "lines, never the headline; ur-lite = the DB-SP generator scaled to --ur-residues (default 5e9: "
                         "a database that is actually sharded; every rank builds only its shard).  Default: sp; with --mode sharded "
                         "ur-lite at 1e9 residues per GPU (13 x DB-SP per 8 GPUs: every device holds 1/N of a table that grows with N)")
    ap.add_argument("--ur-residues", type=float, default=0.0, help="residues of --db ur-lite (default: 1e9 per GPU in sharded mode, else 5e9)")
    ap.add_argument("--exchange-at-w1", type=int, default=0,
                    help="sharded mode at N = 1: 1 = run pack / all-to-all / merge although the one shard's lists are the results "
                         "(profiling the exchange code on one GPU); 0 (default) = the post-steps run on the search results")
    ap.add_argument("--workload", choices=["protein", "reads", "mix"], default="protein",
                    help="protein = BASELINE configs[1] (default); reads = configs[2]: 150-nt reads, 6-frame path; "
                         "mix = Q-mix of configs[4]: 100/150/250-nt reads + 5 %% long reads")
    ap.add_argument("--queries", type=int, default=0, help="sequences per batch (default 10000 proteins / 1000000 reads)")
    ap.add_argument("--batches-per-step", type=int, default=0,
                    help="batches pushed back to back in one step (default 200 protein batches / 3 read batches: ~25 ms)")
    ap.add_argument("--distinct-batches", type=int, default=8,
                    help="different pre-generated batches the steps rotate over (their bucket footprint exceeds the Infinity Cache)")
    ap.add_argument("--load-factor", type=float, default=0.5)
    ap.add_argument("--g-tier-slots", type=int, default=0,
                    help="workspace bound: slots of the HBM counting tables of queries that overflow their LDS table "
                         "(default: the library's 32 M; 1 G for --db zipf, where most queries overflow)")
    ap.add_argument("--max-hits", type=int, default=0, help="workspace bound: hit entries per batch (default: sized for the workload)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--check", type=int, default=50, help="queries checked against the oracle after timing")
    ap.add_argument("--mode", choices=["replicas", "sharded"], default="replicas",
                    help="N>1: replicas = every rank holds the table and its own batches (default; the DB fits one GPU); "
                         "sharded = the table is split by hash prefix, all ranks search one common batch, partial hit "
                         "lists are exchanged once per batch and merged by the query's owner")
    ap.add_argument("--transport", choices=["rccl", "torch"], default="rccl",
                    help="sharded mode: rccl = kaamer_rccl_alltoall on an ncclComm_t made with ncclCommInitRank (what a Go host "
                         "calls); torch = torch.distributed.all_to_all_single (backend nccl = RCCL)")
    ap.add_argument("--sharded-leg", type=int, default=1,
                    help="replicas mode with N > 1: after the timed region also run a short pass of the hash-prefix sharded index "
                         "(configs[3]'s data path: search -> pack -> RCCL all-to-all -> merge -> top-N) on the same database and "
                         "report it under `sharded_leg` (per-phase ms, bytes exchanged, the rank count RCCL saw); 0 = skip; "
                         "2 = run it at N = 1 too (rehearsal of the code path on one GPU)")
    ap.add_argument("--time-every", type=int, default=16,
                    help="bracket the kernels of every k-th timed launch with HIP events (roofline.achieved is their average)")
    ap.add_argument("--post", type=int, default=0,
                    help="1: every batch also runs the device post-steps (sortMapByValue, SetBestStartCodon for reads, "
                         "FilterResults: kaamer_topn_device with the reference's defaults)")
    ap.add_argument("--host-api", type=int, default=0,
                    help="1: also time the host-buffer calls once (PCIe-inclusive, informational, never `value`)")
    ap.add_argument("--exchange-entries", type=int, default=0,
                    help="sharded mode: partial hit entries one exchange block holds (default: sized for the workload)")
    ap.add_argument("--inflight", type=int, default=0,
                    help="batches in flight: batch i runs on workspace/stream i %% inflight.  Default 3 for protein batches "
                         "(the probe kernel is bound by memory requests, the counting kernel by latency: neighbouring batches "
                         "overlap, +32 %% throughput) and for 1 M-read batches (+8-12 %%), 4 on the skewed database, 1 in sharded mode")
    ap.add_argument("--probe-streams", type=int, default=0,
                    help="0 (default): every batch in flight has a stream of its own.  P >= 1: prep + probe of ALL batches go to P "
                         "streams (round-robin) and the counting stages to --count-streams streams (kaamer_workspace_set_count_stream)")
    ap.add_argument("--count-streams", type=int, default=2)
    ap.add_argument("--compact", type=int, default=0,
                    help="1: finish every batch with the hit lists packed in query order (one more scan + copy pass); "
                         "0 (default): each query's list stays where the search kernel wrote it (offset + count per query)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    # stdout carries ONE line, the JSON: everything else a library prints there (RCCL's version banner at
    # communicator creation, for one) goes to stderr
    json_fd = os.dup(1)
    os.dup2(2, 1)
    sys.stdout = os.fdopen(os.dup(2), "w")

    def emit(line):
        os.write(json_fd, (line + "\n").encode())

    if args.db is None:
        args.db = "ur-lite" if args.mode == "sharded" else "sp"
    if args.ur_residues <= 0:
        args.ur_residues = 1e9 * max(1, args.gpus) if args.mode == "sharded" else 5e9
    nucl = args.workload in ("reads", "mix")
    if args.queries <= 0:
        args.queries = 1000000 if nucl else 10000
    if args.batches_per_step <= 0:
        args.batches_per_step = 3 if nucl else (20 if args.db.startswith("zipf") else 200)
    if args.inflight <= 0:
        # batches in flight, each on its own stream and workspace.  The skewed databases gain most: their batches end in
        # a tail of a few monster queries' workgroups, which the next batches' kernels fill (1 / 2 / 3 / 4 / 6 in flight:
        # 4.28 / 2.32 / 1.67 / 1.42 / 1.78 ms per batch); 1 M-read batches gain 8-12 % (round 4, same box: 7.10 ms with one
        # in flight, 6.25-6.56 with three: the probe kernel of batch i+1 runs next to the counting kernel of batch i)
        # (sharded mode: three sharded steps in flight -- the exchange of one batch travels while the next is searched and the
        # one before is merged; world 1 with the exchange forced: 0.386 / 0.29 / 0.243 ms per protein batch with 1 / 2 / 3)
        args.inflight = 3 if args.mode == "sharded" else (4 if args.db == "zipf" else 3)
    if args.db.startswith("zipf"):
        args.g_tier_slots = args.g_tier_slots or (1 << 30)
        args.max_hits = args.max_hits or (1 << 28)
    if args.db == "ur-lite":
        args.db_proteins = int(args.ur_residues / 358.5)   # mean length of workload._lengths
    if args.db_proteins > 3000000:
        args.check = 0   # the oracle's index of the whole database does not fit the bounded check of a bench run
    _tame_malloc()
    import numpy as np
    ur_db = None
    if args.db == "ur-lite":
        # generated by a pool of processes, which must be over before this process touches the GPU
        from kaamer_amd import workload as _w
        t0 = time.time()
        ur_db = _w.make_db_parallel(args.db_proteins, workers=max(1, (os.cpu_count() or 2) // max(1, int(os.environ.get("WORLD_SIZE", "1"))) - 1))
        log("DB (ur-lite) generated in %.1fs" % (time.time() - t0))
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:   # a line whose n_gpus is not what was asked for would read as a flat scaling curve
        log("error: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
        sys.exit(2)
    sharded_mode = args.mode == "sharded"
    if world > 1 or sharded_mode:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)

    from kaamer_amd import abi, api, workload

    t0 = time.time()
    if args.db == "zipf":
        db = workload.make_db_zipf(args.db_proteins)
    elif args.db == "zipf-mid":   # a tenth of the motif reuse: longest postings list ~1e3
        db = workload.make_db_zipf(args.db_proteins, zipf_a=0.45, per_residues=240)
    elif ur_db is not None:
        db = ur_db
    else:
        db = workload.make_db(args.db_proteins)
    log("DB (%s): %d proteins, %d residues (%.1fs)" % (args.db, args.db_proteins, int(db[1][-1]), time.time() - t0))
    t0 = time.time()
    # the table is built on the device it is searched on (builder_device.hip; byte-identical to the host builder's image)
    if os.environ.get("KAAMER_EXP_ARENA_ORDER"):   # experiment: the image comes back to the host and is re-laid there at open time
        img_ = api.Image.from_proteins(packed=db, load_factor=args.load_factor, device=local_rank)
        if os.environ["KAAMER_EXP_ARENA_ORDER"] == "2":   # lists in first-touch order of a walk over the proteins' windows
            t0 = time.time()
            L_ = abi.lib()
            L_.kaamer_exp_relayout_first_touch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
            b_, o_ = np.ascontiguousarray(db[0]), np.ascontiguousarray(db[1])
            abi.check(L_.kaamer_exp_relayout_first_touch(img_._h, b_.ctypes.data, o_.ctypes.data, len(o_) - 1))
            os.environ["KAAMER_EXP_ARENA_ORDER"] = "0"   # (the open-time hook then only compacts in the order it finds)
            log("experiment: arena re-laid in first-touch order in %.1fs" % (time.time() - t0))
        ix = api.Index.from_image(img_, local_rank)
    else:
        ix = api.Index.from_proteins(packed=db, load_factor=args.load_factor, device=local_rank,
                                     shard=rank if sharded_mode else 0, n_shards=world if sharded_mode else 1)
    st = ix.stats()
    table_bytes = st["n_buckets"] * BUCKET_BYTES + st["arena_words"] * 4
    log("index built on the device and resident in HBM (%.2f GB) in %.1fs: %s" % (table_bytes / 1e9, time.time() - t0, st))

    # ---- the batches: same generator, a different seed per batch (and per rank: replicas search their own batches)
    n_distinct = max(1, args.distinct_batches)
    seq_type = abi.READS if nucl else abi.PROTEIN
    batches = []
    for b in range(n_distinct):
        seed = workload.SEED + 1 + 17 * b + (0 if sharded_mode else 1000 * rank)
        if args.workload == "reads":
            q = workload.make_reads(db, args.queries, seed=seed + 1)
        elif args.workload == "mix":
            q = workload.make_reads_mix(db, args.queries, seed=seed + 1)
        else:
            q = workload.make_protein_queries(db, args.queries, seed=seed)
        batches.append(q)
    d_bufs = [torch.from_numpy(q[0]).cuda() for q in batches]
    d_offs = [torch.from_numpy(q[1].view(np.int64)).cuda() for q in batches]
    max_bytes = max(len(q[0]) for q in batches)
    if args.db == "ur-lite" and not args.max_hits:
        # a dense database: most k-mers meet several proteins by chance, the counting tables grow with that (the library
        # scales them from the previous batch's hits per k-mer, as far as the batch's positions leave room in the hit arrays)
        args.max_hits = int(2.5 * int(args.queries * (125 if nucl else 365) * (0.065 + 2.18 * float(int(db[1][-1])) / 1e9) * 1.3))
    ws_kw = dict(seq_type=seq_type, max_hits=args.max_hits or ((64 << 20) if nucl else 0), g_tier_slots=args.g_tier_slots,
                 compact=bool(args.compact), concurrent_batches=args.inflight)
    stream = torch.cuda.current_stream().cuda_stream

    def exchange_entries():
        """entries one (source, destination) block holds.  A batch's partial hits grow with the database: besides the
        homologues, every k-mer of a query meets the proteins that hold it by chance (pairs of the database / 21^7 per
        lookup), which on DB-UR-lite outnumber the rest"""
        if args.exchange_entries:
            return args.exchange_entries
        return max((96 << 20) if nucl else (4 << 20), partial_entries()) // world + (1 << 16)

    def partial_entries():
        lookups = args.queries * (125 if nucl else 365)
        return int(lookups * (0.065 + 2.18 * float(int(db[1][-1])) / 1e9) * 1.3)   # measured: 0.5 hits per k-mer on DB-SP (2e8 residues), 2.24 at 1e9

    def shard_hits():   # hit-list capacity of a rank's search workspace (0: the library's default)
        if args.db == "ur-lite":
            return int(2.5 * partial_entries()) // world + (1 << 20)
        return max(64 << 20, partial_entries() // world + (1 << 20)) if nucl else 0

    def exchange_report(searcher, tstream, bufs, offs, sizes, n_pass=6, final_batch=None):
        """per-phase times of the sharded step (events around every phase, one batch at a time), what travelled, and
        how many ranks the RCCL communicator has"""
        from kaamer_amd import sharded
        searcher.phase_ms = None
        for i in range(n_pass):
            b = i % len(bufs)
            searcher.step(bufs[b].data_ptr(), offs[b].data_ptr(), args.queries, sizes[b], tstream,
                          topn={}, timed=True)
        searcher.finish(tstream)
        ph = searcher.phase_ms
        final = None
        if final_batch is not None:   # leave the results of this batch in the workspaces (the parity check reads them)
            final = searcher.step(bufs[final_batch].data_ptr(), offs[final_batch].data_ptr(), args.queries, sizes[final_batch], tstream)
            searcher.finish(tstream)
        if searcher.direct:   # one shard: its lists are the results, nothing is packed or exchanged
            return {"transport": "none (one shard: the partial lists are the results; --exchange-at-w1 1 runs the exchange anyway)",
                    "rccl_comm_ranks": searcher.comm.world if searcher.comm is not None else None,
                    "phase_ms_per_batch": {k: ph[k] / ph["batches"] for k in sharded.PHASES}, "phase_batches": ph["batches"],
                    "wire_bytes_per_rank_per_batch": 0, "payload_bytes_per_rank_per_batch": 0}, final
        L = searcher.wire   # the layout of the last batch: blocks sized from what an earlier batch needed, not the capacity
        bw, W = int(L.block_words), int(L.world)
        hdr = torch.stack([searcher.send[d * bw:d * bw + 8] for d in range(W)]).cpu().numpy()   # [entries, status, nq, owned, need, max need, -, -]
        words = 3 if searcher.ws_first_pos else 2
        payload = int(sum(4 * (8 + int(h[3]) + words * int(h[0])) for h in hdr))
        wire = W * 4 * bw
        return {"transport": searcher.transport,
                "rccl_comm_ranks": searcher.comm.world if searcher.comm is not None else None,
                "phase_ms_per_batch": {k: ph[k] / ph["batches"] for k in sharded.PHASES},
                "phase_batches": ph["batches"],
                "wire_bytes_per_rank_per_batch": wire,          # what the all-to-all moves: world equal blocks of this batch's layout
                "payload_bytes_per_rank_per_batch": payload,     # headers + counts + entries actually packed
                "wire_over_payload": wire / max(payload, 1),
                "capacity_bytes_per_rank": W * searcher.block_bytes,   # what the buffers could hold (never travels as such)
                "adaptive_blocks": bool(searcher.adaptive),
                "block_entry_capacity": int(L.e_cap), "entries_packed_per_block": [int(h[0]) for h in hdr]}, final

    if sharded_mode:
        from kaamer_amd import sharded
        pipe = sharded.ShardedPipeline(max(1, args.inflight), ix, rank, world, max_bytes, args.queries, seq_type=seq_type,
                                       max_entries_per_peer=exchange_entries(), max_hits=shard_hits(),
                                       transport=args.transport, direct_at_world1=not args.exchange_at_w1)
        searcher, tstream = pipe.searchers[0], pipe.streams[0]

        def launch(i):
            b = i % n_distinct
            k, r = pipe.step(d_bufs[b].data_ptr(), d_offs[b].data_ptr(), args.queries, len(batches[b][0]))
            if args.post:
                pipe.searchers[k].topn(pipe.streams[k])
            return r
        wss, streams = [searcher.ws], [stream]
    else:
        wss = [api.Workspace(ix, max_bytes, args.queries, **ws_kw) for _ in range(args.inflight)]
        if args.probe_streams > 0:
            # dedicated roles: the probe kernels of all batches queue on a few probe streams, the counting stages on count streams
            pst = [torch.cuda.Stream() for _ in range(args.probe_streams)]
            cst = [torch.cuda.Stream() for _ in range(max(1, args.count_streams))]
            extra_streams = pst + cst
            streams = [pst[w % len(pst)].cuda_stream for w in range(args.inflight)]
            for w, ws_ in enumerate(wss):
                ws_.set_count_stream(cst[w % len(cst)].cuda_stream)
        else:
            extra_streams = [torch.cuda.Stream() for _ in range(args.inflight - 1)]  # kept alive
            streams = [stream] + [x.cuda_stream for x in extra_streams]

        def launch(i):
            b, w = i % n_distinct, i % args.inflight
            r = wss[w].search_device(d_bufs[b].data_ptr(), d_offs[b].data_ptr(), args.queries, len(batches[b][0]), stream=streams[w])
            if args.post:
                wss[w].topn_device(0.05, 10, 10, best_start_codon=nucl, stream=streams[w])
            return r

    def finish_all():
        c = None
        if sharded_mode:
            return pipe.finish()[pipe.last_k][0]
        for w_, s_ in zip(wss, streams):
            c = w_.finish(s_)   # also validates the batch (capacity / overflow)
        return c

    # exact work counters of every distinct batch (counted by the kernels), one untimed pass each
    per_batch = []
    for b in range(n_distinct):
        if sharded_mode:
            launch(b)
            per_batch.append(pipe.finish()[pipe.last_k][0])
        else:
            wss[0].search_device(d_bufs[b].data_ptr(), d_offs[b].data_ptr(), args.queries, len(batches[b][0]), stream=streams[0])
            per_batch.append(wss[0].finish(streams[0]))
    n_launch = 0
    for _ in range(args.warmup * args.batches_per_step):
        launch(n_launch)
        n_launch += 1
    finish_all()
    wss[0].set_timing(args.time_every)  # sampled: an event record idles the stream for a few microseconds
    wss[0].reset_timers()

    def timed_region():
        nonlocal n_launch
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0_ = time.perf_counter()
        first_ = n_launch
        last_ = None
        for _ in range(args.steps):
            for _ in range(args.batches_per_step):
                last_ = launch(n_launch)
                n_launch += 1
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        return time.perf_counter() - t0_, first_, last_

    elapsed, first_timed, last = timed_region()
    try:
        finish_all()
    except abi.KaamerError as e:
        # sharded mode sizes the exchange blocks from what earlier batches needed: a batch that outgrew its blocks fails on
        # every rank alike (the overflow is in every header).  The measurement is then repeated with capacity-sized
        # blocks and says so in `exchange.adaptive_blocks`; anything else is an error
        if not (sharded_mode and e.code == abi.E_CAPACITY and searcher.adaptive):
            raise
        log("a batch outgrew its adaptive exchange blocks: timing again with capacity-sized blocks")
        pipe.set_adaptive(False)
        launch(n_launch); n_launch += 1
        finish_all()
        elapsed, first_timed, last = timed_region()
        finish_all()
    last_batch = (n_launch - 1) % n_distinct
    tm = wss[0].kernel_ms_sum()
    n_calls = max(tm["calls"], 1)
    # with several batches in flight the kernels of neighbouring batches share the device: their durations in the
    # timed region say how the overlap went, not what a kernel can do.  A short untimed pass with ONE batch in flight
    # gives every kernel alone on the device (reported next to the timed-region figures)
    tm_alone = None
    if args.inflight > 1 and not sharded_mode:
        # (a workspace of its own: the timed region's workspaces were created for batches in flight next to each other --
        # concurrent_batches -- and launch the counting kernel accordingly)
        ws1 = api.Workspace(ix, max_bytes, args.queries, **dict(ws_kw, concurrent_batches=1))
        for b in range(min(2, n_distinct)):
            ws1.search_device(d_bufs[b].data_ptr(), d_offs[b].data_ptr(), args.queries, len(batches[b][0]), stream=streams[0])
        ws1.finish(streams[0])
        ws1.set_timing(1)
        ws1.reset_timers()
        for i in range(96):
            b = i % n_distinct
            ws1.search_device(d_bufs[b].data_ptr(), d_offs[b].data_ptr(), args.queries, len(batches[b][0]), stream=streams[0])
            if args.post:
                ws1.topn_device(0.05, 10, 10, best_start_codon=nucl, stream=streams[0])
        ws1.finish(streams[0])
        tm_alone = ws1.kernel_ms_sum()
        ws1.close()

    exch = None
    if sharded_mode:
        exch, last = exchange_report(searcher, tstream, d_bufs, d_offs, [len(q[0]) for q in batches], final_batch=last_batch)

    def sharded_leg(n_batches=12):
        """configs[3]'s data path on this run's database: every rank builds and holds only its hash-prefix shard, all
        ranks search ONE common batch, partial hit lists go to the query's owner over RCCL, the owner merges and runs
        the post-steps.  A short secondary measurement next to the replicas headline (strong scaling: the batch is fixed)."""
        from kaamer_amd import sharded
        t0 = time.time()
        six = api.Index.from_proteins(packed=db, load_factor=args.load_factor, shard=rank, n_shards=world, device=local_rank)
        sst = six.stats()
        build_s = time.time() - t0
        cb = []
        for b in range(2):  # the same batches on every rank
            seed = workload.SEED + 501 + b
            cb.append(workload.make_reads(db, args.queries, seed=seed) if args.workload == "reads" else
                      workload.make_reads_mix(db, args.queries, seed=seed) if args.workload == "mix" else
                      workload.make_protein_queries(db, args.queries, seed=seed))
        cbuf = [torch.from_numpy(q[0]).cuda() for q in cb]
        coff = [torch.from_numpy(q[1].view(np.int64)).cuda() for q in cb]
        sizes = [len(q[0]) for q in cb]
        # three sharded steps in flight (searchers, streams and communicators of their own): a batch's blocks travel while
        # the next batch is searched and the one before is merged
        sp = sharded.ShardedPipeline(3, six, rank, world, max(sizes), args.queries, seq_type=seq_type,
                                     max_entries_per_peer=exchange_entries(), max_hits=shard_hits(),
                                     transport=args.transport)
        ss, ts = sp.searchers[0], sp.streams[0]
        lookups = []
        for b in range(6):   # (each searcher sees both batches: its block layout follows ITS earlier batches)
            sp.step(cbuf[b % 2].data_ptr(), coff[b % 2].data_ptr(), args.queries, sizes[b % 2], topn={})
            c_ = sp.finish()[sp.last_k]
            if b < 2:
                lookups.append(c_[0]["n_lookup"])
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(n_batches):
            sp.step(cbuf[i % 2].data_ptr(), coff[i % 2].data_ptr(), args.queries, sizes[i % 2], topn={})
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t1
        sp.finish()
        tt = torch.tensor([dt, float(sum(lookups[i % 2] for i in range(n_batches)))], dtype=torch.float64, device="cuda")
        if world > 1:
            dist.all_reduce(tt[:1], op=dist.ReduceOp.MAX)
            dist.all_reduce(tt[1:], op=dist.ReduceOp.SUM)   # every rank looks up only the keys its shard owns
        rep, _ = exchange_report(ss, ts, cbuf, coff, sizes)
        rep.update({"ms_per_batch": float(tt[0]) / n_batches * 1e3, "lookups_per_s": float(tt[1]) / float(tt[0]),
                    "query_seqs_per_s": args.queries * n_batches / float(tt[0]), "batches": n_batches, "scaling": "strong",
                    "steps_in_flight": len(sp),
                    "shard_build_s_rank0": build_s, "shard_keys_rank0": sst["n_keys"],
                    "workload": "one common batch of %d %s searched by all %d ranks, each against its hash-prefix shard"
                                % (args.queries, "reads" if nucl else "protein queries", world)})
        sp.close()
        six.close()
        return rep

    # work of the timed region = sum over the launches of their batch's exact counters
    n_timed = n_launch - first_timed
    times_run = [len(range((b - first_timed) % n_distinct, n_timed, n_distinct)) for b in range(n_distinct)]
    keys = per_batch[0].keys()
    tot = {k: sum(per_batch[b][k] * times_run[b] for b in range(n_distinct)) for k in keys}
    avg = {k: tot[k] / n_timed for k in keys}   # per launch

    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    lk = torch.tensor([float(tot["n_lookup"]), float(args.queries) * n_timed / (world if sharded_mode else 1)],
                      dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(lk, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    lookups_total = float(lk[0].item())
    queries_total = float(lk[1].item())

    # ---- roofline, HBM bound (DESIGN.md "Measurement"); per BATCH, from exact kernel-side counters:
    #   probe_kernel : 1 B per residue position + 1 bit per position (k-mer-start bitmap)
    #                  + 64 B per bucket inspected + 4 B per position (the val it writes)
    #   count kernels: 4 B per position (val read back) + 4 B per arena word that must be read (list header + ids)
    #                  + 28 B per query read (descriptor 16 + table offset 8 + hit count 4) + 8 B per query written
    #                  + 8 or 12 B per emitted hit (pid, Kmatch[, first position])
    #   rest of the batch: prep (8 B offsets + 40 B meta + 24 B descriptor/offset per query); nucleotide input: translation
    #                  (1 B per nucleotide + 8 B of offsets per read in; 1 B per ORF residue + 40 B of meta per ORF out) and
    #                  the ORF prep (40 B meta in, 32 B descriptor / table size / hit offset / count out per ORF)
    c = avg
    hit_b = 12 if nucl else 8
    n_pos = c["n_in"] if nucl else float(np.mean([len(q[0]) for q in batches]))  # positions kernel P walks
    probe_bytes = n_pos + n_pos / 8 + BUCKET_BYTES * c["n_probe"] + 4 * n_pos
    count_bytes = 4 * n_pos + 4 * (c["n_lists"] + c["n_list_ids"]) + 36 * c["n_queries"] + hit_b * c["n_hits"]
    if nucl:
        nt = float(np.mean([len(q[0]) for q in batches]))
        rest_bytes = nt + 8 * args.queries + c["n_in"] + 112 * c["n_queries"]
    else:
        rest_bytes = 9 * args.queries + 72 * c["n_queries"]
    probe_s = tm["probe_ms"] / n_calls / 1e3
    count_s = tm["count_ms"] / n_calls / 1e3
    step_s = elapsed / n_timed  # per batch, wall clock of the timed region
    # HBM traffic from the PMC counters is collected in separate rocprofv3 passes (profiles/); reported here only
    # when this run is the workload those passes measured
    traffic = None
    requests = None   # 64-byte fabric requests per batch, from the same PMC passes (FETCH_SIZE + WRITE_SIZE count them)
    for rnd in ("r04", "r03", "r02"):
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", rnd + "_pmc_traffic.json")))
        except Exception:
            continue
        for e in pm["runs"]:
            if (e["db_proteins"], e["queries"], e["workload"], e["db"]) == (args.db_proteins, args.queries, args.workload, args.db) \
                    and not (sharded_mode or args.post or args.compact):
                traffic = e["traffic_bytes_per_batch"]
                d_ = pm.get("detail", {}).get(e["workload"], {})
                if "FETCH_bytes_per_batch_raw" in d_:
                    requests = (d_["FETCH_bytes_per_batch_raw"] + d_["WRITE_bytes_per_batch_raw"]) / 64.0
        if traffic is not None:
            break
    whole_bytes = probe_bytes + count_bytes + rest_bytes

    def kern(name, secs, nbytes, extra=None):
        d = {"name": name, "ms": secs * 1e3, "algorithmic_bytes_per_launch": nbytes,
             "achieved": nbytes / secs / 1e9 if secs > 0 else 0.0,
             "frac": nbytes / secs / 1e9 / HBM_PEAK_GBPS if secs > 0 else 0.0,
             "share_of_batch_time": secs / step_s if step_s > 0 else 0.0}
        d.update(extra or {})
        return d
    def alone(which, nbytes):
        if not tm_alone or not tm_alone["calls"]:
            return {}
        secs = tm_alone[which] / tm_alone["calls"] / 1e3
        d = {"ms": secs * 1e3, "achieved": nbytes / secs / 1e9, "frac": nbytes / secs / 1e9 / HBM_PEAK_GBPS, "launches": tm_alone["calls"]}
        if which == "probe_ms":
            d["G_random_requests_per_s"] = c["n_probe"] / secs / 1e9
        return {"alone_on_the_device": d}
    kernels = [kern("probe_kernel", probe_s, probe_bytes,
                    dict({"G_random_requests_per_s": c["n_probe"] / probe_s / 1e9 if probe_s > 0 else 0.0,
                          "random_request_ceiling_G_per_s": 52.0},  # tools/random_read_bench.hip: 16..128-byte records alike
                         **alone("probe_ms", probe_bytes))),
               kern(("count_async_kernel" if (os.environ.get("KAAMER_COUNT_ASYNC", "0") not in ("", "0") and not nucl) else
                     "count_pack_kernel" if nucl else "count_group_kernel") + " (+ G tier, finalize)" +
                    ("; one workgroup per CU in the timed region, three in alone_on_the_device" if (args.inflight > 1 and not nucl and not sharded_mode) else ""),
                    count_s, count_bytes, alone("count_ms", count_bytes))]
    kernels.sort(key=lambda k: -k["ms"])
    roofline = {
        "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBPS,
        # the WHOLE batch: all algorithmic bytes of a batch / wall time per batch of the timed region
        "achieved": whole_bytes / step_s / 1e9, "frac": whole_bytes / step_s / 1e9 / HBM_PEAK_GBPS,
        "traffic": traffic,
        "scope": "whole batch (prep [+ translation] + probe + count + G tier), wall clock of the timed region; per-kernel "
                 "ms are HIP-event durations inside the timed region" + (" (%d batches in flight: kernels of neighbouring "
                 "batches overlap, `alone_on_the_device` is the same kernel with one batch in flight)" % args.inflight
                                                                        if args.inflight > 1 else ""),
        "algorithmic_bytes_per_batch": whole_bytes,
        "bytes_per_lookup": whole_bytes / max(c["n_lookup"], 1),
        "dominant_kernel": kernels[0], "other_kernels": kernels[1:], "timed_launches": n_calls,
        "hip_event_batch_ms": tm["total_ms"] / n_calls,
        "min_bytes_8B_slot": n_pos + n_pos / 8 + 8 * c["n_lookup"] + 4 * n_pos + count_bytes,
        # the roofline that binds a path of random 64-byte sectors: requests per second against the measured ceiling of
        # the memory system (tools/random_read_bench.hip: ~52 G requests/s whatever the record size up to 128 B)
        "random_request_ceiling_G_per_s": 52.0,
        "requests_per_batch": requests,
        "requests_per_lookup": (requests / max(c["n_lookup"], 1)) if requests else None,
        "frac_of_request_ceiling": (requests / step_s / 52e9) if requests else None,
    }

    cfg_workload = {"protein": "configs[1]: batches of %d protein queries per GPU vs ",
                    "reads": "configs[2]: 6-frame path, batches of %d synthetic 150-nt reads per GPU vs ",
                    "mix": "configs[4] data (Q-mix: 100/150/250-nt + 5%% long reads), batches of %d reads per GPU vs "}[args.workload] % args.queries
    out = {
        "metric": "k-mer lookups/sec", "value": lookups_total / elapsed,
        "unit": "k-mer lookups/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if sharded_mode else "weak",
        "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": cfg_workload + ({"sp": "Swiss-Prot-sized synthetic DB", "ur-lite": "DB-SP generator scaled up (ur-lite) synthetic DB"}.get(args.db, "Zipf-motif synthetic DB")) +
                               " (%d proteins, %d residues, %d distinct 7-mers, longest postings list %d) resident in HBM"
                               % (args.db_proteins, int(db[1][-1]), st["n_keys"], st["max_list"]),
                   "queries_per_batch": args.queries, "batches_per_step": args.batches_per_step, "distinct_batches": n_distinct,
                   "ms_per_batch": step_s * 1e3,
                   "parallelism": ("hash-prefix shards x%d, one exchange of partial hit lists per batch" % world) if sharded_mode
                                  else ("replicas x%d (no collective)" % world if world > 1 else "single GPU"),
                   "result": "device-resident per-query hit lists (offset, count, protein ids, Kmatch)" +
                             (", packed in query order" if args.compact else ""),
                   "batches_in_flight": args.inflight, "device_post_steps": bool(args.post),
                   "table_bytes": table_bytes, "seed": workload.SEED},
        "query_seqs_per_s": queries_total / elapsed,
        "counters_per_batch_rank0": {k: round(v, 1) for k, v in c.items()},
        "roofline": roofline,
    }
    if exch is not None:
        out["exchange"] = exch
    # ---- the secondary sharded leg, after the headline is complete: whatever happens in it -- an exception on one rank, a
    # collective that never returns -- the headline line is printed.  A watchdog per rank ends the process when the leg
    # has not come back in time (a hung collective cannot be interrupted from Python); rank 0 prints the line first.
    if (world > 1 or args.sharded_leg >= 2) and not sharded_mode and args.sharded_leg:
        import threading
        leg_done = threading.Event()

        def leg_watchdog():
            if leg_done.wait(240.0):
                return
            log("sharded leg: no answer after 240 s, giving up on it")
            if rank == 0:
                out["sharded_leg"] = {"error": "timeout: the leg did not return within 240 s"}
                emit(json.dumps(out))
            os._exit(3)   # a collective never returned: the launcher must see a failure
        threading.Thread(target=leg_watchdog, daemon=True).start()
        try:
            leg = sharded_leg()
        except Exception as e:   # the headline line must survive a failure of the secondary leg
            leg = {"error": repr(e)}
        leg_done.set()
        log("sharded leg: %s" % json.dumps(leg))
        out["sharded_leg"] = leg

    if rank == 0:
        want_cpu = not args.no_cpu_baseline and world == 1 and not sharded_mode
        if args.check or want_cpu or args.host_api:
            from oracle import oracle as O  # the checker / the reported CPU baseline, never the product
            t0 = time.time()
            oix = O.Index.from_proteins(None, packed=db)
            log("oracle index built in %.1fs" % (time.time() - t0))
            q = batches[last_batch]
            if args.check:
                # the device-resident result of the last TIMED batch, read back as it lies in HBM
                from kaamer_amd.sharded import dev_tensor
                sub = workload.unpack((q[0], q[1][:args.check + 1]))
                counters = per_batch[last_batch]
                stride = world if sharded_mode else 1   # sharded: result i of rank 0 is query i * world of the batch
                nq_dev = (int(counters["n_queries"]) + stride - 1) // stride
                cap = int(last.hit_capacity)
                r_off = dev_tensor(last.d_hit_off, nq_dev, torch.int64).cpu().numpy()
                r_cnt = dev_tensor(last.d_hit_cnt, nq_dev, torch.int32).cpu().numpy()
                r_pid = dev_tensor(last.d_hit_pid, cap, torch.int32).cpu().numpy().view(np.uint32)
                r_km = dev_tensor(last.d_hit_kmatch, cap, torch.int32).cpu().numpy()

                def hits(i):
                    a = int(r_off[i])
                    return dict(zip(r_pid[a:a + int(r_cnt[i])].tolist(), r_km[a:a + int(r_cnt[i])].tolist()))
                if nucl:
                    qi = 0
                    for s in sub:
                        for o in O.get_orfs(s):
                            if qi % stride == 0:
                                pid, km, _ = oix.search(o["seq"])
                                assert hits(qi // stride) == dict(zip(pid.tolist(), km.tolist())), "bench: ORF %d differs from the oracle" % qi
                            qi += 1
                    assert (qi + stride - 1) // stride <= nq_dev
                else:
                    for i, s in enumerate(sub):
                        if i % stride:
                            continue
                        exp = {}
                        if O.size_in_kmer(s) >= 7:
                            pid, km, _ = oix.search(s)
                            exp = dict(zip(pid.tolist(), km.tolist()))
                        assert hits(i // stride) == exp, "bench: query %d differs from the oracle" % i
                out["parity_checked_queries"] = len(sub)
                log("parity: %d queries of the last timed batch bit-exact vs oracle" % len(sub))
            if args.host_api:
                # PCIe-inclusive rates of the C boundary a Go host calls (never `value`): the raw entry points through
                # ctypes (the GIL is released during a call), results freed at once
                import ctypes as C
                import threading
                L = abi.lib()
                to = abi.TopnOpts(0.05, 10, 10, 0, None, None, 0, 0)
                ins = []
                for qb in batches:
                    bb = np.ascontiguousarray(qb[0]); oo = np.ascontiguousarray(qb[1])
                    ins.append((bb, oo, abi.BatchIn(bb.ctypes.data, oo.ctypes.data, len(oo) - 1, seq_type, 0)))

                def one_call(i):
                    out = C.POINTER(abi.BatchTop)()
                    abi.check(L.kaamer_search_batch_top(ix._h, C.byref(ins[i % len(ins)][2]), C.byref(to), C.byref(out)))
                    L.kaamer_batch_top_free(out)
                hb = {}
                lk_b = float(np.mean([pb["n_lookup"] for pb in per_batch]))
                n_seq_calls = 4 if nucl else 40
                for i in range(4):
                    one_call(i)
                t0 = time.perf_counter()
                for i in range(n_seq_calls):
                    one_call(i)
                dt = (time.perf_counter() - t0) / n_seq_calls
                hb["search_batch_top_one_caller"] = {"ms_per_call": dt * 1e3, "lookups_per_s": lk_b / dt}
                for n_thr in (2, 4, 8):
                    per = max(2, n_seq_calls // 2)

                    def worker(k):
                        for i in range(per):
                            one_call(k * per + i)
                    # one untimed round first: a slot creates its workspace, device staging and pinned buffers at its FIRST
                    # use (hundreds of ms for a 1 M-read chunk).  Round 3 timed that inside the 4-caller case -- the first to
                    # touch slots 2 and 3 -- and reported 120 ms per call where warmed slots take 12
                    warm = [threading.Thread(target=one_call, args=(k,)) for k in range(n_thr)]
                    for t_ in warm:
                        t_.start()
                    for t_ in warm:
                        t_.join()
                    th = [threading.Thread(target=worker, args=(k,)) for k in range(n_thr)]
                    t0 = time.perf_counter()
                    for t_ in th:
                        t_.start()
                    for t_ in th:
                        t_.join()
                    dt = (time.perf_counter() - t0) / (n_thr * per)
                    hb["search_batch_top_%d_callers" % n_thr] = {"ms_per_call": dt * 1e3, "lookups_per_s": lk_b / dt,
                                                                 "seqs_per_s": args.queries / dt}
                # streaming (kaamer_stream_push / _pop): chunks of one batch each, three in flight
                st_h = C.c_void_p()
                abi.check(L.kaamer_stream_open(ix._h, seq_type, C.byref(to), C.byref(st_h)))
                n_chunks = 8 if nucl else 64
                fifo = 0

                def pop_one():
                    o_ = C.POINTER(abi.BatchTop)()
                    abi.check(L.kaamer_stream_pop(st_h, C.byref(o_)))
                    L.kaamer_batch_top_free(o_)
                t0 = time.perf_counter()
                for i in range(n_chunks):
                    bb, oo, _ = ins[i % len(ins)]
                    while True:
                        if fifo < 3:
                            rc = L.kaamer_stream_push(st_h, bb.ctypes.data, oo.ctypes.data, len(oo) - 1)
                            if rc == 0:
                                fifo += 1
                                break
                            if rc != abi.E_BUSY:
                                abi.check(rc)
                        pop_one()
                        fifo -= 1
                while fifo:
                    pop_one()
                    fifo -= 1
                dt = (time.perf_counter() - t0) / n_chunks
                L.kaamer_stream_close(st_h)
                hb["stream_push_pop_3_in_flight"] = {"ms_per_chunk": dt * 1e3, "lookups_per_s": lk_b / dt, "seqs_per_s": args.queries / dt}
                ix.search(packed=q, seq_type=seq_type)
                t0 = time.perf_counter()
                ix.search(packed=q, seq_type=seq_type)
                ix.search(packed=q, seq_type=seq_type)
                hb["search_batch_full_hit_lists"] = {"ms_per_call": (time.perf_counter() - t0) / 2 * 1e3}

                def full_call(i):   # raw entry point: the result arrays are freed unread
                    o_ = C.POINTER(abi.BatchOut)()
                    abi.check(L.kaamer_search_batch(ix._h, C.byref(ins[i % len(ins)][2]), C.byref(o_)))
                    L.kaamer_batch_free(o_)
                per = 2 if nucl else 12

                def full_worker(k):
                    for i in range(per):
                        full_call(k * per + i)
                for rep in range(2):   # the first round gives every slot its workspace
                    th = [threading.Thread(target=full_worker, args=(k,)) for k in range(4)]
                    t0 = time.perf_counter()
                    for t_ in th:
                        t_.start()
                    for t_ in th:
                        t_.join()
                dt = (time.perf_counter() - t0) / (4 * per)
                hb["search_batch_full_hit_lists_4_callers"] = {"ms_per_call": dt * 1e3, "lookups_per_s": lk_b / dt}
                out["host_buffer_calls_pcie_inclusive"] = hb
            if want_cpu:
                out["cpu_baseline"] = cpu_baseline(oix, q, seconds=args.cpu_seconds, kind="reads" if nucl else "protein")
        emit(json.dumps(out))
    if world > 1 or sharded_mode:
        # (a rank that left the sharded leg on an error never reaches this barrier together with the others: the line is
        # out by now, so a barrier that does not come back within a minute just ends the process)
        import threading
        bye = threading.Timer(60.0, lambda: os._exit(3))   # the final barrier hung: not a success
        bye.daemon = True
        bye.start()
        dist.barrier()
        dist.destroy_process_group()
        bye.cancel()


if __name__ == "__main__":
    main()
