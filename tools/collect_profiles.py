#!/usr/bin/env python3
"""gpurun_out/<round>/ (written by tools/profile_round.sh on the GPU box) -> profiles/<round>_*.

HBM traffic per batch comes from the FETCH_SIZE / WRITE_SIZE passes (KiB, separate rocprofv3 --pmc passes with
--kernel-trace only), summed over EVERY kernel of the batch, with the corrections MI355X_MICROARCH.md prescribes:
random 64-B sector reads are counted exactly (checked by the calibration pass on tools/random_read_bench.hip),
wide coalesced streaming reads at half -- so half of the batch's streamed input (residues, bitmap, vals) is added
back.  bench.py reads the result from profiles/<round>_pmc_traffic.json for its roofline.traffic field."""
import csv
import glob
import json
import os
import re
import shutil
import sys

R = sys.argv[1] if len(sys.argv) > 1 else "r03"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", R)
DST = os.path.join(ROOT, "profiles")


def one(pattern):
    fs = sorted(glob.glob(os.path.join(SRC, pattern)), key=os.path.getmtime)  # gpurun merges runs: take the newest
    if not fs:
        raise SystemExit("missing " + pattern)
    return fs[-1]


def counter_sums(d, counter):
    """kernel name -> (sum over dispatches, dispatches)"""
    acc = {}
    for r in csv.DictReader(open(one(d + "/*/*_counter_collection.csv"))):
        # (the table is built on the device once per process -- builder_device.hip's kernels and rocPRIM's: not part of a batch)
        # nor are the fills: the steady state of a workspace has none, they initialise the build's tables and the workspaces
        if "bd_" in r["Kernel_Name"] or "rocprim" in r["Kernel_Name"] or "fillBuffer" in r["Kernel_Name"]:
            continue
        if r["Counter_Name"] == counter:
            a = acc.setdefault(r["Kernel_Name"], [0.0, 0])
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    return acc


def last_json_line(path):
    return json.loads([l for l in open(path) if l.startswith("{")][-1])


def short(k):
    return re.sub(r"\(.*", "", k.replace("void ", ""))[:44]


os.makedirs(DST, exist_ok=True)
for f in ("bench_default_n1", "bench_under_rocprof", "bench_inflight1_under_rocprof", "bench_reads_n1", "bench_reads_under_rocprof", "bench_reads_inflight1_under_rocprof", "bench_zipf", "bench_zipf_inflight1", "bench_reads_inflight3",
          "bench_zipf_mid", "bench_mix", "bench_sharded_w1", "bench_sharded_reads_w1", "bench_sharded_default_w1", "bench_post_hostapi", "bench_reads_post_hostapi",
          "bench_default_with_sharded_leg_w1"):
    if os.path.exists(os.path.join(SRC, f + ".json")):
        shutil.copy(os.path.join(SRC, f + ".json"), os.path.join(DST, "%s_%s.json" % (R, f)))
for f in ("random_read_bench.txt", "bucket_read_bench.txt", "sq_protein.txt", "sq_reads.txt"):
    if os.path.exists(os.path.join(SRC, f)):
        shutil.copy(os.path.join(SRC, f), os.path.join(DST, "%s_%s" % (R, f)))
for d, name in (("stats_protein", "_kernel_stats_protein_config1.csv"), ("stats_protein_if1", "_kernel_stats_protein_config1_one_in_flight.csv"),
                ("stats_reads", "_kernel_stats_reads_config2.csv"), ("stats_reads_if1", "_kernel_stats_reads_config2_one_in_flight.csv"), ("stats_sharded", "_kernel_stats_sharded_w1_protein.csv"),
                ("stats_sharded_reads", "_kernel_stats_sharded_w1_reads.csv")):
    fs = sorted(glob.glob(os.path.join(SRC, d, "*", "*_kernel_stats.csv")) + glob.glob(os.path.join(SRC, d, "*_kernel_stats.csv")), key=os.path.getmtime)
    if fs:
        shutil.copy(fs[-1], os.path.join(DST, R + name))

runs = []
detail = {}
for tag, fd, wd in (("protein", "pmc_fetch", "pmc_write"), ("reads", "pmc_fetch_reads", "pmc_write_reads")):
    if not os.path.exists(os.path.join(SRC, fd + ".json")) or not os.path.exists(os.path.join(SRC, wd + ".json")):
        continue
    b = last_json_line(os.path.join(SRC, fd + ".json"))
    cfg, c = b["config"], b["counters_per_batch_rank0"]
    fetch, write = counter_sums(fd, "FETCH_SIZE"), counter_sums(wd, "WRITE_SIZE")
    # every batch launches each search kernel once: dispatches of the probe kernel = batches profiled
    probe = [k for k in fetch if "probe_kernel" in k][0]
    n_prof = fetch[probe][1]
    f_b = sum(v[0] for v in fetch.values()) * 1024 / n_prof
    w_b = sum(v[0] for v in write.values()) * 1024 / n_prof
    streamed = c["n_in"] * (1 + 1 / 8) + 4 * c["n_in"]      # residues + bitmap (probe), vals (count): coalesced reads
    traffic = f_b + w_b + streamed / 2
    runs.append({"workload": tag, "db": "sp", "db_proteins": 560000, "queries": cfg["queries_per_batch"],
                 "traffic_bytes_per_batch": int(traffic)})
    detail[tag] = {
        "batches_profiled": n_prof, "FETCH_bytes_per_batch_raw": int(f_b), "WRITE_bytes_per_batch_raw": int(w_b),
        "streamed_input_bytes_per_batch": int(streamed), "traffic_bytes_per_batch": int(traffic),
        "algorithmic_bytes_per_batch": b["roofline"]["algorithmic_bytes_per_batch"],
        "traffic_over_algorithmic": round(traffic / b["roofline"]["algorithmic_bytes_per_batch"], 4),
        "per_kernel_KiB_per_batch": {"FETCH_SIZE": {short(k): round(v[0] / n_prof, 1) for k, v in fetch.items()},
                                     "WRITE_SIZE": {short(k): round(v[0] / n_prof, 1) for k, v in write.items()}},
        "counters_per_batch": c}
    # requests the memory system saw: FETCH_SIZE / WRITE_SIZE count 64-byte fabric requests (MI355X_MICROARCH.md)
    detail[tag]["requests_per_batch"] = int((f_b + w_b) / 64)
    detail[tag]["requests_per_lookup"] = round((f_b + w_b) / 64 / max(c["n_lookup"], 1), 4)
have_cal = os.path.isdir(os.path.join(SRC, "pmc_cal")) and os.path.exists(os.path.join(SRC, "random_read_bench.txt"))
cal = counter_sums("pmc_cal", "FETCH_SIZE") if have_cal else {}
txt = open(os.path.join(SRC, "random_read_bench.txt")).read() if have_cal else ""
ceiling = {m.group(1) + "B": float(m.group(2)) for m in re.finditer(r"^record\s+(\d+) B:.*?([\d.]+) G records/s", txt, re.M)}
out = {
    "what": "HBM traffic of one batch of the hot path (all kernels) from rocprofv3 PMC: separate --pmc FETCH_SIZE and --pmc WRITE_SIZE "
            "passes, --kernel-trace only; workloads = bench.py configs[1] (protein) and configs[2] (reads) on DB-SP",
    "produced_by": "tools/profile_round.sh + tools/collect_profiles.py",
    "units": "FETCH_SIZE / WRITE_SIZE are KiB; correction: random 64-B sector reads are counted exactly (calibration below), "
             "coalesced streaming reads at half, so half of the streamed input is added back",
    "runs": runs,
    "detail": detail,
    "calibration_FETCH_SIZE_KiB_per_dispatch_of_random_read_bench": {short(k): round(v[0] / v[1], 1) for k, v in cal.items()},
    "random_request_ceiling_G_records_per_s": ceiling,
}
json.dump(out, open(os.path.join(DST, R + "_pmc_traffic.json"), "w"), indent=1)
for t, d in detail.items():
    print("%s: traffic/algorithmic = %.3f" % (t, d["traffic_over_algorithmic"]))
