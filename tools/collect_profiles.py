#!/usr/bin/env python3
"""gpurun_out/<round>/ (written by tools/profile_round.sh on the GPU box) -> profiles/<round>_*.
Computes the HBM traffic of probe_kernel per launch from the FETCH_SIZE / WRITE_SIZE passes with the
corrections MI355X_MICROARCH.md prescribes: the counters are KiB; random 64-B sector reads are counted
exactly (checked by the calibration pass on tools/random_read_bench.hip), wide streaming reads at half."""
import csv
import glob
import json
import os
import re
import shutil
import sys

R = sys.argv[1] if len(sys.argv) > 1 else "r01"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", R)
DST = os.path.join(ROOT, "profiles")


def one(pattern):
    fs = sorted(glob.glob(os.path.join(SRC, pattern)), key=os.path.getmtime)  # gpurun merges runs: take the newest
    if not fs:
        raise SystemExit("missing " + pattern)
    return fs[-1]


def counter_means(d, counter):
    acc = {}
    for r in csv.DictReader(open(one(d + "/*/*_counter_collection.csv"))):
        if r["Counter_Name"] == counter:
            acc.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def last_json_line(path):
    return json.loads([l for l in open(path) if l.startswith("{")][-1])


os.makedirs(DST, exist_ok=True)
for src, dst in (("bench_default_n1.json", "bench_default_n1.json"), ("bench_under_rocprof.json", "bench_under_rocprof.json"),
                 ("bench_reads_n1.json", "bench_reads_n1.json"), ("bench_reads_under_rocprof.json", "bench_reads_under_rocprof.json"),
                 ("bench_compact1.json", "bench_variant_compact1.json"), ("bench_post_hostapi.json", "bench_variant_post_hostapi.json"),
                 ("bench_inflight3.json", "bench_variant_inflight3.json"), ("random_read_bench.txt", "random_read_bench.txt")):
    shutil.copy(os.path.join(SRC, src), os.path.join(DST, "%s_%s" % (R, dst)))
shutil.copy(one("stats_protein/*/*_kernel_stats.csv"), os.path.join(DST, R + "_kernel_stats_protein_config1.csv"))
shutil.copy(one("stats_reads/*/*_kernel_stats.csv"), os.path.join(DST, R + "_kernel_stats_reads_config2.csv"))

fetch, write = counter_means("pmc_fetch", "FETCH_SIZE"), counter_means("pmc_write", "WRITE_SIZE")
b = last_json_line(os.path.join(SRC, "pmc_fetch.json"))
c = b["counters_per_step_rank0"]
n_pos = (b["roofline"]["algorithmic_bytes_per_launch"] - 64 * c["n_probe"]) * 8 // 41  # bytes = n_pos*(1 + 1/8 + 4) + 64*n_probe
probe = [k for k in fetch if k.startswith("probe_kernel")][0]
streaming = n_pos + n_pos // 8
traffic = fetch[probe] * 1024 + streaming / 2 + write[probe] * 1024
cal = counter_means("pmc_cal", "FETCH_SIZE")
txt = open(os.path.join(SRC, "random_read_bench.txt")).read()
ceiling = {m.group(1) + "B": float(m.group(2)) for m in re.finditer(r"^record\s+(\d+) B:.*?([\d.]+) G records/s", txt, re.M)}
sq = {}
for name in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS",
             "SQ_LDS_BANK_CONFLICT"):
    for k, v in counter_means("pmc_sq", name).items():
        if k.startswith("probe_kernel") or "count_group_kernel" in k:
            sq.setdefault(k[:40], {})[name] = v
out = {
    "what": "HBM traffic of probe_kernel per launch from rocprofv3 PMC (separate --pmc FETCH_SIZE and --pmc WRITE_SIZE passes, "
            "--kernel-trace only), workload = bench.py defaults (configs[1])",
    "produced_by": "tools/profile_round.sh + tools/collect_profiles.py",
    "units": "FETCH_SIZE / WRITE_SIZE are KiB",
    "config": {"db_proteins": 560000, "queries": 10000, "workload": "protein"},
    "probe_kernel": {"FETCH_SIZE_KiB": fetch[probe], "WRITE_SIZE_KiB": write[probe],
                     "streaming_read_bytes_algorithmic": streaming,
                     "correction": "random 64-B sector reads are counted exactly (calibration below); the streaming reads "
                                   "(residues + bitmap) are counted at half, so half of them is added back",
                     "traffic_bytes_per_launch": int(traffic),
                     "algorithmic_bytes_per_launch": b["roofline"]["algorithmic_bytes_per_launch"]},
    "all_kernels_KiB": {"FETCH_SIZE": {k[:48]: v for k, v in fetch.items()}, "WRITE_SIZE": {k[:48]: v for k, v in write.items()}},
    "calibration_FETCH_SIZE_KiB_per_kernel_of_random_read_bench": {k[:60]: v for k, v in cal.items()},
    "random_request_ceiling_G_records_per_s": ceiling,
    "sq_counters_mean_per_launch": sq,
}
json.dump(out, open(os.path.join(DST, R + "_pmc_traffic_probe_kernel.json"), "w"), indent=1)
print("traffic/algorithmic = %.4f" % (traffic / b["roofline"]["algorithmic_bytes_per_launch"]))
