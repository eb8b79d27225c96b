"""DB-UR-lite: builds ONE shard of an S-shard index over a synthetic database of `--residues` residues
through the product builder (kaamer_image_build_proteins), and reports what it took: wall time per phase
(KAAMER_BUILD_TRACE), the shard's table statistics and the process's peak RSS.  The two-pass builder keeps
O(shard) pair memory, so the peak is  database text + 2 x 8 B x windows / S + the shard image.

    python tools/build_ur_lite.py --residues 5e9 --shards 8 --shard 0 [--save /tmp/ur_s0.kgi]
"""
import argparse
import json
import os
import resource
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("KAAMER_BUILD_TRACE", "1")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--residues", type=float, default=5e9)
    ap.add_argument("--shards", type=int, default=8)
    ap.add_argument("--shard", type=int, default=0)
    ap.add_argument("--save", default=None)
    a = ap.parse_args()
    from kaamer_amd import api, build, workload
    build.build()
    n_prot = int(a.residues / 358.5)        # mean length of workload._lengths
    t = time.time()
    db = workload.make_db(n_prot)
    t_gen = time.time() - t
    rss_gen = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6
    print("generated %d proteins, %.3e residues in %.0f s (RSS %.1f GB)" % (n_prot, len(db[0]), t_gen, rss_gen), file=sys.stderr, flush=True)
    t = time.time()
    img = api.Image.from_proteins(packed=db, shard=a.shard, n_shards=a.shards)
    t_build = time.time() - t
    st = img.stats()
    out = dict(residues=int(len(db[0])), proteins=n_prot, shard=a.shard, n_shards=a.shards, build_s=round(t_build, 1),
               generate_s=round(t_gen, 1), peak_rss_gb=round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6, 2),
               rss_after_generate_gb=round(rss_gen, 2), cores=os.cpu_count(),
               image_gb=round((st["n_buckets"] * 64 + st["arena_words"] * 4) / 1e9, 3), stats=st)
    if a.save:
        t = time.time()
        img.save(a.save)
        out["save_s"] = round(time.time() - t, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
