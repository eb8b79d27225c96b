"""numpy restatement of the sharded exchange's fixed-size block format (kaamer_amd/csrc/exchange.hip.inc), test
infrastructure only.  Rank r sends one block of `block_words` u32 words to every rank d:

    [0] entries in the block   [1] status (bit 0: capacity exceeded, bit 1: first positions inside, bit 2: sender failed,
                                   bit 3: some block of this sender exceeded its capacity)
    [2] queries of the batch   [3] queries owned by d = ceil((nq - d) / W)
    [4] entries the block needed   [5] the largest [4] over the sender's W blocks   [6], [7] zero
    [8 .. 8+q_cap)             per owned query i (global query d + i W): number of entries
    then pid[e_cap], kmatch[e_cap], first_pos[e_cap]

`defined_words` lists the words of a block the format defines (the device leaves the rest of a block untouched), so a
device block and a numpy block can be compared bit for bit."""
import numpy as np

X_HDR = 8


def pack_blocks(layout, nq, hit_off, hit_cnt, pid, km, fp, with_fp, src_status=0):
    """partial hit lists of ALL nq queries of one rank -> world blocks (one flat uint32 array)"""
    W, q_cap, e_cap, bw = int(layout.world), int(layout.q_cap), int(layout.e_cap), int(layout.block_words)
    out = np.zeros(W * bw, np.uint32)
    for d in range(W):
        blk = out[d * bw:(d + 1) * bw]
        owned = list(range(d, nq, W))
        st = 0
        if len(owned) > q_cap:
            owned, st = owned[:q_cap], 1
        cnt = np.array([int(hit_cnt[q]) for q in owned], np.int64)
        total = int(cnt.sum())
        blk[X_HDR:X_HDR + len(owned)] = cnt
        if total > e_cap:
            st, total_w = 1, 0
        else:
            total_w = total
            o = 0
            ent = blk[X_HDR + q_cap:]
            for q, n in zip(owned, cnt):
                s = int(hit_off[q])
                ent[o:o + n] = pid[s:s + n]
                ent[e_cap + o:e_cap + o + n] = km[s:s + n]
                if with_fp:
                    ent[2 * e_cap + o:2 * e_cap + o + n] = fp[s:s + n]
                o += int(n)
        blk[0] = total_w
        blk[1] = st | (2 if with_fp else 0) | (4 if src_status else 0)
        blk[2] = nq
        blk[3] = len(owned)
        blk[4] = min(total, 0xFFFFFFFF)
    need = max(int(out[d * bw + 4]) for d in range(W))
    anyov = any(int(out[d * bw + 1]) & 1 for d in range(W))
    for d in range(W):
        out[d * bw + 5] = need
        if anyov:
            out[d * bw + 1] |= 8
    return out


def defined_words(layout, blocks, with_fp):
    """boolean mask over `blocks` (world blocks) of the words the format defines"""
    W, q_cap, e_cap, bw = int(layout.world), int(layout.q_cap), int(layout.e_cap), int(layout.block_words)
    m = np.zeros(W * bw, bool)
    for d in range(W):
        blk = blocks[d * bw:(d + 1) * bw]
        n_ent, n_owned = int(blk[0]), min(int(blk[3]), q_cap)
        b = d * bw
        m[b:b + X_HDR + n_owned] = True
        e0 = b + X_HDR + q_cap
        for a in range(3 if with_fp else 2):
            m[e0 + a * e_cap:e0 + a * e_cap + n_ent] = True
    return m


def unpack_merge(layout, recv, with_fp):
    """received blocks (block s = what rank s packed for this rank) -> per owned query {pid: (Kmatch sum, lowest first
    position)}; raises ValueError where the device reports an error (overflowed block, sender failed, missing first
    positions, headers that do not describe one batch)"""
    W, rank, q_cap, e_cap, bw = int(layout.world), int(layout.rank), int(layout.q_cap), int(layout.e_cap), int(layout.block_words)
    blks = [recv[s * bw:(s + 1) * bw] for s in range(W)]
    nq = int(blks[0][2])
    want = len(range(rank, nq, W))
    for b in blks:
        if int(b[2]) != nq or int(b[3]) != want or want > q_cap:
            raise ValueError("headers disagree")
        if int(b[1]) & 9:
            raise ValueError("block overflow")
        if int(b[1]) & 4:
            raise ValueError("peer failed")
        if with_fp and not int(b[1]) & 2:
            raise ValueError("first positions missing")
    out = [dict() for _ in range(want)]
    for b in blks:
        cnt = b[X_HDR:X_HDR + want].astype(np.int64)
        ent = b[X_HDR + q_cap:]
        o = 0
        for i, n in enumerate(cnt):
            for t in range(o, o + int(n)):
                p, k = int(ent[t]), int(ent[e_cap + t])
                f = int(ent[2 * e_cap + t]) if with_fp else 0
                a = out[i].get(p, (0, 1 << 32))
                out[i][p] = (a[0] + k, min(a[1], f))
            o += int(n)
    return out
