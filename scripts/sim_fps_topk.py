"""How many samples per round would fps_rounds_kernel certify if every wave published its TOP-k points (by running distance) instead of
its best one?  CPU simulation on the bench's clouds with the kernel's bucket -> wave ownership (64-point buckets of the Z-order sort,
bucket b in wave b % 16) and its conditions: candidates taken in decreasing value; a candidate is accepted iff (a) its value is the
strict maximum of the unused candidates, (b) it exceeds the bound s_w (the best NON-published point) of every wave whose published
candidates are all used up... conservatively: of every wave that has had a candidate accepted this round -- no: of every wave, its
bound covers only its unpublished points, so (b) reads v > max_w s_w over ALL waves w (published candidates are compared directly),
(c) fl|c - q_i|^2 >= v for every sample q_i accepted before it this round.  The sequence is checked against plain FPS.
    python scripts/sim_fps_topk.py"""
import sys
import numpy as np
sys.path.insert(0, '/root/repo')
from ws3d_amd import synth
import oracle


def morton(cx, cz, bits=6):
    code = np.zeros_like(cx)
    for i in range(bits):
        code |= ((cx >> i) & 1) << (2 * i) | ((cz >> i) & 1) << (2 * i + 1)
    return code


def sim(kind, seed, topk, kmax, M=4096, NW=16, per_bucket=True):
    xyz = synth.cloud(kind, 16384, seed)[:, :3].astype(np.float32)
    n = xyz.shape[0]
    x, z = xyz[:, 0], xyz[:, 2]
    cx = np.clip(((x - x.min()) * 64 / (x.max() - x.min())).astype(np.int64), 0, 63)
    cz = np.clip(((z - z.min()) * 64 / (z.max() - z.min())).astype(np.int64), 0, 63)
    order = np.argsort(morton(cx, cz), kind='stable')
    bucket_of = np.empty(n, np.int64)
    bucket_of[order] = np.arange(n) // 64
    wave_of = bucket_of % NW
    widx = [np.nonzero(wave_of == w)[0] for w in range(NW)]
    wbkt = [bucket_of[widx[w]] for w in range(NW)]

    def d2(q):
        d = xyz - xyz[q]
        return (d[:, 2] * d[:, 2] + (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1])).astype(np.float32)
    t = np.minimum(np.full(n, 1e10, np.float32), d2(0))
    picks = [0]
    hist = np.zeros(kmax + 1, np.int64)
    rounds = 0
    stop = {"cap": 0, "bound": 0, "touched": 0, "tie": 0, "end": 0}
    while len(picks) < M:
        cands, bound = [], -1.0
        for w in range(NW):
            tv = t[widx[w]]
            if per_bucket:
                # the kernel's form: the best point of each of the wave's top-k BUCKETS; bound = everything else
                bm = {}
                for b in np.unique(wbkt[w]):
                    sel = wbkt[w] == b
                    bm[b] = tv[sel].max()
                top = sorted(bm, key=lambda b: -bm[b])[:topk]
                used = np.zeros(len(tv), bool)
                for b in top:
                    sel = np.nonzero(wbkt[w] == b)[0]
                    a = sel[np.argmax(tv[sel])]
                    cands.append((tv[a], widx[w][a]))
                    used[a] = True
                rest = tv[~used]
                bound = max(bound, rest.max() if rest.size else -1.0)
            else:
                o = np.argsort(-tv, kind='stable')
                for a in o[:topk]:
                    cands.append((tv[a], widx[w][a]))
                bound = max(bound, tv[o[topk]] if len(o) > topk else -1.0)
        cands.sort(key=lambda c: -c[0])
        acc = []
        why = "cap"
        for i, (v, p) in enumerate(cands):
            if len(acc) >= kmax or len(picks) + len(acc) >= M:
                why = "cap" if len(acc) >= kmax else "end"
                break
            if i + 1 < len(cands) and cands[i + 1][0] == v:
                why = "tie"
                break
            if not (v > bound):
                why = "bound"
                break
            ok = True
            for q in acc:
                d = xyz[p] - xyz[q]
                if not (np.float32(d[2] * d[2] + (d[0] * d[0] + d[1] * d[1])) >= v):
                    ok = False
                    break
            if not ok:
                why = "touched"
                break
            acc.append(p)
        if not acc:                       # tie at the head: the resolution round takes one sample
            acc = [cands[0][1]]
        stop[why] += 1
        for p in acc:
            picks.append(int(p))
            t = np.minimum(t, d2(p))
        hist[len(acc)] += 1
        rounds += 1
    return picks, rounds, hist, stop


if __name__ == "__main__":
    ref = {}
    for kind in ("hdl64", "lidar"):
        xyz = synth.cloud(kind, 16384, 3000)[:, :3].astype(np.float32)
        ref[kind] = oracle.furthest_point_sample(xyz[None], 4096)[0]
        for topk, kmax in ((1, 4), (2, 8), (4, 8), (4, 16), (8, 16), (16, 32)):
            p, r, h, stop = sim(kind, 3000, topk, kmax)
            same = np.array_equal(np.asarray(p), ref[kind])
            print("%s top-%d per wave, at most %2d per round: %4d rounds, %.2f samples per round, rounds ended by %s%s" %
                  (kind, topk, kmax, r, 4095 / r, stop, "" if same else "  SEQUENCE DIFFERS (ties resolved by index here)"), flush=True)
