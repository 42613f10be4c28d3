"""Level-1 sampling, phase A of a round lasts as long as its busiest wave: how many bucket updates does the busiest wave get under the kernel's
bucket -> wave map (bucket b in wave b % 16) and under alternatives?  CPU simulation: plain FPS on the bench's clouds, 64-point buckets of the
kernel's Z-order sort, a sample "reaches" a bucket when the kernel's box test would (box distance < the bucket's largest running distance),
rounds of 6 consecutive samples (the kernel certifies 5.6 per round on hdl64).    python scripts/r06/sim_fps_ownership.py"""
import sys
import numpy as np
sys.path.insert(0, '/root/repo')
from ws3d_amd import synth


def morton(cx, cz, bits=6):
    code = np.zeros_like(cx)
    for i in range(bits):
        code |= ((cx >> i) & 1) << (2 * i) | ((cz >> i) & 1) << (2 * i + 1)
    return code


def run(kind, seed, M=4096, K=6):
    xyz = synth.cloud(kind, 16384, seed)[:, :3].astype(np.float32)
    n = len(xyz)
    x, z = xyz[:, 0], xyz[:, 2]
    cx = np.clip(((x - x.min()) * 64 / (x.max() - x.min())).astype(np.int64), 0, 63)
    cz = np.clip(((z - z.min()) * 64 / (z.max() - z.min())).astype(np.int64), 0, 63)
    order = np.argsort(morton(cx, cz), kind='stable')
    P = xyz[order].reshape(256, 64, 3)
    lo, hi = P.min(1), P.max(1)
    t = np.full((256, 64), 1e10, np.float32)
    cur = P[0, 0] * 0 + xyz[0]
    touched = []                                  # per sample: the buckets its box test reaches
    for j in range(1, M):
        bmax = t.max(1)
        g = np.maximum(np.maximum(lo - cur, cur - hi), 0)
        L = (g * g).sum(1)
        reach = np.nonzero(L < bmax)[0]
        touched.append(reach)
        d = ((P[reach] - cur) ** 2).sum(2)
        t[reach] = np.minimum(t[reach], d)
        fl = t.argmax()
        cur = P[fl // 64, fl % 64]
    maps = {
        "b % 16 (the kernel)": lambda b: b % 16,
        "(b ^ (b >> 4)) % 16": lambda b: (b ^ (b >> 4)) % 16,
        "(b ^ (b >> 4) ^ (b >> 8)) % 16": lambda b: (b ^ (b >> 4) ^ (b >> 8)) % 16,
        "(b * 7 + (b >> 4) * 5) % 16": lambda b: (b * 7 + (b >> 4) * 5) % 16,
        "bitrev-ish ((b >> 2) ^ b) % 16": lambda b: ((b >> 2) ^ b) % 16,
    }
    out = []
    for name, f in maps.items():
        mx, mean, tot = [], [], []
        for r0 in range(0, len(touched), K):
            bs = np.unique(np.concatenate(touched[r0:r0 + K])) if len(touched[r0:r0 + K]) else np.array([], int)
            cnt = np.bincount(f(bs), minlength=16)
            mx.append(cnt.max()); mean.append(cnt.mean()); tot.append(len(bs))
        out.append((name, float(np.mean(mx)), float(np.mean(mean)), float(np.mean(tot))))
    return out


for kind in ("hdl64", "lidar"):
    acc = {}
    for seed in (2000, 2001, 2002):
        for name, mx, mean, tot in run(kind, seed):
            acc.setdefault(name, []).append((mx, mean, tot))
    for name, v in acc.items():
        v = np.array(v).mean(0)
        print("%-6s %-34s busiest wave %.2f updates per round, average wave %.2f, %.1f buckets per round" % (kind, name, v[0], v[1], v[2]), flush=True)
