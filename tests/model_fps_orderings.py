#!/usr/bin/env python3
"""CPU model (numpy, minutes) of how many 64-row buckets the exact bucket-pruned FPS has to stream per round under
different point orderings and pruning schemes, on one bench cloud.  Developer tool, not a test: it backs the numbers in
DESIGN.md section 5 ("orderings and triggers modelled on the CPU").  Run:  python tests/model_fps_orderings.py [random|tabletop]
"""
import sys, time, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ('', 'tests', 'tools', 'point-cloud-processing_amd'):
    sys.path.insert(0, os.path.join(ROOT, sub))
from oracle import numpy_ref as R, cbind
from pcp import synth
from gen_hilbert_lut import hilbert_index

def cloud(seed=0, scene='random'):
    if scene == 'random':
        d, c = synth.episode(1, seed0=seed)
    else:
        d, c = synth.scene_episode(1, seed0=seed)
    intr = synth.intrinsics(480, 640)
    o = R.stage1_frame(d[0], c[0], intr, intr)
    pts = o['pts_g']
    lo = np.array([-0.05, -0.12, -0.008]); hi = np.array([0.4, 0.35, 0.6])
    m = np.all((pts > lo) & (pts < hi), axis=1)
    return np.ascontiguousarray(pts[m].astype(np.float32))

def fps_samples(P, K):
    d = np.full(len(P), np.inf, np.float32); idx = [0]
    for k in range(K - 1):
        diff = P - P[idx[-1]]
        dd = diff[:, 0] * diff[:, 0] + diff[:, 1] * diff[:, 1] + diff[:, 2] * diff[:, 2]
        d = np.minimum(d, dd); idx.append(int(np.argmax(d)))
    return idx

def hilbert_keys(P, bits):
    lo, hi = P.min(0), P.max(0)
    q = np.minimum(((P - lo) / (hi - lo) * (1 << bits)).astype(np.int64), (1 << bits) - 1)
    lut = {}
    keys = np.empty(len(P), np.int64)
    code = q[:, 0] << (2 * bits) | q[:, 1] << bits | q[:, 2]
    for cval in np.unique(code):
        x, y, z = cval >> (2 * bits), (cval >> bits) & ((1 << bits) - 1), cval & ((1 << bits) - 1)
        lut[cval] = hilbert_index(int(x), int(y), int(z), bits)
    return np.array([lut[c] for c in code])

def kd_order(P, leaf=64):
    order = np.arange(len(P))
    out = []
    def rec(ix):
        if len(ix) <= leaf:
            out.append(ix); return
        ext = P[ix].max(0) - P[ix].min(0); a = int(np.argmax(ext))
        # split at a multiple of leaf closest to the middle
        nl = (len(ix) + leaf - 1) // leaf; left = (nl // 2) * leaf
        part = np.argpartition(P[ix, a], left - 1) if left < len(ix) else np.arange(len(ix))
        rec(ix[part[:left]]); rec(ix[part[left:]])
    rec(order)
    return np.concatenate(out)

def simulate(P, order, samples, BS=64, tag=''):
    S = P[order]; M = len(S); NB = (M + BS - 1) // BS
    pad = NB * BS - M
    Sp = np.concatenate([S, np.repeat(S[-1:], pad, 0)]) if pad else S
    B3 = Sp.reshape(NB, BS, 3)
    blo, bhi = B3.min(1), B3.max(1)
    md = np.full((NB, BS), np.inf, np.float32); bmax = np.full(NB, np.inf, np.float32)
    act = []
    for k, si in enumerate(samples[:-1]):
        s = P[si]
        q = np.clip(s, blo, bhi); lb = ((s - q) ** 2).sum(1)
        a = lb < bmax
        act.append(int(a.sum()))
        ids = np.nonzero(a)[0]
        dd = ((B3[ids] - s) ** 2).sum(2)
        md[ids] = np.minimum(md[ids], dd); bmax[ids] = md[ids].max(1)
    act = np.array(act)
    vol = np.prod(bhi - blo, axis=1).mean()
    print(f"{tag:28s} buckets {NB}  mean active/round: all {act.mean():6.1f}  r1-255 {act[1:256].mean():6.1f}  r256+ {act[256:].mean():6.2f}  mean box vol {vol:.3e}")

if __name__ == '__main__':
    scene = sys.argv[1] if len(sys.argv) > 1 else 'random'
    P = cloud(0, scene); print(scene, len(P), 'points')
    t = time.time(); samples = fps_samples(P, 1024); print('fps', round(time.time() - t, 1), 's')
    h4 = hilbert_keys(P, 4)
    simulate(P, np.argsort(h4, kind='stable'), samples, tag='hilbert 16^3 (current)')
    h5 = hilbert_keys(P, 5)
    simulate(P, np.argsort(h5, kind='stable'), samples, tag='hilbert 32^3')
    h6 = hilbert_keys(P, 6)
    simulate(P, np.argsort(h6, kind='stable'), samples, tag='hilbert 64^3')
    h8 = hilbert_keys(P, 8)
    simulate(P, np.argsort(h8, kind='stable'), samples, tag='hilbert 256^3')
    simulate(P, kd_order(P), samples, tag='kd median leaves of 64')

def sample_kd_order(P, nsamp, depth=10, seed=1):
    rng = np.random.default_rng(seed)
    samp = P[rng.choice(len(P), nsamp, replace=False)]
    # build splitters: implicit binary tree, node i has children 2i+1, 2i+2
    nn = (1 << depth) - 1
    axis = np.zeros(nn, np.int64); val = np.zeros(nn, np.float32)
    def build(node, pts, lo, hi, level):
        if level == depth: return
        if len(pts) == 0:
            a = int(np.argmax(hi - lo)); v = 0.5 * (lo[a] + hi[a]); l = r = pts
        else:
            ext = pts.max(0) - pts.min(0); a = int(np.argmax(ext)); v = np.float32(np.median(pts[:, a]))
            l = pts[pts[:, a] < v]; r = pts[pts[:, a] >= v]
        axis[node] = a; val[node] = v
        h2 = hi.copy(); h2[a] = v; l2 = lo.copy(); l2[a] = v
        build(2 * node + 1, l, lo, h2, level + 1); build(2 * node + 2, r, l2, hi, level + 1)
    build(0, samp, P.min(0), P.max(0), 0)
    node = np.zeros(len(P), np.int64)
    for lvl in range(depth):
        a = axis[node]; v = val[node]
        right = P[np.arange(len(P)), a] >= v
        node = 2 * node + 1 + right
    leaf = node - nn
    cnt = np.bincount(leaf, minlength=1 << depth)
    print(f"   sample-kd {nsamp}: leaf size mean {cnt.mean():.1f} std {cnt.std():.1f} max {cnt.max()}")
    return np.argsort(leaf, kind='stable')

if __name__ == '__main__':
    for ns in (2048, 4096, 8192, 16384):
        simulate(P, sample_kd_order(P, ns), samples, tag=f'sample-kd {ns} depth 10')
    simulate(P, sample_kd_order(P, 8192, depth=9), samples, tag=f'sample-kd 8192 depth 9')
    simulate(P, sample_kd_order(P, 8192, depth=11), samples, tag=f'sample-kd 8192 depth 11')
    simulate(P, sample_kd_order(P, len(P), depth=10), samples, tag=f'sample-kd all depth 10')

def simulate_multi(P, order, samples, parts, BS=64, tag=''):
    S = P[order]; M = len(S); NB = (M + BS - 1) // BS
    pad = NB * BS - M
    Sp = np.concatenate([S, np.repeat(S[-1:], pad, 0)]) if pad else S
    B3 = Sp.reshape(NB, BS, 3)
    sub = B3.reshape(NB, parts, BS // parts, 3)
    blo, bhi = sub.min(2), sub.max(2)              # (NB, parts, 3)
    md = np.full((NB, BS), np.inf, np.float32); bmax = np.full(NB, np.inf, np.float32)
    act = []
    for k, si in enumerate(samples[:-1]):
        s = P[si]
        q = np.clip(s, blo, bhi); lb = ((s - q) ** 2).sum(2).min(1)
        a = lb < bmax
        act.append(int(a.sum()))
        ids = np.nonzero(a)[0]
        dd = ((B3[ids] - s) ** 2).sum(2)
        md[ids] = np.minimum(md[ids], dd); bmax[ids] = md[ids].max(1)
    act = np.array(act)
    print(f"{tag:28s} mean active/round: all {act.mean():6.1f}  r1-255 {act[1:256].mean():6.1f}  r256+ {act[256:].mean():6.2f}")

if __name__ == '__main__':
    o = np.argsort(h4, kind='stable')
    for parts in (1, 2, 4):
        simulate_multi(P, o, samples, parts, tag=f'hilbert16 {parts} sub-boxes')
    o = np.argsort(h8, kind='stable')
    for parts in (2, 4):
        simulate_multi(P, o, samples, parts, tag=f'hilbert256 {parts} sub-boxes')

def simulate_pstar(P, order, samples, PEND, BS=64, tag=''):
    """bucket keeps an exact key while no pending sample hits its arg-max point p*; flush on hit or when the list is full"""
    S = P[order]; M = len(S); NB = (M + BS - 1) // BS
    pad = NB * BS - M
    Sp = np.concatenate([S, np.repeat(S[-1:], pad, 0)]) if pad else S
    B3 = Sp.reshape(NB, BS, 3)
    blo, bhi = B3.min(1), B3.max(1)
    md = np.full((NB, BS), np.inf, np.float32); bmax = np.full(NB, np.inf, np.float32)
    pstar = B3[:, 0, :].copy()
    pend = [[] for _ in range(NB)]
    fl = []; applied = []
    Pm = P
    for k, si in enumerate(samples[:-1]):
        s = Pm[si]
        q = np.clip(s, blo, bhi); lb = ((s - q) ** 2).sum(1)
        a = np.nonzero(lb < bmax)[0]
        hit = ((pstar[a] - s) ** 2).sum(1) < bmax[a]
        nf = 0; na = 0
        for j, h in zip(a, hit):
            pend[j].append(s)
            if h or len(pend[j]) >= PEND:
                for t in pend[j]:
                    md[j] = np.minimum(md[j], ((B3[j] - t) ** 2).sum(1))
                na += len(pend[j]); pend[j] = []
                i = int(np.argmax(md[j])); bmax[j] = md[j, i]; pstar[j] = B3[j, i]
                nf += 1
        fl.append(nf); applied.append(na)
    fl = np.array(fl); applied = np.array(applied)
    print(f"{tag:28s} flushes/round: all {fl.mean():6.2f}  r1-255 {fl[1:256].mean():6.2f}  r256+ {fl[256:].mean():6.2f}   samples applied per flush {applied.sum() / max(fl.sum(), 1):.2f}")

if __name__ == '__main__':
    o = np.argsort(h4, kind='stable')
    for pd in (1, 3, 4, 8, 1000):
        simulate_pstar(P, o, samples, pd, tag=f'p* trigger, list {pd}')


def simulate_lazy(P, order, samples, PEND, mode, ALPHA=0.95, BS=64, tag=''):
    """Deferred rounds as in fps_lazy_rounds.  mode 'ub': a bucket a sample could change becomes DIRTY at once (its key an
    upper bound, flushed when the bound reaches tau or the list is full) — the kernel's scheme.  mode 'pstar+ub': it stays
    CLEAN (exact key, still a candidate) while no pending sample reaches its arg-max point; only such a hit makes it dirty."""
    S = P[order]; M = len(S); NB = (M + BS - 1) // BS
    pad = NB * BS - M
    Sp = np.concatenate([S, np.repeat(S[-1:], pad, 0)]) if pad else S
    B3 = Sp.reshape(NB, BS, 3)
    blo, bhi = B3.min(1), B3.max(1)
    md = np.full((NB, BS), np.inf, np.float32)
    key = np.full(NB, np.inf, np.float32)          # exact max (clean) or upper bound (dirty)
    pstar = B3[:, 0, :].copy(); pidx = np.zeros(NB, np.int64)
    dirty = np.zeros(NB, bool)
    pend = [[] for _ in range(NB)]
    nfl = 0; nap = 0; second = 0; fl_late = 0
    picks = [0]
    tau = 0.0
    s = P[order[0]] if False else P[samples[0]]
    K = len(samples)
    def flush(j):
        nonlocal nfl, nap
        for t in pend[j]:
            md[j] = np.minimum(md[j], ((B3[j] - t) ** 2).sum(1))
        nap += len(pend[j]); pend[j] = []; dirty[j] = False
        i = int(np.argmax(md[j])); key[j] = md[j, i]; pstar[j] = B3[j, i]; pidx[j] = i
        nfl += 1
    for k in range(K - 1):
        s = P[samples[k]]
        q = np.clip(s, blo, bhi); lb = ((s - q) ** 2).sum(1)
        aff = np.nonzero(lb < key)[0]
        far = np.where(np.abs(s - blo) >= np.abs(s - bhi), blo, bhi)
        ubs = ((s - far) ** 2).sum(1)
        hit = ((pstar - s) ** 2).sum(1) < key
        before = nfl
        for j in aff:
            pend[j].append(s)
            if mode == 'ub' or dirty[j] or hit[j]:
                dirty[j] = True
                key[j] = min(key[j], ubs[j])
        for j in aff:
            if (dirty[j] and key[j] >= tau) or len(pend[j]) >= PEND + (0 if dirty[j] else 0):
                if len(pend[j]) >= PEND or dirty[j]:
                    flush(j)
        # pick among clean buckets
        while True:
            clean = ~dirty
            v = key[clean].max() if clean.any() else -1.0
            if v >= tau or not dirty.any():
                break
            tau = v; second += 1
            for j in np.nonzero(dirty & (key >= tau))[0]:
                flush(j)
        if k >= 256: fl_late += nfl - before
        tau = ALPHA * v
    print(f"{tag:34s} flushes/round: all {nfl / (K - 1):6.2f}  first 256 {(nfl - fl_late) / 256:6.2f}  late {fl_late / (K - 1 - 256):6.2f}   samples/flush {nap / max(nfl, 1):.2f}   second iterations {second}")


if __name__ == '__main__':
    o = np.argsort(h4, kind='stable')
    simulate_lazy(P, o, samples, 3, 'ub', tag='deferred, dirty at once (kernel)')
    simulate_lazy(P, o, samples, 3, 'pstar+ub', tag='deferred, clean until p* is hit')
    simulate_lazy(P, o, samples, 8, 'pstar+ub', tag='same, 8-entry lists')
    for pd in (4, 5, 6, 7):
        simulate_lazy(P, o, samples, pd, 'ub', tag=f'kernel scheme, {pd}-entry lists')
    for al in (0.9, 0.85, 0.8):
        simulate_lazy(P, o, samples, 7, 'ub', ALPHA=al, tag=f'kernel scheme, 7 entries, alpha {al}')
