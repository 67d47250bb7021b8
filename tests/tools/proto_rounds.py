"""numpy model of the round-based parallel merger (design study; not on any product path).

Mirrors the algorithm the HIP kernels implement: lazy-greedy semantics (stored vs fresh
priority) executed in rounds of mutually-best records.  Used to count rounds and to compare
partitions against the CPU oracle before/while writing the kernels.
"""
import sys, time
import numpy as np

f32 = np.float32
import os
SUBROUNDS = int(os.environ.get("SUBROUNDS","1"))
ALPHA = float(os.environ.get("ALPHA","1.0"))

def score(u, v, S, size, lp, cls, omf, bias):
    cu, cv = cls[u], cls[v]
    cdl = np.zeros(u.shape[0], f32)
    mc = cu.copy()
    d = np.nonzero(cu != cv)[0]
    if d.size:
        j = lp[u[d]] + lp[v[d]]
        m = np.argmax(j, axis=1)
        cdl[d] = (j[np.arange(d.size), m] - lp[u[d], cu[d]]) - lp[v[d], cv[d]]
        mc[d] = m
    den = (size[u] + size[v]).astype(f32)
    pr = (S.astype(f32) * f32(omf) + cdl) / den + f32(bias)
    return pr.astype(f32), mc

def run(cp, sp, offs, omf, bias, max_rounds=100000, verbose=True, finisher_at=0):
    C, H, W = cp.shape
    N = H * W
    eps = np.finfo(f32).eps
    cp = cp.clip(eps, 1 - eps).astype(f32); sp = sp.clip(eps, 1 - eps).astype(f32)
    lp = np.log(cp).reshape(C, N).T.copy()            # [N, C] float32
    cls = np.argmax(lp, axis=1).astype(np.int32)
    size = np.ones(N, np.int64)
    parent = np.arange(N)
    us, vs, Ss = [], [], []
    for k, (di, dj) in enumerate(offs):
        r0, r1 = max(0, -di), min(H, H - di); c0, c1 = max(0, -dj), min(W, W - dj)
        if r0 >= r1 or c0 >= c1: continue
        rr, cc = np.meshgrid(np.arange(r0, r1), np.arange(c0, c1), indexing='ij')
        p = (rr * W + cc).ravel(); q = ((rr + di) * W + cc + dj).ravel()
        pv = sp[k].ravel()[p]
        same = np.log(pv); diff = np.log(1.0 - pv.astype(np.float64)).astype(f32)
        us.append(np.minimum(p, q)); vs.append(np.maximum(p, q)); Ss.append((same - diff).astype(np.float64))
    u = np.concatenate(us); v = np.concatenate(vs); S = np.concatenate(Ss)
    stored, mcls = score(u, v, S, size, lp, cls, omf, bias)
    rounds = 0; merges_total = 0
    hist = []
    while rounds < max_rounds:
        fresh, mcls = score(u, v, S, size, lp, cls, omf, bias)
        vis = stored >= 0
        # eager refresh of stale-high records (semantically neutral)
        down = vis & (fresh < stored)
        stored = np.where(down, fresh, stored)
        vis = stored >= 0
        idx = np.nonzero(vis)[0]
        if idx.size == 0: break
        key = stored[idx]
        # greedy matching by SUBROUNDS of locally dominant records on the static scored graph
        free = np.ones(N, bool)
        cand_all = []
        ball = np.full(N, -1.0, f32)
        np.maximum.at(ball, u[idx], stored[idx]); np.maximum.at(ball, v[idx], stored[idx])
        k_ = stored[idx]; bu = ball[u[idx]]; bv = ball[v[idx]]
        pos = k_ > f32(bias)
        ok = np.where(pos, (k_ - f32(bias) >= f32(ALPHA) * (bu - f32(bias))) & (k_ - f32(bias) >= f32(ALPHA) * (bv - f32(bias))),
                      (k_ >= bu) & (k_ >= bv))
        elig = idx[ok]
        for sub in range(SUBROUNDS):
            elig = elig[free[u[elig]] & free[v[elig]]]
            if elig.size == 0: break
            key = stored[elig]
            obj = np.concatenate([u[elig], v[elig]]); par = np.concatenate([v[elig], u[elig]])
            kk = np.concatenate([key, key]); rid = np.concatenate([elig, elig])
            order = np.lexsort((par, -kk, obj))
            obj_s = obj[order]
            first = np.ones(obj_s.size, bool); first[1:] = obj_s[1:] != obj_s[:-1]
            best_rec = np.full(N, -1); best_rec[obj_s[first]] = rid[order][first]
            c = elig[(best_rec[u[elig]] == elig) & (best_rec[v[elig]] == elig)]
            if c.size == 0: break
            cand_all.append(c)
            free[u[c]] = False; free[v[c]] = False
        cand = np.concatenate(cand_all)
        stale_low = fresh[cand] > stored[cand]
        stored[cand[stale_low]] = fresh[cand[stale_low]]
        m = cand[~stale_low]
        rounds += 1
        if m.size == 0:
            hist.append((idx.size, 0)); continue
        a, b = u[m].copy(), v[m].copy()
        sw = size[a] < size[b]
        a[sw], b[sw] = v[m][sw], u[m][sw]
        cls[a] = mcls[m]; size[a] += size[b]; lp[a] += lp[b]; parent[b] = a
        merges_total += m.size
        absorbed = np.zeros(N, bool); absorbed[b] = True
        touched = absorbed[u] | absorbed[v]
        nu, nv = parent[u], parent[v]
        keep = nu != nv
        nu, nv, S2, st2, t2 = nu[keep], nv[keep], S[keep], stored[keep], touched[keep]
        lo, hi = np.minimum(nu, nv), np.maximum(nu, nv)
        keyp = lo.astype(np.int64) * N + hi
        order = np.argsort(keyp, kind='stable')
        keyp = keyp[order]; S2 = S2[order]; st2 = st2[order]; t2 = t2[order]
        first = np.ones(keyp.size, bool); first[1:] = keyp[1:] != keyp[:-1]
        grp = np.cumsum(first) - 1
        u = (keyp[first] // N); v = (keyp[first] % N)
        S = np.bincount(grp, weights=S2)
        tch = np.bincount(grp, weights=t2) > 0
        st = np.full(u.size, -np.inf, f32)
        np.maximum.at(st, grp, st2)     # untouched groups have exactly one member
        fresh2, _ = score(u, v, S, size, lp, cls, omf, bias)
        stored = np.where(tch, fresh2, st).astype(f32)
        hist.append((idx.size, m.size))
        if verbose and (rounds < 10 or rounds % 20 == 0):
            print("round %d: visible %d records %d merges %d objects %d" % (rounds, idx.size, u.size, m.size, N - merges_total), flush=True)
    # final labels
    root = parent.copy()
    while True:
        r2 = root[root]
        if np.array_equal(r2, root): break
        root = r2
    return root.reshape(H, W), cls, rounds, hist

if __name__ == "__main__":
    sys.path.insert(0, '.')
    from mergenet_amd import synth
    from oracle import checker as ck
    H, W = int(sys.argv[1]), int(sys.argv[2]); noise = float(sys.argv[3]); seed = int(sys.argv[4])
    kind = sys.argv[5] if len(sys.argv) > 5 else 'synth'
    if kind == 'adv':
        offs = synth.generate_offsets(6, 5); C = 4
        s = synth.adversarial(H, W, C, offs, seed)
    else:
        offs = synth.generate_offsets(40, 10); C = 9
        s = synth.synth_v1(H, W, C, offs, seed, noise=noise)
    t = time.time(); part, cls, rounds, hist = run(s.class_probs, s.sameness_probs, offs, 1.0, 0.03); tp = time.time() - t
    t = time.time(); o = ck.run_csegment(s.class_probs, s.sameness_probs, C, offs, 0, 1.0, 0.03); to = time.time() - t
    print("rounds", rounds, "proto %.1fs oracle %.1fs" % (tp, to), "objects", np.unique(part).size, o.stats['n_objects'],
          "same partition:", ck.same_partition(part, o.partition), "mismatch px", ck.partition_mismatch(part, o.partition))
