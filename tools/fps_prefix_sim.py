"""Dev (round 6): how many picks of a sync of the list + judge FPS could be accepted WITHOUT the sequential loop.
A pool candidate c is 'unaffected' if no candidate that is picked before it in this sync lies within sqrt(value(c)) of it;
the longest rank-ordered prefix of unaffected candidates above the bound RB is a batch of picks that needs no sequential
arg-max / update chain (all-pairs test, parallel over the waves).  Counts, per sync: picks accepted by the sequential judge,
length of the unaffected prefix, and how many sequential picks are left after it."""
import numpy as np, sys
sys.path.insert(0, __file__.rsplit("/", 1)[0])
from fps_list_sim import morton_sorted
def run(N, m, W, D, kind, seed=0):
    rng = np.random.default_rng(seed)
    if kind == "cube": p = rng.random((N, 3)).astype(np.float32)
    else:
        a = rng.random((N, 3)).astype(np.float32); a[: N // 2, 2] *= 0.02; a[N // 2:, 0] *= 0.05; p = a
    ps = morton_sorted(p); R = N // W
    md = ((ps - ps[0]) ** 2).sum(1).astype(np.float32)
    npick = 1; syncs = 0; tot_seq = 0; tot_prefix = 0; left = 0; hist = {}
    while npick < m:
        syncs += 1
        cand = []; RB = -1.0
        for w in range(W):
            v = md[w * R:(w + 1) * R]
            o = np.argsort(-v, kind="stable")[:D + 1]
            cand += [w * R + int(i) for i in o[:D]]
            RB = max(RB, float(v[o[D]]))
        cand = np.array(cand); v0 = md[cand].copy(); cp = ps[cand]
        # sequential judge
        cv = v0.copy(); seq = []
        while len(seq) < 32 and npick + len(seq) < m:
            j = int(np.argmax(cv))
            if seq and not cv[j] > RB: break
            seq.append(j)
            d = ((cp - cp[j]) ** 2).sum(1).astype(np.float32); cv = np.minimum(cv, d)
        # unaffected prefix in rank order of the ORIGINAL values
        order = np.argsort(-v0, kind="stable")
        k = 0
        for r, c in enumerate(order):
            if r > 0 and not v0[c] > RB: break
            if r >= 32 or npick + r >= m: break
            d = ((cp[order[:r]] - cp[c]) ** 2).sum(1) if r else np.array([np.inf])
            if r and d.min() < v0[c]: break
            k += 1
        assert list(order[:k]) == seq[:k], (list(order[:k]), seq[:k])
        tot_seq += len(seq); tot_prefix += k; left += len(seq) - k
        hist[len(seq) - k] = hist.get(len(seq) - k, 0) + 1
        for j in seq:
            d = ((ps - cp[j]) ** 2).sum(1).astype(np.float32); md = np.minimum(md, d)
        npick += len(seq)
    print(f"N {N} W {W} D {D} {kind}: syncs {syncs}, picks/sync {tot_seq/syncs:.2f}, unaffected prefix {tot_prefix/syncs:.2f} "
          f"({100*tot_prefix/tot_seq:.0f} % of the picks), sequential picks left per sync {left/syncs:.2f}; histogram of left: {dict(sorted(hist.items()))}")
if __name__ == "__main__":
    N = int(sys.argv[1]); kind = sys.argv[2] if len(sys.argv) > 2 else "cube"
    for D in (2, 3, 4):
        run(N, N // 8, 16, D, kind)
