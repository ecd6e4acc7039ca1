"""Exact batched FPS with depth-D candidate lists per region and a mini-FPS judge.
Checks the pick sequence equals sequential FPS; reports picks/sync."""
import numpy as np, sys
def morton_sorted(p):
    q = np.minimum((p * 64).astype(np.int64), 63)
    def spread(v):
        r = np.zeros_like(v)
        for b in range(6): r |= ((v >> b) & 1) << (3 * b)
        return r
    code = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    return p[np.argsort(code, kind="stable")]
def seq_fps(ps, m):
    md = np.full(len(ps), 1e38, np.float32); out = [0]
    for _ in range(m - 1):
        d = ((ps - ps[out[-1]]) ** 2).sum(1).astype(np.float32); md = np.minimum(md, d)
        out.append(int(np.argmax(md)))
    return out
def run(N, m, W, D, CAP, kind, seed=0):
    rng = np.random.default_rng(seed)
    if kind == "cube": p = rng.random((N, 3)).astype(np.float32)
    else:
        a = rng.random((N, 3)).astype(np.float32); a[: N // 2, 2] *= 0.02; a[N // 2:, 0] *= 0.05; p = a
    ps = morton_sorted(p); R = N // W
    md = np.full(N, 1e38, np.float32)
    def upd(i):
        nonlocal md
        d = ((ps - ps[i]) ** 2).sum(1).astype(np.float32); md = np.minimum(md, d)
    picks = [0]; upd(0); syncs = 0; hist = {}
    while len(picks) < m:
        syncs += 1
        cand = []; RB = -1.0
        for w in range(W):
            v = md[w * R:(w + 1) * R]
            o = np.argsort(-v, kind="stable")[:D + 1]
            cand += [w * R + int(i) for i in o[:D]]
            RB = max(RB, float(v[o[D]]))
        cand = np.array(cand); cv = md[cand].copy(); cp = ps[cand]
        acc = 0
        while acc < CAP and len(picks) < m:
            j = int(np.argmax(cv))   # first max = smallest index (ties ignored in the sim)
            if not cv[j] > RB: break
            picks.append(int(cand[j])); acc += 1
            d = ((cp - cp[j]) ** 2).sum(1).astype(np.float32); cv = np.minimum(cv, d)
        assert acc >= 1
        for i in picks[-acc:]: upd(i)
        hist[acc] = hist.get(acc, 0) + 1
    ref = seq_fps(ps, m)
    ok = ref == picks
    print(f"N {N} W {W} D {D} CAP {CAP} {kind}: syncs {syncs} picks/sync {(m-1)/syncs:.2f} exact {ok}")
    return syncs
if __name__ == "__main__":
    N = int(sys.argv[1]); m = N // 8
    for W, D in [(16,1),(16,2),(16,3),(16,4),(16,6),(16,8),(32,2),(32,4),(64,1),(64,2)]:
        run(N, m, W, D, 64, sys.argv[2] if len(sys.argv) > 2 else "cube")
