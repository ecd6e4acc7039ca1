import numpy as np, sys
from fps_list_sim import morton_sorted, seq_fps
def make(N, kind, seed):
    rng = np.random.default_rng(seed)
    if kind == "cube": return rng.random((N, 3)).astype(np.float32)
    if kind == "lidar":
        a = rng.random((N, 3)).astype(np.float32); a[: N // 2, 2] *= 0.02; a[N // 2:, 0] *= 0.05; return a
    if kind == "gauss": return rng.standard_normal((N, 3)).astype(np.float32)
    if kind == "lattice":
        g = int(round(N ** (1/3))) + 1
        a = np.stack(np.meshgrid(*[np.arange(g)]*3, indexing="ij"), -1).reshape(-1, 3)[:N].astype(np.float32) / g
        return a[rng.permutation(N)]
def run(N, m, W, L, target, CAP, kind, seed=0, verbose=True):
    p = make(N, kind, seed)
    ps = morton_sorted((p - p.min(0)) / (p.max(0) - p.min(0) + 1e-9)); ps = p[np.lexsort((np.zeros(N),))] if False else ps
    R = N // W; PPT = R // 64
    md = np.full(N, 1e38, np.float32)
    picks = [0]
    d = ((ps - ps[0]) ** 2).sum(1).astype(np.float32); md = np.minimum(md, d)
    syncs = 0; tau = np.float32(1e30); pools = []; hist = {}; accs = []
    lists = [None] * W; bounds = np.zeros(W, np.float32)
    delta = None
    while len(picks) < m:
        syncs += 1
        for w in range(W):
            v = md[w * R:(w + 1) * R].reshape(PPT, 64)
            j1 = np.argmax(v, axis=0); b1 = v[j1, np.arange(64)]
            v2 = v.copy(); v2[j1, np.arange(64)] = -2; b2 = v2.max(axis=0)
            wl = int(np.argmax(b1))
            flagged = (b1 > tau); flagged[wl] = True
            order = [wl] + [l for l in range(64) if flagged[l] and l != wl]
            listed = order[:L]
            unl = np.ones(64, bool); unl[listed] = False
            bnd = max(b2[listed].max() if True else 0, b2.max(), b1[unl].max() if unl.any() else -2)
            lists[w] = [(w * R + int(j1[l]) * 64 + l) for l in listed]; bounds[w] = bnd
        cand = np.array(sum(lists, [])); cv = md[cand].copy(); cp = ps[cand]; RB = bounds.max()
        pools.append(len(cand))
        acc = 0; vlast = None; V0 = cv.max()
        while acc < CAP and len(picks) < m:
            j = int(np.argmax(cv))
            if acc > 0 and not cv[j] > RB: break
            vlast = cv[j]; picks.append(int(cand[j])); acc += 1
            d = ((cp - cp[j]) ** 2).sum(1).astype(np.float32); cv = np.minimum(cv, d)
        for i in picks[-acc:]:
            d = ((ps - ps[i]) ** 2).sum(1).astype(np.float32); md = np.minimum(md, d)
        # controller: listing range that produced this pool -> scale to the target pool
        rng_now = max(float(V0) - float(min(tau, V0)), 1e-3 * float(V0)) if tau < 1e29 else 0.05 * float(V0)
        delta = rng_now * target / max(len(cand), W)
        delta = min(delta, 0.5 * float(vlast))
        tau = np.float32(float(vlast) - delta)
        accs.append(acc)
    ok = seq_fps(ps, m) == picks
    if verbose:
        print(f"N {N} W {W} L {L} target {target} CAP {CAP} {kind}: syncs {syncs} picks/sync {(m-1)/syncs:.2f} pool mean {np.mean(pools):.0f} max {max(pools)} exact {ok}")
    return syncs
if __name__ == "__main__":
    N = int(sys.argv[1]); kind = sys.argv[2]
    for L in (4, 8):
        for t in (24, 32, 48, 64, 96):
            if t <= L * 16: run(N, N // 8, 16, L, t, 64, kind)
    run(N, N // 8, 16, 8, 48, 16, kind); run(N, N // 8, 16, 8, 48, 24, kind); run(N, N // 8, 16, 8, 64, 32, kind)
