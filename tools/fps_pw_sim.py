import numpy as np, sys
from fps_list_sim import morton_sorted, seq_fps
from fps_ctl_sim import make
def run(N, m, W, L, lo, hi, CAP, kind, seed=0, verbose=True, up=1.3, dn=0.7):
    p = make(N, kind, seed)
    ps = morton_sorted((p - p.min(0)) / (p.max(0) - p.min(0) + 1e-9))
    R = N // W; PPT = R // 64
    md = np.full(N, 1e38, np.float32)
    picks = [0]
    d = ((ps - ps[0]) ** 2).sum(1).astype(np.float32); md = np.minimum(md, d)
    syncs = 0; pools = []; ovf = 0
    lists = [None] * W; bounds = np.zeros(W, np.float32)
    delta = np.full(W, 0.05, np.float32)   # relative to the wave's max
    while len(picks) < m:
        syncs += 1
        for w in range(W):
            v = md[w * R:(w + 1) * R].reshape(PPT, 64)
            j1 = np.argmax(v, axis=0); b1 = v[j1, np.arange(64)]
            v2 = v.copy(); v2[j1, np.arange(64)] = -2; b2 = v2.max(axis=0)
            wl = int(np.argmax(b1)); tau = b1[wl] * (1 - delta[w])
            flagged = (b1 > tau); flagged[wl] = True
            cnt = int(flagged.sum())
            order = [wl] + [l for l in range(64) if flagged[l] and l != wl]
            listed = order[:L]
            if cnt > L: ovf += 1
            unl = np.ones(64, bool); unl[listed] = False
            bnd = max(b2.max(), b1[unl].max() if unl.any() else -2)
            lists[w] = [(w * R + int(j1[l]) * 64 + l) for l in listed]; bounds[w] = bnd
            if cnt > hi: delta[w] *= dn
            elif cnt < lo: delta[w] = min(delta[w] * up, 0.5)
        cand = np.array(sum(lists, [])); cv = md[cand].copy(); cp = ps[cand]; RB = bounds.max()
        pools.append(len(cand))
        acc = 0
        while acc < CAP and len(picks) < m:
            j = int(np.argmax(cv))
            if acc > 0 and not cv[j] > RB: break
            picks.append(int(cand[j])); acc += 1
            d = ((cp - cp[j]) ** 2).sum(1).astype(np.float32); cv = np.minimum(cv, d)
        for i in picks[-acc:]:
            d = ((ps - ps[i]) ** 2).sum(1).astype(np.float32); md = np.minimum(md, d)
    ok = seq_fps(ps, m) == picks
    if verbose:
        print(f"N {N} W {W} L {L} lo/hi {lo}/{hi} CAP {CAP} {kind}: syncs {syncs} picks/sync {(m-1)/syncs:.2f} pool mean {np.mean(pools):.0f} max {max(pools)} overflow {ovf/(syncs*W):.2f} exact {ok}")
    return syncs
if __name__ == "__main__":
    N = int(sys.argv[1]); kind = sys.argv[2]
    for (L, lo, hi) in [(4,2,3),(4,3,4),(8,4,6),(8,5,7),(8,6,8),(16,8,12),(16,10,14)]:
        run(N, N // 8, 16, L, lo, hi, 64, kind)
