import numpy as np, sys
rng = np.random.default_rng(0)
N, m, W = 8192, 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 16
kind = sys.argv[2] if len(sys.argv) > 2 else "cube"
if kind == "cube":
    p = rng.random((N, 3)).astype(np.float32)
else:  # lidar-ish: points on a few planes + noise
    a = rng.random((N, 3)).astype(np.float32); a[: N // 2, 2] *= 0.02; a[N // 2 :, 0] *= 0.05; p = a
# morton sort 6 bits
q = np.minimum((p * 64).astype(np.int64), 63)
def spread(v):
    r = np.zeros_like(v)
    for b in range(6): r |= ((v >> b) & 1) << (3 * b)
    return r
code = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
order = np.argsort(code, kind="stable")
ps = p[order]; reg = np.arange(N) // (N // W)
md = np.full(N, 1e38, np.float32)
def upd(i):
    global md
    d = ((ps - ps[i]) ** 2).sum(1).astype(np.float32); md = np.minimum(md, d)
upd(0); picks = 1; syncs = 0
hist = {}
for T in [int(sys.argv[3]) if len(sys.argv) > 3 else 16]:
    while picks < m:
        syncs += 1
        best1 = np.zeros(W, np.int64); b1v = np.zeros(W, np.float32); b2v = np.zeros(W, np.float32)
        for w in range(W):
            s = slice(w * (N // W), (w + 1) * (N // W)); v = md[s]
            o = np.argsort(-v, kind="stable")[:2]
            best1[w] = s.start + o[0]; b1v[w] = v[o[0]]; b2v[w] = v[o[1]]
        srt = np.argsort(-b1v, kind="stable")
        acc = []
        for j, w in enumerate(srt[:T]):
            c = best1[w]; ok = True
            for w2 in acc:
                c2 = best1[w2]
                d = np.float32(((ps[c] - ps[c2]) ** 2).sum())
                if d < b1v[w] or b2v[w2] >= b1v[w]: ok = False; break
            if not ok: break
            acc.append(w)
            if picks + len(acc) >= m: break
        for w in acc: upd(best1[w])
        picks += len(acc); hist[len(acc)] = hist.get(len(acc), 0) + 1
    print("W", W, kind, "T", T, "syncs", syncs, "picks/sync %.2f" % ((m - 1) / syncs), dict(sorted(hist.items())))
