"""Float64 restatement of the TRAINABLE half of the Siamese step (TEST INFRASTRUCTURE ONLY -- never imported by dh3d_amd).

`head_loss_f64` is the global head of core/model.py:112-133 (global_before_assemble flex_conv -> three_interpolate ->
attention MLP -> NetVLAD + context gating, core/backbones.py:156-320) in TRAINING mode followed by l2-normalisation and
lazy_quadruplet_loss (core/losses.py:173-200), written with float64 numpy gathers and matmuls as a function of the head's
weights; the frozen inputs -- local descriptors and the integer geometry (FPS picks, sampled-set kNN, three_nn) -- come
from the float32 oracle (oracle/model_np.training_step_forward with a `trace`).  It exists for ONE purpose: central
differences along parameter directions that are accurate to ~1e-9, against which the HIP backward's gradients are
projected (tests/test_cfg4_gpu.py::test_cfg4_gradients_vs_float64_central_differences_of_the_oracle_graph).  It is
anchored on the float32 oracle by its loss (same graph, same ids: agreement to float32 rounding).
PARITY UNPINNED like the rest of oracle/model_np.py (no reference vector exists for the head).
"""
import numpy as np

D64 = np.float64


def _bn_train(x, gamma, beta, eps):
    """batch statistics over every axis but the last, biased variance (what normalises in training mode)."""
    ax = tuple(range(x.ndim - 1))
    mu = x.mean(axis=ax, keepdims=True)
    var = x.var(axis=ax, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * gamma + beta


def _l2n(x, axis, eps):
    return x / np.sqrt(np.maximum(np.sum(x * x, axis=axis, keepdims=True), eps))


def _sig(x):
    return 1.0 / (1.0 + np.exp(-x))


def head_loss_f64(w, points, localdesc, fps_idx, knn, nn3_idx, nn3_dist, batch_size, num_pos, num_neg, margin1=0.5,
                  margin2=0.2, tp_eps=1e-5, slim_eps=1e-3):
    """w: {TF variable name: array} (any float dtype; cast to float64 here).  points [Bt,N,3], localdesc [Bt,N,128],
    fps_idx [Bt,m], knn [Bt,K,m] (the oracle's layout), nn3_idx / nn3_dist [Bt,N,3].  Returns the loss (float64)."""
    g = lambda name: np.asarray(w[name], D64)
    P, F = np.asarray(points, D64), np.asarray(localdesc, D64)
    Bt, N, _ = P.shape
    b1 = np.arange(Bt)[:, None]
    xyz_s, feat_s = P[b1, fps_idx], F[b1, fps_idx]                       # group_point (tf_utils.py:92-95)
    nb = np.asarray(knn).transpose(0, 2, 1)                              # [Bt,m,K]
    b2 = np.arange(Bt)[:, None, None]
    fn = feat_s[b2, nb]                                                  # [Bt,m,K,128]
    dp = xyz_s[b2, nb] - xyz_s[:, :, None, :]                            # centre = the point itself
    sc = "global_before_assemble/flexconv_0"
    theta, bias = g(sc + "/position_theta"), g(sc + "/position_bias")
    x = fn.sum(2) @ bias
    for d in range(3):
        x = x + (dp[..., d:d + 1] * fn).sum(2) @ theta[d]
    x = x + g(sc + "/feature_bias").reshape(1, 1, -1)
    x = np.maximum(_bn_train(x, g(sc + "_bn/gamma"), g(sc + "_bn/beta"), tp_eps), 0)      # [Bt,m,256]
    dist = np.maximum(np.asarray(nn3_dist, D64), 1e-10)
    wt = (1.0 / dist) / np.sum(1.0 / dist, axis=2, keepdims=True)
    up = (x[b2, nn3_idx] * wt[..., None]).sum(2)                          # three_interpolate [Bt,N,256]
    a = "globalatt/detec_conv0"
    h = up @ g(a + "/W").reshape(up.shape[2], -1) + g(a + "/b")
    h = np.maximum(_bn_train(h, g(a + "/bn/gamma"), g(a + "/bn/beta"), tp_eps), 0)
    fc = "globalatt/detec_conv_fc"
    att = _sig(h @ g(fc + "/W").reshape(-1, 1) + g(fc + "/b"))           # [Bt,N,1]
    # NetVLAD + context gating (backbones.py:202-320)
    Dm, C = up.shape[2], g("cluster_weights").shape[1]
    xr = _l2n(up.reshape(-1, Dm), 1, 1e-12)
    act = _bn_train(xr @ g("cluster_weights"), g("cluster_bn/gamma"), g("cluster_bn/beta"), slim_eps)
    act = np.exp(act - act.max(axis=1, keepdims=True))
    act = act / act.sum(axis=1, keepdims=True) * att.reshape(-1, 1)
    act = act.reshape(Bt, N, C)
    vlad = np.matmul(act.transpose(0, 2, 1), xr.reshape(Bt, N, Dm)).transpose(0, 2, 1) \
        - act.sum(axis=1, keepdims=True) * g("cluster_weights2")
    vlad = _l2n(_l2n(vlad, 1, 1e-12).reshape(Bt, C * Dm), 1, 1e-12)
    v = _bn_train(vlad @ g("hidden1_weights"), g("bn/gamma"), g("bn/beta"), slim_eps)
    gates = _bn_train(v @ g("gating_weights"), g("gating_bn/gamma"), g("gating_bn/beta"), slim_eps)
    desc = _l2n(v * _sig(gates), 1, 1e-8)                                 # model.py:205
    # lazy_quadruplet_loss (losses.py:173-200)
    a_, b_ = batch_size, batch_size + num_pos * batch_size
    c_ = b_ + num_neg * batch_size
    Dd = desc.shape[1]
    q, pos = desc[:a_].reshape(batch_size, 1, Dd), desc[a_:b_].reshape(batch_size, num_pos, Dd)
    neg, oth = desc[b_:c_].reshape(batch_size, num_neg, Dd), desc[c_:].reshape(batch_size, 1, Dd)
    best_pos = np.min(np.sum((pos - q) ** 2, 2), 1).reshape(-1, 1)
    trip = np.mean(np.max(np.maximum(margin1 + best_pos - np.sum((neg - q) ** 2, 2), 0), 1))
    second = np.mean(np.max(np.maximum(margin2 + best_pos - np.sum((neg - oth) ** 2, 2), 0), 1))
    return float(trip + second)
