"""GPU: the steps-in-flight engine (dh3d_amd/engine.py) -- the mode bench.py's `value` is measured in.

  * every slot of a depth-4 pipeline at the BASELINE shape (B=8, N=8192), each with its OWN batch, against the serial
    forward of the same batch: kNN / FPS / sampled-set kNN / three_nn ids bit-equal, the descriptors within 2e-6 (the
    engine's mode selects the one-launch local tail); one cloud per slot against the CPU oracle (oracle/model_np.py)
    within 1e-4;
  * the persistent flex_conv's placement hint (`reserve_cus_per_xcd`, the only thing a step in flight changes inside
    a kernel) at every value the engine can produce, against the oracle on a row sample and bit-equal to hint 0;
  * the global path two deep.
"""
import numpy as np
import pytest
import torch

from test_parity_fullsize_gpu import _build, _weights_np

pytestmark = pytest.mark.gpu

IDS = ("knn_inds", "fps_inds", "sampled_knn_inds", "nn3_inds")


def _batches(n, B, N, seed, dev):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        p = rng.random((B, N, 3), dtype=np.float32)
        if i % 3 == 2:
            p = p * 40 - 20  # Oxford-like extent
        out.append(torch.from_numpy(p).to(dev))
    return out


def test_local_pipeline_depth4_B8_N8192_slots_equal_serial_and_oracle(dev):
    from oracle import model_np
    m = _build("basic_config", dev, seed=41)
    B, N, depth = 8, 8192, 4
    fetch = ("xyz_feat",) + IDS
    batches = _batches(2 * depth + 1, B, N, 4100, dev)  # every slot used twice (+1: the ring wraps unevenly)
    with torch.no_grad():
        serial = []
        for b in batches:
            o = m(b, fetch=fetch)
            serial.append({k: o[k].clone() for k in fetch})
        pipe = m.pipeline(batches[0], depth=depth, outputs=fetch)
        assert pipe.depth == depth and len({pipe.input_buffer(k).data_ptr() for k in range(depth)}) == depth
        got = list(pipe.map(batches))
    torch.cuda.synchronize()
    assert len(got) == len(batches)
    for i, (g, s) in enumerate(zip(got, serial)):
        for k in IDS:
            assert torch.equal(g[k], s[k]), (i, k)
        # coordinates bit-equal; descriptors: with steps in flight the local tail runs as ONE launch (both 64 -> 128 GEMMs,
        # up-sampling, BatchNorm / ReLU, sum and row normalisation in the accumulator layout, csrc/dense_tail.hip) where the
        # serial forward runs three -- the same bf16x6 products summed in another association: 2e-6 on unit-norm rows
        # (tests/test_pm_gpu.py::test_local_tail_fused_vs_float64_and_vs_the_three_launch_form), each within 1e-4 of the oracle
        assert torch.equal(g["xyz_feat"][..., :3], s["xyz_feat"][..., :3]), i
        d = float((g["xyz_feat"] - s["xyz_feat"]).abs().max())
        assert d <= 2e-6, (i, "xyz_feat", d)
    # one cloud per slot against the oracle (a single 8192-point cloud takes the C restatement ~2 s)
    w = _weights_np(m)
    for slot in range(depth):
        i, c = slot, (3 * slot + 1) % B
        pts = batches[i][c:c + 1].cpu().numpy()
        trace = {}
        exp = model_np.forward(pts, w, detection=False, extract_global=False, trace=trace)
        g = got[i]
        assert np.array_equal(g["knn_inds"][c:c + 1].cpu().numpy(), exp["knn_indices"].transpose(0, 2, 1))
        assert np.array_equal(g["fps_inds"][c:c + 1].cpu().numpy(), trace["stage2/fps_idx"])
        assert np.array_equal(g["sampled_knn_inds"][c:c + 1].cpu().numpy(), trace["stage2/knn"].transpose(0, 2, 1))
        assert np.array_equal(g["nn3_inds"][c:c + 1].cpu().numpy(), trace["stage2/nn3_idx"])
        err = float(np.abs(g["xyz_feat"][c:c + 1].cpu().numpy() - exp["xyz_feat"]).max())
        assert err < 1e-4, (slot, err)


def test_pipeline_submit_result_zero_copy_and_slot_reuse_guard(dev):
    m = _build("basic_config", dev, seed=42)
    B, N, depth = 2, 4096, 2
    batches = _batches(4, B, N, 4200, dev)
    with torch.no_grad():
        serial = [m(b, fetch=("xyz_feat",))["xyz_feat"].clone() for b in batches]
        pipe = m.pipeline(batches[0], depth=depth, outputs=("xyz_feat",))
        # zero-copy: the batch is written into the slot's buffer on the slot's stream, submit() takes no argument
        tickets, outs = [], []
        for i, b in enumerate(batches):
            k = pipe.next_slot
            if i >= depth:  # the slot's previous result must be consumed before its buffers are overwritten
                o = pipe.result(tickets[i - depth])
                outs.append(o["xyz_feat"].clone())
                pipe.release(tickets[i - depth])
            with torch.cuda.stream(pipe.stream(k)):
                pipe.input_buffer(k).copy_(b, non_blocking=True)
            tickets.append(pipe.submit())
        with pytest.raises(RuntimeError):
            pipe.result(tickets[0])  # reused since
        for t in tickets[-depth:]:
            outs.append(pipe.result(t, wait="host")["xyz_feat"].clone())
    torch.cuda.synchronize()
    for i in range(len(batches)):
        assert torch.equal(outs[i], serial[i]), i
    m.invalidate()
    m.prepare()
    with pytest.raises(RuntimeError):
        pipe.submit()


@pytest.mark.parametrize("Din", [32, 64])
def test_flex_conv_x6_reserve_hint_vs_oracle(dev, oracle, Din):
    """B=8 x N=8192, K=8, reserve_cus_per_xcd in {0, 1, 3, 4, 8}: a different tile -> workgroup assignment each, the same
    bits out; rows sampled against the oracle's reference formulation (the full launch is 19 GF of scalar C)."""
    from dh3d_amd import pm
    g = torch.Generator().manual_seed(100 + Din)
    B, N, K, Dout = 8, 8192, 8, 64
    xyz = torch.rand(B, N, 3, generator=g).to(dev)
    nbr, _ = pm.knn_xyz(xyz, K)
    f = torch.randn(B, N, Din, generator=g).to(dev)
    theta = (torch.randn(3, Din, Dout, generator=g) / Din ** 0.5).to(dev)
    bias = (torch.randn(Din, Dout, generator=g) / (8 * Din) ** 0.5).to(dev)
    wp3 = pm.pack_flex_weight_x3(theta, bias)
    base = pm.flex_conv_x6(f, xyz, nbr, wp3, Dout, reserve_cus_per_xcd=0)
    outs = {0: base}
    for r in (1, 3, 4, 8, 31):
        outs[r] = pm.flex_conv_x6(f, xyz, nbr, wp3, Dout, reserve_cus_per_xcd=r)
        assert torch.equal(outs[r], base), r
    torch.cuda.synchronize()
    # the oracle on every 61st row of every cloud (first and last rows included): the sampled rows' neighbourhoods
    # gathered into a compact cloud of R*K points, row r*K = query r with itself as rank-0 neighbour
    rows = np.unique(np.concatenate([np.arange(0, N, 61), [N - 1]]))
    R = len(rows)
    th, bi = theta.cpu().numpy(), bias.cpu().numpy()
    for b in range(B):
        nb = nbr[b].cpu().numpy()[rows]
        assert np.array_equal(nb[:, 0], rows)
        ids = nb.reshape(-1)
        sub_f = np.ascontiguousarray(f[b].cpu().numpy()[ids].T[None])
        sub_p = np.ascontiguousarray(xyz[b].cpu().numpy()[ids].T[None])
        sub_nb = np.zeros((1, K, R * K), np.int32)
        sub_nb[0, :, ::K] = (np.arange(R)[None, :] * K + np.arange(K)[:, None])
        exp = oracle.flex_convolution(sub_f, sub_p, sub_nb, th, bi, center_self=True)[0][:, ::K].T  # [R, Dout]
        for r in (0, 4, 8):
            got = outs[r][b].cpu().numpy()[rows]
            assert np.abs(got - exp).max() <= 2e-6 * np.abs(exp).max() + 1e-4 * np.abs(exp).mean(), (b, r)


def test_global_pipeline_depth2_B32_N4096_slots_equal_serial_and_oracle(dev):
    from oracle import model_np
    m = _build("global_config", dev, seed=43)
    B, N, depth = 32, 4096, 2
    fetch = ("globaldesc",) + IDS
    batches = _batches(2 * depth + 1, B, N, 4300, dev)
    with torch.no_grad():
        serial = []
        for b in batches:
            o = m(b, fetch=fetch)
            serial.append({k: o[k].clone() for k in fetch})
        pipe = m.pipeline(batches[0], depth=depth, outputs=fetch)
        got = list(pipe.map(batches))
    torch.cuda.synchronize()
    for i, (g, s) in enumerate(zip(got, serial)):
        for k in IDS:
            assert torch.equal(g[k], s[k]), (i, k)
        # (the global tail sums A'^T c and the split-K projection with f32 atomics: run-to-run rounding, not bit-equal)
        d = float((g["globaldesc"] - s["globaldesc"]).abs().max())
        assert d <= 2e-6, (i, d)
    w = _weights_np(m)
    for slot in range(depth):
        i, c = slot + depth, (7 * slot + 3) % B
        pts = batches[i][c:c + 1].cpu().numpy()
        exp = model_np.forward(pts, w, detection=False, extract_global=True)
        err = float(np.abs(got[i]["globaldesc"][c:c + 1].cpu().numpy() - exp["globaldesc"]).max())
        assert err < 1e-4, (slot, err)


def test_replays_refuse_to_run_on_weights_that_moved(dev):
    """A graphed() forward / a Pipeline holds packed and BatchNorm-folded COPIES of the weights: after an optimiser step
    (QuadrupletTrainer: invalidate(head_only=True); LocalTrainer: mark_weights_changed) or load_state_dict a replay
    would silently compute with the old ones -- it raises instead."""
    from dh3d_amd import ConfigFactory
    from dh3d_amd.model import DH3D
    m = DH3D(ConfigFactory("global_config").getconfig()).init_synthetic(0).to(dev).eval().prepare()
    pts = torch.rand(2, 1024, 3, device=dev)
    with torch.no_grad():
        run = m.graphed(pts, outputs=("globaldesc",))
        pipe = m.pipeline(pts, depth=2, outputs=("globaldesc",))
        a = run()["globaldesc"].clone()
        t = pipe.submit(pts)
        assert torch.equal(pipe.result(t, wait="host")["globaldesc"], a)
        m.invalidate(head_only=True)            # what every QuadrupletTrainer step does
        with pytest.raises(RuntimeError):
            run()
        with pytest.raises(RuntimeError):
            pipe.submit(pts)
        run2 = m.graphed(pts, outputs=("globaldesc",))
        assert torch.equal(run2()["globaldesc"], a)
        m.mark_weights_changed()                # what every LocalTrainer step does
        with pytest.raises(RuntimeError):
            run2()


def test_submit_copies_a_batch_the_caller_drops_at_once(dev):
    """submit(batch) copies on the slot's stream: the batch is recorded on that stream, so dropping it right after the
    call (the usual `b = cpu.to(dev, non_blocking=True); pipe.submit(b)` loop) cannot hand its block to the next
    allocation before the copy has read it."""
    from dh3d_amd import ConfigFactory
    from dh3d_amd.model import DH3D
    m = DH3D(ConfigFactory("basic_config").getconfig()).init_synthetic(0).to(dev).eval().prepare()
    rng = np.random.default_rng(5)
    host = [torch.from_numpy(rng.random((2, 2048, 3), dtype=np.float32)) for _ in range(6)]
    with torch.no_grad():
        exp = [m(h.to(dev), fetch=("xyz_feat",))["xyz_feat"].clone() for h in host]
        pipe = m.pipeline(host[0].to(dev), depth=3, outputs=("xyz_feat",))
        tickets, got = [], []
        for h in host:
            if len(tickets) == 3:
                tk = tickets.pop(0)
                got.append(pipe.result(tk)["xyz_feat"].clone()); pipe.release(tk)
            b = h.to(dev, non_blocking=True)
            tickets.append(pipe.submit(b))
            del b
            junk = torch.full((2, 2048, 3), 7.0, device=dev)   # the allocator's next hand-out of that size
            del junk
        for tk in tickets:
            got.append(pipe.result(tk)["xyz_feat"].clone()); pipe.release(tk)
    torch.cuda.synchronize()
    for g, e in zip(got, exp):
        assert torch.equal(g, e)


def test_submit_host_batches_h2d_and_d2h_on_the_slot_stream(dev):
    """The serving loop (localdesc_extract.py:106-138 / globaldesc_extract.py:84-100 feed host arrays and save host arrays):
    pinned HOST batches go in through submit(host_batch) -- an asynchronous H2D copy on the slot's own stream, no ordering
    behind the caller's stream -- and `fetch_to` brings the descriptors back into pinned host memory on the same stream;
    ticket.event.synchronize() = "the descriptors are in host memory".  Every step against the serial forward of its batch."""
    m = _build("global_config", dev, seed=43)
    B, N, depth = 4, 4096, 3
    rng = np.random.default_rng(4300)
    host_in = [torch.from_numpy(rng.random((B, N, 3), dtype=np.float32)).pin_memory() for _ in range(2 * depth + 1)]
    with torch.no_grad():
        serial = [m(h.to(dev), fetch=("globaldesc",))["globaldesc"].cpu() for h in host_in]
        pipe = m.pipeline(host_in[0].to(dev), depth=depth, outputs=("globaldesc",))
        host_out = [torch.empty((B, 256), dtype=torch.float32).pin_memory() for _ in range(depth)]
        tickets, got = [], []
        for i, h in enumerate(host_in):
            if len(tickets) == depth:
                tk = tickets.pop(0)
                tk.event.synchronize()
                got.append(host_out[tk.slot].clone())
            k = pipe.next_slot
            tickets.append(pipe.submit(h, fetch_to={"globaldesc": host_out[k]}))
        for tk in tickets:
            tk.event.synchronize()
            got.append(host_out[tk.slot].clone())
    assert len(got) == len(serial)
    for i, (g, s) in enumerate(zip(got, serial)):
        # (the global tail sums A'^T c and the split-K projection with f32 atomics: run-to-run rounding, not bit-equal)
        assert float((g - s).abs().max()) <= 2e-6, (i, (g - s).abs().max().item())
