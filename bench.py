#!/usr/bin/env python
"""Benchmark of the DH3D hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload local|global|cfg5|train] [--scaling weak|strong]

A "step" is one pass of the hot path over one batch of synthetic clouds already resident in HBM:
  local  (default, BASELINE config[1]): local-descriptor forward, basic_config, N=8192 K=8, batch 8
  global (BASELINE config[2])         : global-descriptor forward, global_config, N=4096, batch 32
  cfg5   (BASELINE config[4])         : save_all dense feature map, detection_config, N=16384, batch 4
                                        (localdesc_extract.py:65,146,166) + the flex_conv 128->128 K=12 kernel line
  train  (BASELINE config[3])         : Siamese quadruplet training step, 1+2+18+1 clouds of N=4096
One process per GPU over RCCL.  `--gpus N` with no torchrun environment re-launches itself under
`python -m torch.distributed.run --nproc-per-node N` (127.0.0.1 rendezvous); under torchrun WORLD_SIZE must equal N.
  --scaling weak   (default): the batch above PER GPU, clouds are independent -> no data-path collective;
  --scaling strong          : the batch above IN TOTAL, ceil(B/N) clouds per GPU (SURVEY 8e "Expected scaling");
  train is always strong (one role-ordered Siamese batch sharded over the ranks + descriptor all-gather).
The timed region is bracketed by barrier + synchronize, the max over ranks is taken, rank 0 prints ONE JSON line.
The forward step is a hipGraph replay of dh3d_amd.model.DH3D.forward.

Steps in flight.  One forward of this path is a latency chain on a few CUs (farthest point sampling: one CU per cloud
for two thirds of the local step) followed by chip-wide kernels, so an engine that extracts descriptors for a stream of
batches keeps SEVERAL steps in flight (local and cfg5: four, global: three): one graph instance per step in flight, each on
its own stream with its own batch buffers; step i is one full pass over one batch on stream i % depth, and the K timed
steps include the pipeline's fill and drain.  That is `value` since round 3 (`config.steps_in_flight`).  `one_step_at_a_time` in the same line is the number rounds 1-2 reported as
`value` (each step finishes before the next one starts); `--inflight 1` makes it `value` again.  The training step
depends on the previous step's weights and always runs one at a time.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec (MI355X_MICROARCH.md)
F32_MFMA_PEAK_TF = 157.3  # dense f32 MFMA = f32 vector peak
BF16_MFMA_PEAK_TF = 2500.0

WORKLOADS = {
    "local": dict(preset="basic_config", B=8, N=8192, seed=2002, out="xyz_feat", inflight=4,
                  name="local-descriptor forward (basic_config), N=8192 K=8, batch=8"),
    "global": dict(preset="global_config", B=32, N=4096, seed=3003, out="globaldesc", inflight=3,
                   name="global-descriptor forward (global_config), N=4096, 64-cluster NetVLAD, batch=32"),
    "cfg5": dict(preset="detection_config", B=4, N=16384, seed=5005, out="xyz_feat_att", inflight=4,
                 name="dense local feature map (save_all path, detection_config), N=16384 K=8, batch=4, device kNN"),
    "train_local": dict(preset="basic_config", B=20, N=8192, seed=6006, out=None, inflight=1,
                        name="stage-1 training step (basic_config: 10 anchors + 10 positives, N=8192, 512 keypoints per cloud, "
                             "desc_local_loss), whole local backbone forward + backward + Adam"),
    "train": dict(preset="global_config", B=22, N=4096, seed=4004, out=None, inflight=1,
                  name="Siamese quadruplet training step, Oxford-shaped batch (1 anchor + 2 pos + 18 neg + 1 other-neg), "
                       "N=4096, frozen backbone, batch sharded over ranks + RCCL all-gather of descriptors"),
}


# --------------------------------------------------------------------------------------------- launch
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def relaunch_under_torchrun(n):
    """`python bench.py --gpus N` without a torchrun environment: become the launcher (one rank per GPU)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.execvpe(cmd[0], cmd, env)


# --------------------------------------------------------------------------------------------- helpers
def build_model(preset, dev, seed=0, num_points=None):
    from dh3d_amd import ConfigFactory
    from dh3d_amd.model import DH3D
    cfg = ConfigFactory(preset).getconfig()
    if num_points:
        cfg.num_points = num_points
    m = DH3D(cfg).init_synthetic(seed)
    return m.to(dev).eval().prepare()


def synthetic_clouds(B, N, seed, dev, rank=0):
    rng = np.random.default_rng(seed + rank)
    return torch.from_numpy(rng.random((B, N, 3), dtype=np.float32)).to(dev)


def scene_like_clouds(B, N, seed, dev):
    """A street scene normalised to [-1, 1] -- 55 % ground plane, 35 % on six walls, 10 % clutter; z extent a tenth of x / y,
    a third of the 16^3 cells occupied -- for the `data_sensitivity` key: the data-dependent kernels (FPS box pruning,
    kNN, three_nn) are timed on the uniform cube everywhere else."""
    rng = np.random.default_rng(seed)
    out = np.empty((B, N, 3), np.float32)
    for b in range(B):
        n_g, n_w = int(N * 0.55), int(N * 0.35)
        g = np.stack([rng.uniform(-1, 1, n_g), rng.uniform(-1, 1, n_g), rng.normal(-0.08, 0.004, n_g)], 1)
        walls = []
        for _ in range(6):
            m = n_w // 6
            x0, y0, ang, ln = rng.uniform(-0.8, 0.8), rng.uniform(-0.8, 0.8), rng.uniform(0, np.pi), rng.uniform(0.3, 0.9)
            t = rng.uniform(0, ln, m)
            walls.append(np.stack([x0 + t * np.cos(ang), y0 + t * np.sin(ang), rng.uniform(-0.08, 0.12, m)], 1)
                         + rng.normal(0, 0.003, (m, 3)))
        w = np.concatenate(walls)
        c = rng.uniform(-1, 1, (N - n_g - len(w), 3)) * np.array([1, 1, 0.1])
        out[b] = np.clip(np.concatenate([g, w, c])[rng.permutation(N)], -1, 1)
    return torch.from_numpy(out).to(dev)


def real_oxford_clouds(B, N, dev):
    """B clouds of N points from the reference's own demo inputs (tests/golden/demo_clouds.npz: 268.bin / 642.bin, Oxford
    LiDAR sub-maps in metres): the N points nearest to the centroid in a seeded random order, as Global_test_dataset /
    get_fixednum_pcd crop them (core/utils.py:92-99); N = 16384 takes the clouds whole.  None if the fixture is missing."""
    path = os.path.join(ROOT, "tests", "golden", "demo_clouds.npz")
    if not os.path.isfile(path):
        return None
    from dh3d_amd.utils import get_fixednum_pcd
    z = np.load(path)
    src = [z["local_268"], z["local_642"]]
    out = np.stack([np.ascontiguousarray(get_fixednum_pcd(src[b % 2], N, rng=np.random.default_rng(1000 + b))[0], np.float32)
                    for b in range(B)])
    return torch.from_numpy(out).to(dev)


def time_steps(run, pts, steps, warmup, dev):
    from dh3d_amd import dist as D
    # clock ramp: a fresh process finds the GPU in a low power state and a millisecond-scale step does not pull it
    # up within a handful of warmup steps (measured: the same graph at 0.97 vs 0.91 ms).  ~0.2 s of untimed
    # replays first, then the W warmup steps the contract asks for, then exactly K timed steps.
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < 0.2:
        run(pts)
        torch.cuda.synchronize(dev)
    for _ in range(warmup):
        run(pts)
    torch.cuda.synchronize(dev)
    D.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        run(pts)
    torch.cuda.synchronize(dev)
    D.barrier()
    dt = time.perf_counter() - t0
    return D.max_over_ranks(dt, dev)


def repeat_blocks(run, pts, steps, dev, repeats):
    """`repeats` more blocks of `steps` timed steps (same barrier + synchronize bracketing): the spread of the
    contract number.  K steps of this path are ~10 ms of timed region, run-to-run +-1.5 %; effects below that are only
    visible in the median / minimum over several blocks.  Informational -- `value` is the first block alone."""
    from dh3d_amd import dist as D
    out = []
    for _ in range(repeats):
        torch.cuda.synchronize(dev)
        D.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            run(pts)
        torch.cuda.synchronize(dev)
        D.barrier()
        out.append(D.max_over_ranks(time.perf_counter() - t0, dev) / steps * 1e3)
    return out


def event_time_ms(fn, iters=50, warm=5):
    """Average duration of fn() (kernels on torch's current stream) from HIP events on that stream."""
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


# --------------------------------------------------------------------------------------------- roofline
def flex_figures(B, N, K, Din, Dout):
    """Algorithmic figures per launch:
      Bc = 4*[B*N*(Din+Dout+3+K) + 4*Din*Dout]   compulsory HBM bytes
      Bg = 4*B*N*[K*(Din+4)+3+Dout]              bytes requested by the gather (cache hierarchy)
      F  = 2*B*N*4*Din*(K+Dout)                  flops of the factorised form (gather-reduce + GEMM)"""
    return (4.0 * (B * N * (Din + Dout + 3 + K) + 4 * Din * Dout), 4.0 * B * N * (K * (Din + 4) + 3 + Dout),
            2.0 * B * N * 4 * Din * (K + Dout))


def _flex_inputs(dev, B, N, K, Din, Dout):
    from dh3d_amd import pm
    g = torch.Generator(device="cpu").manual_seed(1)
    xyz = torch.rand(B, N, 3, generator=g).to(dev)
    f = torch.randn(B, N, Din, generator=g).to(dev)
    nbr, _ = pm.knn_xyz(xyz, K)
    theta = (torch.randn(3, Din, Dout, generator=g) / Din ** 0.5).to(dev)
    bias = (torch.randn(Din, Dout, generator=g) / (8 * Din) ** 0.5).to(dev)
    return xyz, f, nbr, theta, bias


def three_fractions(ms, B, N, K, Din, Dout):
    Bc, Bg, F = flex_figures(B, N, K, Din, Dout)
    t = ms * 1e-3
    return {"launch_ms": ms, "algorithmic_bytes": Bc,
            "strict_hbm": {"achieved": Bc / t / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": Bc / t / 1e9 / HBM_PEAK_GBS},
            "gather_effective": {"achieved": Bg / t / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": Bg / t / 1e9 / HBM_PEAK_GBS, "bytes": Bg},
            "f32_equivalent_flops": {"achieved": F / t / 1e12, "peak": F32_MFMA_PEAK_TF, "unit": "TFLOP/s",
                                     "frac": F / t / 1e12 / F32_MFMA_PEAK_TF, "flops": F}}


def live_pmc_traffic(timeout=240):
    """HBM bytes per launch of the roofline kernel from rocprofv3 PMC counters, collected NOW on this box: two separate
    --pmc passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace only, the guide's recipe) over tools/flex_conv_pmc.py, the
    FETCH correction calibrated on a 64 MiB device copy in the same run (gfx950 tallies wide streaming reads at 1/2).
    Returns (bytes or None, source string)."""
    import shutil
    import sqlite3
    if not shutil.which("rocprofv3"):
        return None, "rocprofv3 not on PATH"
    vals, cal = {}, {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = tempfile.mkdtemp(prefix="dh3d_pmc_", dir="/tmp")
            env = dict(os.environ, PYTHONPATH=ROOT, TMPDIR="/tmp")
            for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
                env.pop(k, None)
            subprocess.run(["rocprofv3", "--pmc", counter, "--kernel-trace", "-d", d, "-o", "p", "--",
                            sys.executable, os.path.join(ROOT, "tools", "flex_conv_pmc.py")],
                           cwd="/tmp", env=env, timeout=timeout, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            db = [os.path.join(r, f) for r, _, fs in os.walk(d) for f in fs if f.endswith(".db")]
            if not db:
                return None, "rocprofv3 --pmc %s produced no database" % counter
            c = sqlite3.connect(db[0])
            rows = c.execute("select name, avg(counter_value), max(counter_value) from pmc_events where counter_name=? "
                             "group by name", (counter,)).fetchall()
            for name, avg, mx in rows:
                if "flex_conv_x6_kernel" in name:
                    vals[counter] = avg
                if "copyBuffer" in name:
                    cal[counter] = mx
            shutil.rmtree(d, ignore_errors=True)
        if "FETCH_SIZE" not in vals or "WRITE_SIZE" not in vals:
            return None, "kernel not found in the PMC tables"
        fcorr = 65536.0 / cal["FETCH_SIZE"] if cal.get("FETCH_SIZE") else 2.0
        wcorr = 65536.0 / cal["WRITE_SIZE"] if cal.get("WRITE_SIZE") else 1.0
        if not (1.8 < fcorr < 2.2):
            fcorr = 2.0
        if not (0.9 < wcorr < 1.1):
            wcorr = 1.0
        total = (vals["FETCH_SIZE"] * fcorr + vals["WRITE_SIZE"] * wcorr) * 1024.0
        return total, ("live: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace) in this "
                       "run; raw %.1f KB x %.2f + %.1f KB x %.2f (corrections calibrated on a 64 MiB copy)"
                       % (vals["FETCH_SIZE"], fcorr, vals["WRITE_SIZE"], wcorr))
    except Exception as e:  # noqa: BLE001  (measurement aid: never fail the bench line)
        return None, "live PMC failed: %r" % (e,)


def flex_conv_roofline(dev, in_step_ms=None, pmc=True, B=8, N=8192, K=8, Din=64, Dout=64):
    """The kernel BASELINE.json names: flex_conv at N=8192, K=8 (stage-1 layer 64->64, batch 8)."""
    from dh3d_amd import pm
    xyz, f, nbr, theta, bias = _flex_inputs(dev, B, N, K, Din, Dout)
    wp = pm.pack_flex_weight(theta, bias)
    wp3 = pm.pack_flex_weight_x3(theta, bias)
    fb = torch.zeros(Dout, device=dev)
    # the kernel the model runs at this shape: the persistent bf16x6 pipeline (csrc/flex_x6.hip); the exact-f32
    # MFMA kernel (csrc/flex_pm.hip, used for the other shapes) is timed beside it
    ms = event_time_ms(lambda: pm.flex_conv_x6(f, xyz, nbr, wp3, Dout, pre_bias=fb, scale=fb + 1, shift=fb,
                                               act=pm.ACT_RELU))
    ms_f32 = event_time_ms(lambda: pm.flex_conv(f, xyz, nbr, wp, Dout, pre_bias=fb, scale=fb + 1, shift=fb,
                                                act=pm.ACT_RELU))
    fr = three_fractions(ms, B, N, K, Din, Dout)
    traffic, src = (None, "skipped (--no-pmc)")
    if pmc:
        traffic, src = live_pmc_traffic()
    if traffic is None:
        static = os.path.join(ROOT, "profiles", "pmc_flex_conv.json")
        if os.path.isfile(static):
            rec = json.load(open(static))
            traffic, src = rec["hbm_bytes_per_launch"], "STATIC (%s); live attempt: %s" % (rec["source"], src)
    out = {
        "bound": "hbm", "kernel": "flex_conv_x6_kernel<%d,%d> B=%d N=%d K=%d" % (Din, Dout, B, N, K),
        "achieved": fr["strict_hbm"]["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": fr["strict_hbm"]["frac"],
        "traffic": traffic, "traffic_source": src, "launch_ms": ms, "algorithmic_bytes": fr["algorithmic_bytes"],
        "launch_ms_f32_mfma_kernel": ms_f32,
        "gather_effective": fr["gather_effective"], "f32_equivalent_flops": fr["f32_equivalent_flops"],
        "binding_roof": "VALU issue port (one instruction per ~4.7 cycles per SIMD) + matrix pipe (DESIGN.md 3.5)",
    }
    if in_step_ms is not None:
        Bc = fr["algorithmic_bytes"]
        out["in_step"] = {"launch_ms": in_step_ms, "frac": Bc / (in_step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                          "note": "the same kernel timed with HIP events on its own stream inside the local forward "
                                  "(FPS running beside it on the other stream)"}
    return out


def global_tail_roofline(dev, B=32, N=4096):
    """MFMA figures of the global path's head (core/backbones.py:156-173,202-279) at config 3's
    shape, timed with HIP events on isolated launches:
      * `netvlad_block`: the NetVLAD + gating block on a MATERIALISED [B,N,256] map (csrc/netvlad.hip: assignment GEMM,
        BN + softmax x attention, VLAD contraction, hidden split-K projection, gate) -- the block as BASELINE.md counts
        it: 4*B*N*256*64 + 2*B*16384*256 + 2*B*256^2 flop against the 157.3 TF f32-MFMA peak;
      * `fused_tail`: what the model runs at this shape (pm.global_tail): the attention conv 256->1024 and NetVLAD's
        assignment both commuted through the three_interpolate up-sampling -- GEMMs on the N/8 coarse rows (bf16x6:
        f32-accurate on the bf16 pipe), one walk over the fine points, `A'^T c`, the same hidden / gate tail.  Its
        reference-formulation flops include the attention conv on the fine rows (2*B*N*256*1024) that the commuted
        form never executes; `flops_executed` counts what the kernels do (bf16x6 products counted once, as f32 flops).
    PMC counters of the same kernels (MFMA busy cycles, L2 requests / hits of the walk): profiles/r03_c_pmc_global_tail.txt."""
    from dh3d_amd import pm
    model = build_model("global_config", dev, seed=0, num_points=N)
    pts = synthetic_clouds(B, N, 3003, dev, 0)
    out = {}
    with torch.no_grad():
        model(pts, fetch=("globaldesc",))
        geo = model._geometry(pts, None)
        model._join_side(geo)
        _, local = model.compute_local(pts, _geo=geo)
        lv = geo.level(8, model.knn_num)
        gba, ga, nv = model.global_before_assemble, model.globalatt, model._netvlad
        coarse = gba(geo, local, coarse_only=True)
        last = ga.detec_conv0
        lp, gp, p = last._prep, ga._prep or ga.prepare(), nv._prep or nv.prepare()
        torch.cuda.synchronize(dev)
        m = coarse.shape[1]
        f_att_ref = 2.0 * B * N * 256 * 1024
        f_vlad = 4.0 * B * N * 256 * 64 + 2.0 * B * 16384 * 256 + 2.0 * B * 256 * 256
        if "wslices" in gp and "_ordered" in lv:
            ms = event_time_ms(lambda: pm.global_tail(
                coarse, lv["nn3_idx"], lv["nn3_dist"], lv["_ordered"][0], gp["wslices"], last.cout, gp["w_fc"], gp["b_fc"],
                (lp["b"], lp["scale"], lp["shift"], pm.ACT_RELU), p["wc"], p["cs"], p["ch"], p["W2"], p["Wh"], p["s1"],
                p["h1"], p["Wg"], p["s2"], p["h2"], l2_eps=1e-8), iters=20, warm=3)
            # executed: GEMMs on the coarse rows (attention 256->1024, cluster logits 256->64), interpolation of the 1024
            # + 64 wide rows at the fine points (3 fma each), slot-matrix scatter product per 128-point block,
            # A'^T c, hidden projection, gate
            f_exec = (2.0 * B * m * 256 * (1024 + 64) + 2.0 * B * N * 3 * (1024 + 64) + 2.0 * B * N * 64 * 64
                      + 2.0 * B * 64 * m * 256 + 2.0 * B * 16384 * 256 + 2.0 * B * 256 * 256)
            t = ms * 1e-3
            out["fused_tail"] = {
                "kernels": "linear_x6 slices (attention GEMM on coarse rows) + interp_head_lds_kernel<true> (walk) + "
                           "gemm_x6 (A'^T c) + netvlad_finalize / hidden_splitk / gate",
                "launch_ms": ms, "flops_reference": f_att_ref + f_vlad, "flops_executed": f_exec,
                "reference_flops_rate": {"achieved": (f_att_ref + f_vlad) / t / 1e12, "unit": "TFLOP/s",
                                         "note": "the REFERENCE formulation's flops over this time: informational, no fraction -- "
                                                 "the commuted form never executes the attention conv on the fine rows"},
                "executed_flops_rate": {"achieved": f_exec / t / 1e12, "peak": F32_MFMA_PEAK_TF, "unit": "TFLOP/s",
                                        "frac": f_exec / t / 1e12 / F32_MFMA_PEAK_TF},
                "binding_roof": "L2 gather of the walk (12 KB of coarse rows per fine point) + VALU; the GEMMs are 1/8 of "
                                "the reference's flops"}
        # the literal block on materialised rows
        d = torch.clamp(lv["nn3_dist"], min=1e-10)
        w = (1.0 / d) / (1.0 / d).sum(2, keepdim=True)
        from dh3d_amd import ops
        forglobal = ops.three_interpolate(coarse, lv["nn3_idx"], w.contiguous())
        att = torch.rand(B, N, 1, device=dev)
        ms2 = event_time_ms(lambda: nv(forglobal, att, l2_eps=1e-8), iters=20, warm=3)
        t2 = ms2 * 1e-3
        out["netvlad_block"] = {
            "kernels": "netvlad_assign_accumulate + finalize + l2scale + hidden_splitk + gate (csrc/netvlad.hip) on the "
                       "materialised [B,N,256] map",
            "launch_ms": ms2, "flops": f_vlad, "bound": "mfma",
            "achieved": f_vlad / t2 / 1e12, "peak": F32_MFMA_PEAK_TF, "unit": "TFLOP/s",
            "frac": f_vlad / t2 / 1e12 / F32_MFMA_PEAK_TF,
            "hbm": {"bytes": 4.0 * B * N * 257 + 4.0 * 16384 * 256, "achieved": (4.0 * B * N * 257 + 4.0 * 16384 * 256) / t2 / 1e9,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": (4.0 * B * N * 257 + 4.0 * 16384 * 256) / t2 / 1e9 / HBM_PEAK_GBS},
            "note": "8.9 GF over 151 MB of compulsory bytes = 58 flop/B, three times ABOVE the f32-MFMA ridge (157.3 TF / 8 TB/s = "
                    "19.7 flop/B): the block is MFMA-bound; the HBM fraction is given beside it"}
    out["workload"] = "global_config, B=%d, N=%d, 64-cluster NetVLAD" % (B, N)
    return out


def step_kernel_rows(workload):
    """The kernels of one step of `workload` with their ALGORITHMIC work (SURVEY 8d's per-op figures at the bench shape):
    (regex on the kernel name, launches per step, what it is, bytes, flops, bound).  Bytes are the compulsory HBM bytes
    of the operator as executed (inputs once + outputs once; for the commuted global tail the intermediates that do
    travel through memory are counted once written + once read), flops the factorised / executed count in f32 flops
    (bf16x6 products counted once)."""
    wl = WORKLOADS[workload]
    B, N, K = wl["B"], wl["N"], 8
    M = N // 8
    R, Rs = float(B * N), float(B * M)
    ff = flex_figures
    rows = [
        ("spatial_sort_kernel<%d>" % (N // 1024), 1, "Morton sort of the clouds", 4 * R * 7 + 32 * R / 64, 0.0, "latency (one workgroup per cloud)"),
        ("fps_list_kernel", 1, "farthest point sampling N -> N/8", 16 * R + 16 * Rs, 0.0, "latency (N/8 dependent picks, one CU per cloud; the box lists prune most of the 8*B*N*N/8 brute-force flops: executed work not counted, floor = bytes only)"),
        ("knn_split_kernel", 1, "the pruned scan's launch behind the cell lists: serves the clouds the sort flags as crowded -- none of the "
                                "uniform bench input, every workgroup leaves at its first instruction (no work, no floor)", 0.0, 0.0, "launch"),
        ("knn_grid_kernel", 1, "kNN K=8 on the full clouds (cell lists on the sort's grid)", 16 * R + 8 * R * K, "knn_grid_executed", "f32 VALU on the candidates VISITED (lower bound: every point of the 3x3x3 cell block around the query, 8 flop each) -- not the 8*B*N*N brute-force count"),
        ("pointset_sum_kernel", 1, "conv_pointset: neighbour-offset sums", 4 * R * (3 + K + 4), 6.0 * R * K, "hbm"),
        ("pointset_pool_kernel", 1, "conv_pointset 3->32 + BNReLU + flex_pool", 4 * R * (4 + K + 32), 2.0 * R * K * 32 * 3, "hbm"),
        ("conv_pointset_pm_kernel", 1, "conv_pointset 3->32 + BNReLU", 4 * R * (3 + K + 32), 2.0 * R * K * 32 * 3, "hbm"),
        ("flex_pool_pm_kernel", 1, "flex_pool D=32", 4 * R * (2 * 32 + K), 0.0, "hbm"),
        ("flex_conv_x6_kernel<32, 64", 1, "flex_conv 32->64 @N", ff(B, N, K, 32, 64)[0], ff(B, N, K, 32, 64)[2], "hbm (contract) / matrix pipe"),
        ("flex_conv_x6_kernel<64, 64", 1, "flex_conv 64->64 @N", ff(B, N, K, 64, 64)[0], ff(B, N, K, 64, 64)[2], "hbm (contract) / matrix pipe"),
        ("se_res_mfma_kernel<64, true, true", 1, "flex_pool + SE + residual + 1x1 conv 64->64 @N", 4 * R * (64 + K + 64 + 64), 2.0 * R * (2 * 64 * 16 + 64 * 64), "hbm"),
        ("knn_small_kernel", 1, "kNN K=8 on the sampled sets", 12 * Rs + 8 * Rs * K, 0.0, "latency / VALU issue (pruned scan: executed work not counted, floor = bytes only)"),
        ("spatial_sort_kernel<1>", 1, "Morton sort of the sampled sets", 4 * Rs * 7 + 32 * Rs / 64, 0.0, "latency"),
        ("three_nn_pruned_kernel", 1, "three_nn N vs N/8", 16 * R + 16 * Rs + 24 * R, 0.0, "latency / VALU issue (pruned scan: executed work not counted, floor = bytes only)"),
        ("flex_conv_tx6_kernel<64, 128", 1, "flex_conv 64->128 @N/8 (32-point tiles, bf16x6 tile GEMM)", ff(B, M, K, 64, 128)[0], ff(B, M, K, 64, 128)[2], "gather + matrix pipe / L2 (weights)"),
        ("flex_conv_tx6_kernel<128, 128", 1, "flex_conv 128->128 @N/8 (32-point tiles, bf16x6 tile GEMM)", ff(B, M, K, 128, 128)[0], ff(B, M, K, 128, 128)[2], "gather + matrix pipe / L2 (weights)"),
        ("flex_conv_pm_kernel<64, 128", 1, "flex_conv 64->128 @N/8 (exact-f32 tiles: rounds 1-3)", ff(B, M, K, 64, 128)[0], ff(B, M, K, 64, 128)[2], "f32 MFMA"),
        ("flex_conv_pm_kernel<128, 128", 1, "flex_conv 128->128 @N/8 (exact-f32 tiles: rounds 1-3)", ff(B, M, K, 128, 128)[0], ff(B, M, K, 128, 128)[2], "f32 MFMA"),
        ("se_res_mfma_kernel<128, true, false", 1, "flex_pool + SE + residual @N/8", 4 * Rs * (128 + K + 128), 2.0 * Rs * 2 * 128 * 32, "hbm"),
        ("se_res_mfma_kernel<128, true, true", 1, "flex_pool + SE + residual + concat conv's upper block 128->128 @N/8",
         4 * Rs * (128 + K + 128 + 128), 2.0 * Rs * (2 * 128 * 32 + 128 * 128), "hbm"),
    ]
    if workload == "global":
        rows += [
            ("group_point_fwd4_kernel", 2, "group_point of the sampled rows (C=64, C=128)", 4 * Rs * (2 + 2 * 64 + 2 * 128), 0.0, "hbm"),
            ("linear_x6_kernel<1, true>", 1, "concat conv [interp(c)|x2] + shortcut conv -> 128 @N (fused up-sampling; rounds 1-4)",
             4 * R * (64 + 64 + 6 + 128) + 4 * Rs * 128, 2.0 * R * 256 * 128, "hbm / matrix pipe"),
            ("local_tail_fused_kernel", 1, "the local-feature tail in one launch: shortcut conv + concat conv's lower block + "
                                           "up-sampling of the commuted upper block + BN/ReLU + sum -> 128 @N",
             4 * R * (64 + 64 + 6 + 128) + 4 * Rs * 128, 2.0 * R * 2 * 64 * 128, "hbm / matrix pipe"),
            ("flex_conv_tx6_kernel<128, 256", 1, "flex_conv 128->256 @N/8 + cluster logits (global; bf16x6 tile GEMM)", ff(B, M, K, 128, 256)[0], ff(B, M, K, 128, 256)[2], "gather + matrix pipe / L2 (weights)"),
            ("flex_conv_pm_kernel<128, 256", 1, "flex_conv 128->256 @N/8 (global; exact-f32 tiles: rounds 1-3)", ff(B, M, K, 128, 256)[0], ff(B, M, K, 128, 256)[2], "f32 MFMA"),
            ("linear_x6_kernel<2, false>", 1, "attention conv 256->1024 on the coarse rows (commuted)", 4 * Rs * (256 + 1024), 2.0 * Rs * 256 * 1024, "matrix pipe"),
            ("linear_pm_kernel<1>", 1, "cluster logits 256->64 on the coarse rows", 4 * Rs * (256 + 64), 2.0 * Rs * 256 * 64, "hbm"),
            ("fillBufferAligned", 1, "zero fill of the walk's accumulators", 4.0 * B * M * 64, 0.0, "hbm"),
            ("interp_head_lds_kernel", 1, "walk: interpolated attention logit + NetVLAD assignment -> A'",
             4 * Rs * (1024 + 64) + 4 * R * (6 + 4) + 4.0 * B * M * 64, 2.0 * R * 3 * (1024 + 64) + 2.0 * R * 1024 + 2.0 * R * 64 * 64, "L2 gather + VALU"),
            ("netvlad_assign_finalize", 1, "VLAD = A'^T c on the coarse rows + residual + intra-normalise",
             4.0 * B * (M * 64 + M * 256 + 64 * 256), 2.0 * B * 64 * M * 256, "f32 VALU / L2"),
            ("gemm_x6_kernel", 1, "VLAD = A'^T c (rounds 2-3: its own launch)", 4.0 * B * (M * 64 + M * 256 + 64 * 256), 2.0 * B * 64 * M * 256, "matrix pipe"),
            ("netvlad_finalize", 1, "VLAD residual + intra-normalise (rounds 2-3)", 4.0 * B * 64 * 256 * 2, 0.0, "hbm"),
            ("netvlad_hidden_splitk", 1, "hidden projection 16384->256", 4.0 * 16384 * 256 + 4.0 * B * 16384, 2.0 * B * 16384 * 256, "hbm (16.8 MB of weights)"),
            ("netvlad_gate", 1, "BN + context gating", 4.0 * 256 * 256 + 4.0 * B * 512, 2.0 * B * 256 * 256, "latency"),
        ]
    else:
        rows += [
            ("group_point_fwd4_kernel", 1, "group_point of the sampled rows (C=64)", 4 * Rs * (1 + 2 * 64), 0.0, "hbm"),
            ("linear_k64_x6_kernel", 2, "shortcut conv 64->128 and concat conv's lower block 64->128 @N", 2 * 4 * R * (64 + 128), 2 * 2.0 * R * 64 * 128, "hbm"),
            ("linear_x6_kernel<1, false>", 2, "the same two convs on the long-K kernel (rounds 1-3)", 2 * 4 * R * (64 + 128), 2 * 2.0 * R * 64 * 128, "hbm"),
            ("linear_x6_kernel<2, false>", 1, "shortcut + lower block in one launch @N", 4 * R * (64 + 64 + 256), 2 * 2.0 * R * 64 * 128, "hbm"),
            ("linear_pm_kernel<2>", 1, "concat conv's upper block 128->128 on the coarse rows", 4 * Rs * 256, 2.0 * Rs * 128 * 128, "hbm"),
            ("interp_combine_kernel", 1, "up-sampling + bias/BN/ReLU + shortcut + l2-normalise/concat store @N",
             4 * R * (128 + 128 + 6 + 131) + 4 * Rs * 128, 2.0 * R * 3 * 128, "hbm"),
            ("local_tail_fused_kernel", 1, "(DH3D_TAIL_FUSED=1) the three kernels above in one launch",
             4 * R * (64 + 64 + 6 + 131) + 4 * Rs * 128, 2.0 * R * 2 * 64 * 128, "hbm / matrix pipe"),
        ]
    return rows


def knn_grid_executed_flops(workload, dev):
    """Lower bound of what knn_grid_kernel executes on the bench input: 8 flop (3 sub, 3 fma-class, compare/insert not
    counted) for every point of the 3 x 3 x 3 block of grid cells around each query -- the first two shells, which the
    kernel always pools (csrc/knn.hip) -- from the sort's own cell table.  The brute-force count 8*B*N*N is NOT executed."""
    from dh3d_amd import pm
    wl = WORKLOADS[workload]
    pts = synthetic_clouds(wl["B"], wl["N"], wl["seed"], dev, 0)
    _, _, cells = pm.spatial_sort_cells(pts)
    cells_h = cells.cpu().numpy()
    ct = cells_h[:, :4097].astype(np.int64)
    cnt = np.diff(ct, axis=1)                                   # points per grid cell, cells in code order [B, 4096]
    code = np.arange(4096)
    total = 0.0
    for b in range(cnt.shape[0]):
        # per-axis cell coordinates from the cloud's own bit schedule (include/dh3d_hip.h, cells[4107]: field s = the axis
        # of code bit 11 - s; a cube's 0x186186 is the plain z-y-x Morton code, any other cloud deals its bits by extent)
        sched = int(cells_h[b, 4107]) & 0xFFFFFF
        coord = [np.zeros(4096, np.int64) for _ in range(3)]
        for st in range(12):
            a = (sched >> (2 * st)) & 3
            coord[a] = (coord[a] << 1) | ((code >> (11 - st)) & 1)
        dims = [int(c.max()) + 1 for c in coord]
        g = np.zeros((dims[0] + 2, dims[1] + 2, dims[2] + 2))
        g[coord[0] + 1, coord[1] + 1, coord[2] + 1] = cnt[b]
        box = sum(g[1 + dx:1 + dims[0] + dx, 1 + dy:1 + dims[1] + dy, 1 + dz:1 + dims[2] + dz]
                  for dx in (-1, 0, 1) for dy in (-1, 0, 1) for dz in (-1, 0, 1))
        total += float((g[1:1 + dims[0], 1:1 + dims[1], 1:1 + dims[2]] * box).sum())
    return 8.0 * total, total / (wl["B"] * wl["N"])


def step_roofline(workload, serial_ms=None, in_flight_ms=None, depth=None, timeout=240, dev=None):
    """Step-level roofline: every kernel of one step -- algorithmic bytes / flops (step_kernel_rows), its floor at 8 TB/s
    and 157.3 TF (f32 MFMA = f32 VALU peak), its stand-alone duration measured NOW by `rocprofv3 --kernel-trace` over
    tools/step_forward.py (the eager forward on ONE stream: no kernel overlaps another) -- and how far the whole step is
    from the sum of the floors, one step at a time and in flight."""
    import re
    import shutil
    import sqlite3
    out = {"workload": WORKLOADS[workload]["name"], "peaks": {"hbm_GBps": HBM_PEAK_GBS, "f32_TFps": F32_MFMA_PEAK_TF}}
    if not shutil.which("rocprofv3"):
        out["error"] = "rocprofv3 not on PATH"
        return out
    d = tempfile.mkdtemp(prefix="dh3d_step_", dir="/tmp")
    try:
        env = dict(os.environ, PYTHONPATH=ROOT, TMPDIR="/tmp")
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        subprocess.run(["rocprofv3", "--kernel-trace", "-d", d, "-o", "p", "--", sys.executable,
                        os.path.join(ROOT, "tools", "step_forward.py"), workload, "6"],
                       cwd="/tmp", env=env, timeout=timeout, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        db = [os.path.join(r, f) for r, _, fs in os.walk(d) for f in fs if f.endswith(".db")]
        if not db:
            out["error"] = "rocprofv3 produced no database"
            return out
        c = sqlite3.connect(db[0])
        cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
        namecol = "name" if "name" in cols else "kernel_name"
        disp = c.execute("select %s, start, end from kernels order by start" % namecol).fetchall()
    except Exception as e:  # noqa: BLE001 -- measurement aid: never fail the bench line
        out["error"] = "kernel trace failed: %r" % (e,)
        return out
    finally:
        shutil.rmtree(d, ignore_errors=True)
    # the LAST forward of the trace (everything after the last dispatch of the step's first kernel)
    first = "spatial_sort_kernel"
    starts = [i for i, (n, _, _) in enumerate(disp) if first in n and "<1>" not in n]
    last = disp[starts[-1]:] if starts else disp
    rows, tot_floor, tot_meas, listed = [], 0.0, 0.0, 0
    for pat, calls, what, nbytes, flops, bound in step_kernel_rows(workload):
        durs = [(e - s) / 1e3 for n, s, e in last if pat in n]
        if not durs:
            continue
        extra = {}
        if flops == "knn_grid_executed":
            try:
                flops, per_query = knn_grid_executed_flops(workload, dev or torch.device("cuda"))
                extra = {"candidates_per_query_lower_bound": round(per_query, 1),
                         "bruteforce_flops_not_executed": 8.0 * WORKLOADS[workload]["B"] * WORKLOADS[workload]["N"] ** 2}
            except Exception as e:  # noqa: BLE001
                flops, extra = 0.0, {"executed_flops_error": repr(e)[:120]}
        listed += len(durs)
        meas = sum(durs)
        f_hbm, f_fl = nbytes / (HBM_PEAK_GBS * 1e9) * 1e6, flops / (F32_MFMA_PEAK_TF * 1e12) * 1e6
        floor = max(f_hbm, f_fl)
        tot_floor += floor
        tot_meas += meas
        rows.append({"kernel": pat, "launches": len(durs), "what": what, "bytes": nbytes, "flops": flops, "bound": bound,
                     "floor_us": round(floor, 2), "floor_hbm_us": round(f_hbm, 2), "floor_f32_us": round(f_fl, 2),
                     "measured_us": round(meas, 2), "frac": round(floor / meas, 4) if meas > 0 and floor > 0 else None})
        rows[-1].update(extra)
    other = [(n, (e - s) / 1e3) for n, s, e in last if not any(r["kernel"] in n for r in rows)]
    out["kernels"] = rows
    out["unlisted_kernels_us"] = round(sum(t for _, t in other), 2)
    out["unlisted_kernel_names"] = sorted({re.sub(r"\(.*", "", n)[:60] for n, _ in other})[:8]
    out["sum_of_floors_ms"] = tot_floor / 1e3
    out["sum_of_measured_kernel_ms"] = tot_meas / 1e3
    out["frac_of_kernel_time"] = tot_floor / tot_meas if tot_meas else None
    if serial_ms:
        out["one_step_at_a_time"] = {"ms_per_step": serial_ms, "frac_step": tot_floor / 1e3 / serial_ms}
    if in_flight_ms:
        out["in_flight"] = {"steps_in_flight": depth, "ms_per_step": in_flight_ms, "frac_step": tot_floor / 1e3 / in_flight_ms}
    out["note"] = ("floor_us = max(bytes / 8 TB/s, EXECUTED flops / 157.3 TF) per kernel -- the pruned searches (FPS box lists, "
                   "kNN of the sampled sets, three_nn) carry a bytes-only floor, knn_grid a lower bound of the candidates it "
                   "visits: no fraction is taken against brute-force pairs that are never computed; FPS and the sorts are latency chains on "
                   "one CU per cloud -- their floors assume the whole chip and are never reached by one step, which is "
                   "why steps run in flight; PMC utilisation of the same forward: profiles/r04_*_pmc_step_*.txt")
    return out


def flex_in_step_ms(dev, reps=10):
    """Duration of the stage-1 flex_conv 64->64 launch INSIDE the local forward (events on the stream it runs on)."""
    from dh3d_amd import pm
    wl = WORKLOADS["local"]
    model = build_model(wl["preset"], dev, seed=0)
    pts = synthetic_clouds(wl["B"], wl["N"], wl["seed"], dev)
    orig = pm.flex_conv_x6
    rec = []

    def timed(features, *a, **kw):
        if features.shape[-1] == 64 and features.shape[1] == wl["N"]:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = orig(features, *a, **kw)
            e1.record()
            rec.append((e0, e1))
            return out
        return orig(features, *a, **kw)
    pm.flex_conv_x6 = timed
    try:
        with torch.no_grad():
            for _ in range(reps + 2):
                model(pts, fetch=(wl["out"],))
        torch.cuda.synchronize(dev)
    finally:
        pm.flex_conv_x6 = orig
    ts = [a.elapsed_time(b) for a, b in rec[2:]]
    return sum(ts) / max(len(ts), 1)


def cfg5_kernel_line(dev):
    """BASELINE config[4]'s kernel: flex_conv 128->128 at B=1, N=16384, K=12 (three fractions)."""
    from dh3d_amd import pm
    B, N, K, Din, Dout = 1, 16384, 12, 128, 128
    xyz, f, nbr, theta, bias = _flex_inputs(dev, B, N, K, Din, Dout)
    wp, wp3 = pm.pack_flex_weight(theta, bias), pm.pack_flex_weight_x3(theta, bias)
    fb = torch.zeros(Dout, device=dev)
    kw = dict(pre_bias=fb, scale=fb + 1, shift=fb, act=pm.ACT_RELU)
    ms = event_time_ms(lambda: pm.flex_conv_tile_x6(f, xyz, nbr, wp3, Dout, **kw))  # what the model launches (backbones.py)
    out = three_fractions(ms, B, N, K, Din, Dout)
    out["kernel"] = "flex_conv_tx6_kernel<128,128,12> B=1 N=16384 K=12 (32-point tiles, bf16x6 tile GEMM)"
    out["exact_f32_kernel_launch_ms"] = event_time_ms(lambda: pm.flex_conv(f, xyz, nbr, wp, Dout, **kw))
    return out


def kernel_breakdown(dev, B, N):
    """Stand-alone event timings of the main kernels at the bench shape (ms per launch)."""
    from dh3d_amd import pm, ops
    xyz = torch.rand(B, N, 3, device=dev)
    out = {}
    out["knn_xyz K=8"] = event_time_ms(lambda: pm.knn_xyz(xyz, 8), iters=10, warm=2)
    out["fps N->N/8"] = event_time_ms(lambda: ops.farthest_point_sample(N // 8, xyz), iters=5, warm=1)
    out["spatial_sort"] = event_time_ms(lambda: pm.spatial_sort(xyz), iters=10, warm=2)
    srt, gbox = pm.spatial_sort(xyz)
    out["knn_sorted K=8"] = event_time_ms(lambda: pm.knn_sorted(srt, gbox, 8), iters=10, warm=2)
    if 4096 <= N <= 16384:
        out["fps_sorted N->N/8"] = event_time_ms(lambda: pm.fps_sorted(srt, gbox, N // 8, xyz=xyz if N > 12288 else None),
                                                 iters=5, warm=1)
    sub = xyz[:, : N // 8].contiguous()
    out["three_nn"] = event_time_ms(lambda: ops.three_nn(xyz, sub), iters=10, warm=2)
    if N // 8 >= 256:
        out["spatial_sort N/8"] = event_time_ms(lambda: pm.spatial_sort(sub), iters=10, warm=2)
        ss, gs = pm.spatial_sort(sub)
        out["three_nn_sorted"] = event_time_ms(lambda: pm.three_nn_sorted(srt, gbox, ss, gs), iters=10, warm=2)
    return out


def dropin_ops_line(dev):
    """The reference-signature operators (section A of the C ABI, channels-first [B,C,N]) at the roofline shape, beside
    the fused point-major kernel: what a TF-side integrator following INTEGRATION.md gets."""
    from dh3d_amd import ops, pm
    B, N, K, Din, Dout = 8, 8192, 8, 64, 64
    xyz, f, nbr, theta, bias = _flex_inputs(dev, B, N, K, Din, Dout)
    f_cf, p_cf = f.transpose(1, 2).contiguous(), xyz.transpose(1, 2).contiguous()
    nbr_cf = nbr.transpose(1, 2).contiguous()
    out = {}
    with torch.no_grad():
        out["ops.flex_convolution 64->64 fwd"] = event_time_ms(
            lambda: ops.flex_convolution(f_cf, p_cf, nbr_cf, theta, bias), iters=10, warm=2)
        out["ops.flex_pooling D=64 fwd"] = event_time_ms(lambda: ops.flex_pooling(f_cf, nbr_cf), iters=10, warm=2)
        # a shape no DH3D layer has (48 -> 96): the factorisation in two launches on the GEMM kernels
        g = torch.Generator(device="cpu").manual_seed(2)
        f48 = torch.randn(B, 48, N, generator=g).to(dev)
        th48, b48 = torch.randn(3, 48, 96, generator=g).to(dev) / 7, torch.randn(48, 96, generator=g).to(dev) / 20
        out["ops.flex_convolution 48->96 fwd (generic channel counts)"] = event_time_ms(
            lambda: ops.flex_convolution(f48, p_cf, nbr_cf, th48, b48), iters=10, warm=2)
        th0, b0 = torch.randn(3, 32, device=dev), torch.randn(32, device=dev)
        out["ops.convolution_pointset 3->32 fwd"] = event_time_ms(
            lambda: ops.convolution_pointset(p_cf, nbr_cf, th0, b0), iters=10, warm=2)
        wp3 = pm.pack_flex_weight_x3(theta, bias)
        out["pm.flex_conv_x6 64->64 fwd"] = event_time_ms(lambda: pm.flex_conv_x6(f, xyz, nbr, wp3, Dout), iters=20)
    return out


# --------------------------------------------------------------------------------------------- CPU baseline
def cpu_model_string():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def usable_cpus():
    """Cores this process may actually run on: the affinity mask, capped by the cgroup's CPU quota (os.cpu_count() is
    the machine's -- round 2 started 256 workers on a pod that may use far fewer and reported it as 256 cores)."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().split()[0])
            break
        except (OSError, ValueError, IndexError):
            continue
    if quota is not None:
        n = max(1, min(n, int(quota)))
    return n, quota


def cpu_baseline(workload):
    """The CPU oracle (numpy graph + C ops, oracle/) on the workload's clouds: single thread, and one single-threaded
    process per USABLE host core.  A reported baseline ("port"), never the thing measured as `value`."""
    from dh3d_amd import ConfigFactory
    from dh3d_amd.model import DH3D, tf_variable_name
    wl = WORKLOADS[workload]
    model = DH3D(ConfigFactory(wl["preset"]).getconfig()).init_synthetic(0)
    w = {tf_variable_name(k): v.detach().numpy() for k, v in model.state_dict().items()}
    tmp = tempfile.mkdtemp(prefix="dh3d_cpu_", dir="/tmp")
    path = os.path.join(tmp, "w.npz")
    np.savez(path, **w)
    glob = "1" if workload == "global" else "0"
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1", PYTHONPATH=ROOT)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)

    def worker(count, seed):
        return subprocess.Popen([sys.executable, "-m", "oracle.cpu_worker", path, str(wl["N"]), str(count), str(seed), glob],
                                cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    n1 = 4  # bounded sample: ~10-20 s of single-core work
    p = worker(n1, wl["seed"])
    t1 = float(p.communicate()[0].strip() or "nan")
    usable, quota = usable_cpus()
    workers = min(usable, 64)  # bounded: 64 single-threaded workers, two clouds each
    per_worker = 2
    t0 = time.perf_counter()
    procs = [worker(per_worker, wl["seed"] + 1 + i) for i in range(workers)]
    inner = [float(q.communicate()[0].strip() or "nan") for q in procs]
    wall = time.perf_counter() - t0
    # BASELINE config 1 on this host: knn_bruteforce + one flex_conv 32->32, N=1024, K=8, one thread (C oracle ops)
    q1 = subprocess.run([sys.executable, "-m", "oracle.cpu_worker", "cfg1", "5"], cwd=ROOT, env=env, capture_output=True,
                        text=True)
    try:
        cfg1_ms = float(q1.stdout.strip())
    except ValueError:
        cfg1_ms = None
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    single = n1 / t1
    allc = workers * per_worker / wall
    return {"value": single, "unit": "point-clouds/sec", "cores": 1, "kind": "port",
            "sample": "%d clouds of N=%d through oracle/model_np.forward (C oracle ops + numpy dense, one thread), %.1f s"
                      % (n1, wl["N"], t1),
            "cfg1_ms": cfg1_ms,
            "cfg1_sample": "BASELINE config 1: knn_bruteforce + one flex_conv 32->32 on one cloud of N=1024, K=8, C oracle "
                           "ops, one thread, best of 5",
            "all_cores": {"value": allc, "unit": "point-clouds/sec", "cores": workers,
                          "workers": workers, "clouds_per_worker": per_worker,
                          "median_worker_s": float(np.nanmedian(inner)), "slowest_worker_s": float(np.nanmax(inner)),
                          "scaling_vs_one_thread": allc / single if single > 0 else None,
                          "sample": "one single-threaded oracle process per usable core (capped at 64), %d clouds each, "
                                    "wall %.1f s (process start included)" % (per_worker, wall)},
            "cpu_model": cpu_model_string(), "host_cpus": os.cpu_count(), "usable_cpus": usable,
            "cgroup_cpu_quota": quota}


# --------------------------------------------------------------------------------------------- the stdout line
LINE_LIMIT = 8192   # bytes: the driver could not parse round 4's 32.8 kB line; everything but the contract goes to a side file

_ROOF_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes", "launch_ms")
_CPU_KEYS = ("value", "unit", "cores", "kind", "sample", "cpu_model", "cfg1_ms", "usable_cpus")


def _r(x, nd=8):
    """Floats to `nd` significant digits (the line is for reading and diffing, the side file keeps full precision)."""
    if isinstance(x, float):
        return float("%.*g" % (nd, x))
    if isinstance(x, dict):
        return {k: _r(v, nd) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, nd) for v in x]
    return x


def compact_line(full, extras_path=None, limit=LINE_LIMIT):
    """The ONE stdout line, from the full record: the contract's keys, `one_step_at_a_time`, `roofline`, `cpu_baseline`,
    the multi-GPU `global_scaling` block and the path of the side file -- never more than `limit` bytes, always
    json.loads-able (asserted here, at run time, and on a canned record by tests/test_bench_line.py)."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "ranks_seen")
    out = {k: full[k] for k in keep if k in full}
    cfg = dict(full.get("config", {}))
    for k, v in list(cfg.items()):
        if isinstance(v, str) and len(v) > 160:
            cfg[k] = v[:157] + "..."
    out["config"] = cfg
    if "in_flight_error" in full:
        out["in_flight_error"] = str(full["in_flight_error"])[:200]
    if "one_step_at_a_time" in full:
        s = full["one_step_at_a_time"]
        out["one_step_at_a_time"] = {"value": s.get("value"), "ms_per_step": s.get("ms_per_step")}
    for key in ("at_20_steps", "at_100_steps"):   # the same pipeline over the other K (20: rounds 1-5 / the driver; 100: the default)
        if isinstance(full.get(key), dict):
            out[key] = {k: full[key].get(k) for k in ("value", "ms_per_step")}
    for k in ("phases_ms", "step_graphed", "collectives_per_step"):   # the training workloads' few scalars
        if k in full and len(json.dumps(full[k])) < 600:
            out[k] = full[k]
    if "roofline" in full:
        r = full["roofline"]
        c = {k: r.get(k) for k in _ROOF_KEYS if k in r}
        for k in ("gather_effective", "f32_equivalent_flops"):
            if isinstance(r.get(k), dict):
                c[k] = {"frac": r[k].get("frac")}
        if isinstance(r.get("in_step"), dict):
            c["in_step"] = {"launch_ms": r["in_step"].get("launch_ms"), "frac": r["in_step"].get("frac")}
        c["traffic_source"] = str(r.get("traffic_source", ""))[:200]
        c["binding_roof"] = str(r.get("binding_roof", ""))[:120]
        out["roofline"] = c
    if "cpu_baseline" in full:
        b = full["cpu_baseline"]
        c = {k: b.get(k) for k in _CPU_KEYS if k in b}
        if isinstance(c.get("sample"), str):
            c["sample"] = c["sample"][:200]
        if isinstance(b.get("all_cores"), dict):
            c["all_cores"] = {"value": b["all_cores"].get("value"), "cores": b["all_cores"].get("cores")}
        out["cpu_baseline"] = c
    if "global_scaling" in full:
        out["global_scaling"] = full["global_scaling"]
    if isinstance(full.get("value_streaming"), dict):   # the serving loop: host batch in (+ descriptors out) every step
        v = full["value_streaming"]
        c = {k: v.get(k) for k in ("value", "ms_per_step", "steps_in_flight") if k in v}
        if "what" in v:
            c["what"] = str(v["what"])[:160]
        if isinstance(v.get("d2h_inclusive"), dict):
            c["d2h_inclusive"] = {k: v["d2h_inclusive"].get(k) for k in ("value", "GBps_d2h")}
        if isinstance(v.get("steady_state"), dict):
            c["steady_state"] = {k: v["steady_state"].get(k) for k in ("steps", "value", "ratio_to_resident")}
        if "error" in v:
            c["error"] = str(v["error"])[:160]
        out["value_streaming"] = c
    if isinstance(full.get("global"), dict):   # the other half of BASELINE.json's metric (fresh process, N = 1)
        g = full["global"]
        c = {k: g.get(k) for k in ("workload", "value", "ms_per_step", "steps_in_flight", "clouds_per_gpu", "points") if k in g}
        if isinstance(g.get("value_streaming"), dict):
            c["value_streaming"] = {k: g["value_streaming"].get(k) for k in ("value", "ms_per_step")}
            if isinstance(g["value_streaming"].get("steady_state"), dict):
                c["value_streaming"]["steady_state_ratio"] = g["value_streaming"]["steady_state"].get("ratio_to_resident")
        if isinstance(g.get("one_step_at_a_time"), dict):
            c["one_step_at_a_time"] = {"value": g["one_step_at_a_time"].get("value"),
                                       "ms_per_step": g["one_step_at_a_time"].get("ms_per_step")}
        for key in ("at_20_steps", "at_100_steps"):
            if isinstance(g.get(key), dict):
                c[key] = {k: g[key].get(k) for k in ("value", "ms_per_step")}
        if "error" in g:
            c["error"] = str(g["error"])[:200]
        out["global"] = c
    out["extras"] = extras_path
    out = _r(out)
    text = json.dumps(out, separators=(", ", ": "))
    # never over the limit: shed the optional blocks, least important first (none of these fire on today's record)
    for k in ("global_scaling.note", "roofline.traffic_source", "cpu_baseline.sample", "config.execution",
              "config.weights", "config.parallelism", "phases_ms", "global_scaling", "cpu_baseline.all_cores",
              "value_streaming.what", "value_streaming.d2h_inclusive", "value_streaming.steady_state", "value_streaming",
              "global.workload", "global"):
        if len(text.encode()) <= limit:
            break
        head, _, leaf = k.partition(".")
        if leaf and isinstance(out.get(head), dict):
            out[head].pop(leaf, None)
        else:
            out.pop(head, None)
        text = json.dumps(out, separators=(", ", ": "))
    assert len(text.encode()) <= limit, "bench line is %d bytes (> %d)" % (len(text.encode()), limit)
    assert "\n" not in text
    back = json.loads(text)
    assert back["metric"] == full["metric"] and back["value"] == _r(full["value"])
    return text


def write_side_file(full, path=None, workload="local"):
    """The full record (everything measured, full precision) as indented JSON; returns the path relative to the repo
    root, or None when nothing could be written (the stdout line never depends on it)."""
    if path is None:
        name = "bench_extras.json" if workload == "local" else "bench_extras_%s.json" % workload
        path = os.path.join(ROOT, "gpurun_out", name)
    try:
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, "w") as f:
            json.dump(full, f, indent=1)
            f.write("\n")
        return os.path.relpath(path, ROOT)
    except OSError:
        return None


# --------------------------------------------------------------------------------------------- main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # K = 100 by default since round 6: with steps in flight the K timed steps include the fill and the drain of the pipeline
    # (four slots for the local forward), which weighed ~5 % at the K = 20 of rounds 1-5 (and of the driver's command line,
    # `--steps 20 --warmup 5`); the line carries the other K's figure as `at_20_steps` / `at_100_steps`
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="local")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--batch", type=int, default=0, help="override the workload's batch (clouds)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the roofline of the named kernel too (headline only)")
    ap.add_argument("--extras", action="store_true",
                    help="also measure the evidence tables (step-level roofline, other workloads, sweeps in fresh processes, "
                         "data sensitivity): minutes of wall time; they go to the side file named by the line's `extras` key, "
                         "never onto the stdout line")
    ap.add_argument("--extras-file", default=None,
                    help="where the full record goes (default gpurun_out/bench_extras[_<workload>].json under the repo)")
    ap.add_argument("--no-global-scaling", action="store_true",
                    help="with --gpus N > 1 and the default workload: skip the global path's weak / strong lines")
    ap.add_argument("--no-pmc", action="store_true", help="do not collect live PMC traffic for the roofline kernel")
    ap.add_argument("--no-streaming", action="store_true",
                    help="skip `value_streaming` (the same pipeline fed a different pinned-host batch every step)")
    ap.add_argument("--no-global-line", action="store_true",
                    help="default workload at one GPU: do not measure the global-descriptor forward (the other half of "
                         "BASELINE.json's metric) in a fresh process for the line's `global` block")
    ap.add_argument("--repeats", type=int, default=7,
                    help="further blocks of K timed steps after the contract block (median / min / max as extra keys)")
    ap.add_argument("--inflight", type=int, default=0,
                    help="independent steps in flight (graph instances on separate streams, each with its own batch "
                         "buffers, every step one full pass over one batch).  Default 0 = the workload's own (local 4, "
                         "global 3, cfg5 4): a forward of this path is a latency chain on a few CUs (FPS: one CU per "
                         "cloud) followed by chip-wide kernels, so the engine overlaps consecutive batches; 1 = one step "
                         "at a time (also measured and reported as `one_step_at_a_time` in every line).  The training "
                         "step always runs one at a time")
    args = ap.parse_args()
    explicit_inflight = args.inflight > 0
    if not explicit_inflight:
        args.inflight = WORKLOADS[args.workload]["inflight"]

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        relaunch_under_torchrun(args.gpus)  # does not return

    # The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a version / host block on stdout
    # when a communicator is created): everything but the final line is routed to stderr at the file-descriptor level.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    from dh3d_amd import dist as D
    rank, world = D.init_from_env()
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with --nproc-per-node %d, or drop the torchrun "
                         "environment and let bench.py spawn the ranks itself)" % (args.gpus, world, args.gpus))
    if os.environ.get("DH3D_BENCH_SHARE_GPU") == "1":
        # dev: every rank on GPU 0 (with DH3D_DIST_BACKEND=gloo) -- exercises the N > 1 code path of this file on a
        # one-GPU box; the numbers of such a run mean nothing
        local_rank = 0
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit("bench.py: rank %d has no GPU (device_count=%d)" % (local_rank, torch.cuda.device_count()))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    ranks_seen = 1
    if world == 1 and os.environ.get("DH3D_BENCH_RCCL_AT_ONE_RANK") == "1":
        # dev: the process state of a multi-GPU rank on one GPU -- a 1-rank RCCL communicator (its streams and kernels)
        # exists and the barriers of the timed region go through it, before anything is measured
        torch.distributed.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1)
        one = torch.ones(1, device=dev)
        torch.distributed.all_reduce(one)
    if world > 1:
        one = torch.ones(1, device=dev)
        torch.distributed.all_reduce(one)  # over RCCL: every rank is alive on its own GPU
        ranks_seen = int(one.item())
    strong = args.scaling == "strong" or args.workload == "train"

    def per_rank_batch(B, strong_=None):
        if not (strong if strong_ is None else strong_):
            return B, B * world
        per = (B + world - 1) // world
        return per, B

    def resident_step(tr):
        """step(p) with the batch resident in the replayed step's own input buffer once that exists (the loader's
        zero-copy hand-over, QuadrupletTrainer.input_buffer) -- as the inference workloads' graphs read theirs."""
        state = {"buf": None}

        def step(p):
            if state["buf"] is None:
                b = tr.input_buffer(p.shape)
                if b is not None:
                    b.copy_(p)
                    state["buf"] = b
            return tr.step(state["buf"] if state["buf"] is not None else p, sync=False)
        return step

    def measure_train():
        from dh3d_amd import ConfigFactory
        from dh3d_amd.model import DH3D
        from dh3d_amd.training import QuadrupletTrainer
        wl = WORKLOADS["train"]
        cfg = ConfigFactory("global_config").getconfig()
        cfg.batch_size, cfg.num_pos, cfg.num_neg, cfg.num_points = 1, 2, 18, wl["N"]
        model = DH3D(cfg).init_synthetic(0).to(dev).eval().prepare()
        trainer = QuadrupletTrainer(model)
        pts = synthetic_clouds(wl["B"], wl["N"], wl["seed"], dev, 0)  # same role-ordered batch on every rank
        # (the loss stays on the device: nothing in the timed loop waits for the GPU but the closing synchronize)
        dt = time_steps(resident_step(trainer), pts, args.steps, args.warmup, dev)
        trainer.time_phases(True)   # five more (untimed) steps with events around the phases
        D.COLLECTIVE_CALLS[0] = 0
        for _ in range(5):
            trainer.step(pts)
        extra = {"phases_ms": trainer.phase_times_ms(), "head_implementation": trainer.impl,
                 "step_graphed": bool(trainer._step_graphs), "collectives_per_step": D.COLLECTIVE_CALLS[0] / 5.0,
                 "frozen_backbone_batchnorm": "moving averages (fused inference path); the reference graph uses batch "
                                              "statistics there -- QuadrupletTrainer(backbone_bn='batch'), DESIGN.md 6"}
        trainer.time_phases(False)
        if world == 1:
            extra["sharded_path"] = sharded_path_at_one_rank(model, pts, dt / args.steps * 1e3)
            # the reference's own graph: the frozen backbone's BatchNorms on BATCH statistics, moving averages updated
            # (core/tf_utils.py:145-153), on HIP kernels inside the same whole-step hipGraph
            try:
                model2 = DH3D(cfg).init_synthetic(0).to(dev).eval().prepare()
                tr2 = QuadrupletTrainer(model2, backbone_bn="batch")
                dt2 = time_steps(resident_step(tr2), pts, args.steps, max(args.warmup, 5), dev)
                extra["reference_semantics"] = {
                    "ms_per_step": dt2 / args.steps * 1e3, "value": wl["B"] * args.steps / dt2,
                    "step_graphed": bool(tr2._step_graphs), "ratio_to_moving_average_step": dt2 / dt,
                    "note": "QuadrupletTrainer(backbone_bn='batch'): batch-statistics BatchNorm + EMA updates in the frozen "
                            "backbone (8 sites: colstats / finalize / apply on HIP kernels), what upstream trains with"}
            except Exception as e:  # noqa: BLE001 -- informational key
                extra["reference_semantics"] = {"error": repr(e)[:200]}
        return wl["B"] * args.steps / dt, dt / args.steps * 1e3, extra

    def sharded_path_at_one_rank(model, pts, plain_ms):
        """The step as the sharded run issues it -- sync-BN statistics all-reduced, descriptors all-gathered, the flat
        gradient arena all-reduced, all inside the replayed hipGraph -- on a 1-rank RCCL group, next to the plain
        single-process step: what the collectives' code path costs before any second GPU is involved."""
        import socket
        import torch.distributed as tdist
        from dh3d_amd.training import QuadrupletTrainer
        out = {"plain_ms": plain_ms}
        own_group = not tdist.is_initialized()
        try:
            if own_group:
                with socket.socket() as sk:
                    sk.bind(("127.0.0.1", 0))
                    port = sk.getsockname()[1]
                tdist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
            D.FORCE_COLLECTIVES = True
            tr = QuadrupletTrainer(model, sync_bn=True)
            dt = time_steps(resident_step(tr), pts, args.steps, max(args.warmup, 5), dev)
            out["sharded_path_ms"] = dt / args.steps * 1e3
            out["ratio"] = out["sharded_path_ms"] / plain_ms
            out["step_graphed"] = bool(tr._step_graphs)
            tr.graph_step = False  # one eager step: count what a step issues
            D.COLLECTIVE_CALLS[0] = 0
            launches = None
            try:
                from torch.profiler import profile, ProfilerActivity
                with profile(activities=[ProfilerActivity.CUDA]) as prof:
                    tr.step(pts)
                    torch.cuda.synchronize()
                launches = sum(1 for e in prof.events() if str(getattr(e, "device_type", "")).endswith("CUDA"))
            except Exception:  # no profiler on this box: the collective count alone
                tr.step(pts)
            out["collectives_per_step"] = D.COLLECTIVE_CALLS[0]
            out["device_launches_per_step"] = launches
        except Exception as e:  # noqa: BLE001 -- informational key: the bench line must not die with it
            out["error"] = repr(e)[:200]
        finally:
            D.FORCE_COLLECTIVES = False
            if own_group and tdist.is_initialized():
                tdist.destroy_process_group()
        return out

    def measure_train_local():
        """Stage-1 training (core/configs.py:35-82 basic_config: batch_size 10, num_pos 1, 8192 points, 512 keypoints):
        the whole local backbone in training mode, forward + backward + Adam, one hipGraph per step."""
        from dh3d_amd import ConfigFactory
        from dh3d_amd.model import DH3D
        from dh3d_amd.training import LocalTrainer
        wl = WORKLOADS["train_local"]
        cfg = ConfigFactory(wl["preset"]).getconfig()
        model = DH3D(cfg).init_synthetic(0).to(dev).eval().prepare()
        tr = LocalTrainer(model)
        pairs, M = wl["B"] // 2, cfg.sampled_kpnum
        rng = np.random.default_rng(wl["seed"])
        anc = (rng.random((pairs, wl["N"], 3), dtype=np.float32) * 30.0).astype(np.float32)
        Rm = np.tile(np.eye(3, dtype=np.float32), (pairs, 1, 1))
        pos = (anc + rng.normal(0, 0.02, anc.shape)).astype(np.float32)
        ia = np.stack([rng.permutation(wl["N"])[:M] for _ in range(pairs)]).astype(np.int32)
        pts = torch.from_numpy(np.concatenate([anc, pos])).to(dev)
        Rt, idx = torch.from_numpy(Rm).to(dev), torch.from_numpy(np.concatenate([ia, ia])).to(dev)
        for _ in range(4):  # three eager steps, then the capture
            tr.step(pts, Rt, idx)
        key = next(iter(tr._graphs)) if tr._graphs else None
        if key is not None:  # the batch resident in the graph's own input buffers
            pts, Rt, idx = tr._graphs[key][1]
        dt = time_steps(lambda p: tr.step(pts, Rt, idx, sync=False), pts, args.steps, args.warmup, dev)
        extra = {"step_graphed": bool(tr._graphs), "trainable_tensors": len(tr.params),
                 "losses": "desc_local_loss x local_loss_weight (core/losses.py:29-63) via losses.compute_loss"}
        return wl["B"] * args.steps / dt, dt / args.steps * 1e3, extra

    def measure(workload, batch=None, strong_=None):
        if workload == "train":
            return measure_train()
        if workload == "train_local":
            return measure_train_local()
        wl = WORKLOADS[workload]
        per, total = per_rank_batch(batch or (args.batch if workload == args.workload and args.batch else wl["B"]), strong_)
        model = build_model(wl["preset"], dev, seed=0, num_points=wl["N"])
        pts = synthetic_clouds(per, wl["N"], wl["seed"], dev, rank)
        with torch.no_grad():
            run = model.graphed(pts, outputs=(wl["out"],))
            # the batch is resident in the graph's input buffer (where a loader's H2D copy would put it)
            run.static_input.copy_(pts)
            dt = time_steps(run, run.static_input, args.steps, args.warmup, dev)
            info = {"clouds_per_gpu": per}
            if workload == args.workload and args.repeats > 0:
                blocks = repeat_blocks(run, run.static_input, args.steps, dev, args.repeats)
                info["repeats"] = {"blocks": args.repeats, "steps_per_block": args.steps,
                                   "ms_per_step": [round(b, 5) for b in blocks],
                                   "median_ms": float(np.median(blocks)), "min_ms": float(np.min(blocks)),
                                   "max_ms": float(np.max(blocks)),
                                   "median_value": total / (float(np.median(blocks)) * 1e-3),
                                   "note": "further timed blocks after the contract one; `value` is the first block"}
        return total * args.steps / dt, dt / args.steps * 1e3, info

    _STREAM_POOL = []

    def measure_in_flight(workload, depth=2, repeats=0, steps=None, strong_=None, streaming=False):
        """Throughput with `depth` independent steps in flight: `depth` graph instances on `depth` streams, each with its
        own batch buffers; step i is one full pass over one batch on stream i % depth.  A single forward of this path
        leaves most of the GPU idle (FPS: one CU per cloud for two thirds of the step), so consecutive batches overlap;
        the K timed steps include the pipeline's fill and drain (barrier + synchronize on both sides as always)."""
        wl = WORKLOADS[workload]
        per, total = per_rank_batch(args.batch if workload == args.workload and args.batch else wl["B"], strong_)
        model = build_model(wl["preset"], dev, seed=0, num_points=wl["N"])
        pts = synthetic_clouds(per, wl["N"], wl["seed"], dev, rank)
        # (the same streams for every measurement of the process)
        while len(_STREAM_POOL) < depth:
            _STREAM_POOL.append(torch.cuda.Stream(device=dev))
        with torch.no_grad():
            # the package's engine (dh3d_amd/engine.py: model.pipeline): one hipGraph instance + batch buffers + stream
            # per slot, the persistent kernels' placement hint set for `depth` steps in flight
            pipe = model.pipeline(pts, depth=depth, outputs=(wl["out"],), streams=_STREAM_POOL[:depth])
            for k in range(depth):  # a DIFFERENT batch resident in every slot (where a loader's H2D copy would put it)
                pipe.input_buffer(k).copy_(synthetic_clouds(per, wl["N"], wl["seed"] + 7919 * k, dev, rank))
            torch.cuda.synchronize()

            def step(p):
                pipe.submit()  # zero-copy: the slot's batch is already in its input buffer

            nsteps = steps or args.steps
            dt = time_steps(step, pts, nsteps, args.warmup, dev)
            if workload == args.workload and not steps:
                # the same pipeline over the OTHER K: 20 = the K of rounds 1-5 and of the driver's command line, 100 = this
                # file's default (the timed region includes the fill and the drain of the slots: ~5 % at K = 20)
                ko = 100 if nsteps == 20 else 20
                dk = time_steps(step, pts, ko, args.warmup, dev)
                _AT20[workload] = {"steps": ko, "value": total * ko / dk, "ms_per_step": dk / ko * 1e3}
            rep = None
            if repeats > 0:
                blocks = repeat_blocks(step, pts, args.steps, dev, repeats)
                rep = {"blocks": repeats, "steps_per_block": args.steps, "ms_per_step": [round(b, 5) for b in blocks],
                       "median_ms": float(np.median(blocks)), "min_ms": float(np.min(blocks)),
                       "max_ms": float(np.max(blocks)), "median_value": total / (float(np.median(blocks)) * 1e-3),
                       "note": "further timed blocks after the contract one; `value` is the first block"}
            if streaming and world == 1:
                try:
                    _STREAMING[workload] = measure_streaming(pipe, wl, per, total, nsteps)
                except Exception as e:  # noqa: BLE001 -- informational key
                    _STREAMING[workload] = {"error": repr(e)[:200]}
        return total * nsteps / dt, dt / nsteps * 1e3, rep

    _STREAMING = {}
    _AT20 = {}

    def measure_streaming(pipe, wl, per, total, nsteps):
        """The same pipeline as a SERVING loop (what engine.Pipeline.map runs; localdesc_extract.py:106-138,
        globaldesc_extract.py:84-100): every step's batch is a DIFFERENT pinned-host batch, copied H2D on the slot's own
        stream in front of the replay (Pipeline.submit(host_batch)); every step's output is consumed before its slot is
        reused -- small outputs (the [B, 256] global descriptors) are copied D2H into pinned memory on the slot's stream,
        the dense local map [B, N, 131] (34 MB per step: 140 GB/s at this step rate, more than a PCIe 5 x16 link carries)
        is copied out of the slot's buffers into a device-side consumer's buffer (what Pipeline.map's clone does); the
        D2H-inclusive rate of the dense map is reported beside it.  Timed like `value`: K steps, fill and drain included."""
        depth = pipe.depth
        nb = 2 * depth + 1
        host = [torch.from_numpy(np.random.default_rng(wl["seed"] + 104729 * (i + 1)).random((per, wl["N"], 3), dtype=np.float32)).pin_memory()
                for i in range(nb)]
        name = wl["out"]
        shape = tuple(pipe._runs[0].outputs[name].shape)
        out_bytes = int(np.prod(shape)) * 4
        small = out_bytes <= (1 << 20)
        # two generations of host buffers per slot: the host-side consumer may lag a whole round behind the submitting loop
        # (Pipeline keeps 2 * depth ticket events), so the loop blocks only on a step submitted 2 * depth steps ago
        host_out = [torch.empty(shape, dtype=torch.float32).pin_memory() for _ in range(2 * depth)]
        sink = [torch.empty(shape, dtype=torch.float32, device=dev) for _ in range(depth)]

        # Both consumers live on the SLOT's stream (Pipeline.submit(fetch_to=...)): a separate consumer stream costs a
        # cross-stream event pair per step and shares one of the process's four hardware queues with a slot -- its waits
        # stall that slot (first version of this block: 6.7 k clouds/s against 33 k).
        dbg = os.environ.get("DH3D_STREAM_DEBUG")

        def loop(n, d2h):
            tickets = []
            tl = time.perf_counter()
            for i in range(n):
                if d2h and len(tickets) == 2 * depth:
                    tickets.pop(0).event.synchronize()   # that step's descriptors are in host memory; its buffer is free again
                k = pipe.next_slot
                tickets.append(pipe.submit(host[i % nb], fetch_to={name: host_out[i % (2 * depth)] if d2h else sink[k]}))
                if dbg:
                    tn = time.perf_counter()
                    if tn - tl > 1e-3:
                        print("[stream dbg] n=%d d2h=%s submit %d took %.2f ms" % (n, d2h, i, (tn - tl) * 1e3), file=sys.stderr)
                    tl = tn
            if d2h:
                for tk in tickets:
                    tk.event.synchronize()

        def timed(d2h, n=None):
            n = n or nsteps
            loop(max(args.warmup, depth), d2h)
            torch.cuda.synchronize()
            D.barrier()
            t0 = time.perf_counter()
            loop(n, d2h)
            torch.cuda.synchronize()
            D.barrier()
            return time.perf_counter() - t0

        def zero_copy_long(n):   # the resident-batch loop of `value` over the same long run, for the steady-state ratio
            for _ in range(depth):
                pipe.submit()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                pipe.submit()
            torch.cuda.synchronize()
            return time.perf_counter() - t0
        rec = {"steps_in_flight": depth, "distinct_host_batches": nb, "h2d_bytes_per_step": per * wl["N"] * 12,
               "output_bytes_per_step": out_bytes}
        long_n = 10 * nsteps
        # Three blocks, the median reported (all listed): a serving loop's blocks see a sporadic one-off host / runtime stall
        # of ~85 ms once in a few runs (r06d: in the first block, 1.7 k clouds/s; r06f: in the long run, ratio 0.34; not
        # reproducible in tools/hiccup_probe*.py, tools/streaming_steady.py: 0.96-0.97 of the resident loop over 1000 steps)
        def blocks3(d2h):
            ts = sorted(timed(d2h) for _ in range(3))
            rec.setdefault("blocks_ms_per_step", {})["d2h" if d2h else "device_sink"] = [round(t / nsteps * 1e3, 5) for t in ts]
            return ts[1]

        def long2(d2h):
            ts = [timed(d2h, long_n) for _ in range(2)]
            rec["steady_state_runs_ms_per_step"] = [round(t / long_n * 1e3, 5) for t in ts]
            return min(ts)
        if small:
            dt = blocks3(True)
            rec.update({"value": total * nsteps / dt, "ms_per_step": dt / nsteps * 1e3,
                        "what": "pinned host batch -> H2D on the slot's stream -> step -> D2H of `%s` into pinned host memory "
                                "on the slot's stream; the host picks results up one round (2 x depth buffers) behind the submitting loop; median of three blocks" % name})
            dl, dz = long2(True), zero_copy_long(long_n)
        else:
            dt = blocks3(False)
            rec.update({"value": total * nsteps / dt, "ms_per_step": dt / nsteps * 1e3,
                        "what": "pinned host batch -> H2D on the slot's stream -> step -> `%s` copied out of the slot's buffers "
                                "into the consumer's device buffer on the slot's stream; median of three blocks" % name})
            dl, dz = long2(False), zero_copy_long(long_n)
            dt2 = blocks3(True)
            rec["d2h_inclusive"] = {"value": total * nsteps / dt2, "ms_per_step": dt2 / nsteps * 1e3,
                                    "GBps_d2h": out_bytes * nsteps / dt2 / 1e9,
                                    "note": "the dense map copied to pinned host memory every step: bound by the host link, "
                                            "not by the path (never `value`)"}
        # K = 20 steps through a pipeline `depth` deep weigh its fill and drain, where the staging copies' latency shows; over
        # 10 K steps: the same loop against the resident-batch loop of `value`
        rec["steady_state"] = {"steps": long_n, "value": total * long_n / dl, "ms_per_step": dl / long_n * 1e3,
                               "resident_batches_ms_per_step": dz / long_n * 1e3, "ratio_to_resident": dz / dl}
        return rec

    pipelined = args.inflight > 1 and args.workload not in ("train", "train_local")
    value, ms, info = measure(args.workload)  # one step at a time (the definition of rounds 1-2; `value` for train)
    serial = {"value": value, "unit": "point-clouds/sec", "ms_per_step": ms,
              "note": "each step finishes before the next one starts (rounds 1-2 reported this as `value`)"}
    if "repeats" in info:
        serial["repeats"] = info.pop("repeats")
    in_flight_error = None
    if pipelined:
        try:
            value, ms, rep = measure_in_flight(args.workload, args.inflight, args.repeats, streaming=not args.no_streaming)
            if rep:
                info["repeats"] = rep
        except Exception as e:  # noqa: BLE001 -- the line must not be lost: fall back to the one-at-a-time measurement
            if world > 1:  # (the ranks could not agree on the fallback without another collective)
                raise
            in_flight_error = repr(e)[:300]
            pipelined = False
    wl = WORKLOADS[args.workload]
    per, total = per_rank_batch(args.batch or wl["B"])
    line = {
        "metric": "point-clouds/sec", "value": value, "unit": "point-clouds/sec", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
        "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "ranks_seen": ranks_seen,
        "config": {"workload": wl["name"], "clouds_per_gpu": per, "clouds_total": total, "points": wl["N"], "knn": 8,
                   "parallelism": "clouds sharded over %d GPU(s), no data-path collective" % world,
                   "weights": "random-init (no checkpoint blobs exist upstream)",
                   "execution": ("dh3d_amd.engine.Pipeline (DH3D.pipeline): hipGraph replay, %d steps in flight -- one graph "
                                 "instance + its own batch + stream per slot, every step one full pass over one batch"
                                 % args.inflight) if pipelined else "hipGraph replay",
                   "steps_in_flight": args.inflight if pipelined else 1},
    }
    if in_flight_error:
        line["in_flight_error"] = in_flight_error
    if pipelined and args.workload in _AT20:
        line["at_%d_steps" % _AT20[args.workload]["steps"]] = {k: _AT20[args.workload][k] for k in ("value", "ms_per_step")}
    if pipelined and args.workload in _STREAMING:
        line["value_streaming"] = _STREAMING[args.workload]
    if args.workload not in ("train", "train_local"):
        line["one_step_at_a_time"] = serial
    if args.workload == "train_local":
        line["config"]["execution"] = ("whole step (local backbone fwd/bwd in training mode, loss, weight decay, fused Adam) "
                                       "replayed as one hipGraph" if info.get("step_graphed") else "eager")
        line["config"]["parallelism"] = "single GPU (LocalTrainer has no sharded path yet: DESIGN.md 6)"
        line["config"].pop("knn", None)
        line.update(info)
    if "repeats" in info:
        line["repeats"] = info["repeats"]
    if args.workload == "train":
        line["config"]["execution"] = (
            "whole step (backbone, head fwd/bwd, loss, fused Adam) replayed as one hipGraph from its own input buffer; phases_ms from eager steps"
            if info.get("step_graphed") else
            "eager: backbone hipGraph (no grad) + trainable global head fwd/bwd + fused Adam")
        line["config"]["parallelism"] = ("one role-ordered batch sharded over %d GPU(s); all-gather of [clouds,256] "
                                         "descriptors + SUM all-reduce of head gradients over RCCL" % world)
        line.update(info)
    def fresh_process_in_flight(workload, depth, steps, batch=None):
        """(value, ms_per_step) of `bench.py --workload w --inflight depth` in a process of its own.  How well steps in
        flight overlap depends on the live streams and graph instances of the process (they share four hardware
        queues): measured after other workloads in THIS process the same configuration gave 16.9 k where a fresh process
        gives 24 k.  The line's `value` is measured first, in the state a process that only runs this workload has; the
        informational numbers get that state too."""
        import subprocess
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--workload", workload, "--inflight", str(depth),
               "--steps", str(steps), "--warmup", str(args.warmup), "--no-extras", "--no-cpu-baseline", "--repeats", "0",
               "--no-streaming", "--extras-file", os.devnull]
        if batch:
            cmd += ["--batch", str(batch)]
        try:
            out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=300).stdout.decode()
            rec = json.loads([ln for ln in out.splitlines() if ln.startswith('{"metric"')][-1])
            return rec["value"], rec["ms_per_step"]
        except Exception:  # noqa: BLE001 -- informational numbers: fall back to this process
            v, ms, _ = measure_in_flight(workload, depth, steps=steps)
            return v, ms

    def fresh_process_line(workload):
        """The compact record of `bench.py --workload w` (its own steps in flight AND one step at a time) measured in a
        process of its own, for the default line's `global` block: BASELINE.json's metric names both forwards, the driver
        runs this file once without flags."""
        import subprocess
        wl_ = WORKLOADS[workload]
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--workload", workload, "--steps", str(args.steps),
               "--warmup", str(args.warmup), "--no-extras", "--no-cpu-baseline", "--repeats", "0", "--extras-file", os.devnull]
        try:
            t0 = time.time()
            out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=300).stdout.decode()
            rec = json.loads([ln for ln in out.splitlines() if ln.startswith('{"metric"')][-1])
            return {"workload": wl_["name"], "value": rec["value"], "unit": rec["unit"], "ms_per_step": rec["ms_per_step"],
                    "steps": rec["steps"], "warmup": rec["warmup"], "clouds_per_gpu": rec["config"]["clouds_per_gpu"],
                    "points": rec["config"]["points"], "steps_in_flight": rec["config"]["steps_in_flight"],
                    "one_step_at_a_time": rec.get("one_step_at_a_time"), "value_streaming": rec.get("value_streaming"),
                    "at_20_steps": rec.get("at_20_steps"), "at_100_steps": rec.get("at_100_steps"),
                    "measured_in": "a fresh process of this file "
                    "(--workload %s), same timed-region definition as `value`" % workload, "wall_s": time.time() - t0}
        except Exception as e:  # noqa: BLE001 -- the headline must not be lost with it
            return {"workload": wl_["name"], "error": repr(e)[:200]}

    if rank == 0 and world == 1 and args.workload == "local" and not args.no_extras and not args.no_global_line:
        line["global"] = fresh_process_line("global")

    # --- N > 1, default workload: the GLOBAL path's weak and strong figures ride on the same line (BASELINE's ">= 6.5x at 8
    # GPUs on the global-descriptor path" is read both ways; every rank takes part: the timed regions hold barriers)
    if world > 1 and args.workload == "local" and not args.no_global_scaling:
        gw = WORKLOADS["global"]
        gs = {"workload": gw["name"], "steps_in_flight": gw["inflight"]}
        for tag, st in (("weak", False), ("strong", True)):
            try:
                v1, m1, _ = measure("global", batch=gw["B"], strong_=st)
                rec = {"clouds_per_gpu": per_rank_batch(gw["B"], st)[0], "clouds_total": per_rank_batch(gw["B"], st)[1],
                       "one_step_at_a_time": {"value": v1, "ms_per_step": m1}}
                try:
                    v2, m2, _ = measure_in_flight("global", gw["inflight"], strong_=st)
                    rec["in_flight"] = {"value": v2, "ms_per_step": m2}
                except Exception as e:  # noqa: BLE001
                    if world > 1:
                        raise
                    rec["in_flight"] = {"error": repr(e)[:120]}
                gs[tag] = rec
            except Exception as e:  # noqa: BLE001 -- all ranks raise together or none (same code path, same shapes)
                if world > 1:
                    raise
                gs[tag] = {"error": repr(e)[:120]}
        gs["note"] = ("weak = 32 clouds PER GPU (no data-path collective); strong = ONE 32-cloud batch split over the GPUs "
                      "(ceil(32/N) each) -- bounded by the per-cloud latency chain, predicted 2.2-2.7x at 8 GPUs from the "
                      "1-GPU batch sweep (DESIGN.md 6)")
        line["global_scaling"] = gs

    roof = rank == 0 and not args.no_extras and args.workload in ("local", "global")
    extras = roof and args.extras
    if roof:
        with torch.no_grad():
            in_step = flex_in_step_ms(dev)
            line["roofline"] = flex_conv_roofline(dev, in_step_ms=in_step, pmc=(not args.no_pmc and world == 1))
    if extras:
        with torch.no_grad():
            if world == 1:
                try:
                    line["roofline_global"] = global_tail_roofline(dev)
                except Exception as e:  # noqa: BLE001 -- informational key
                    line["roofline_global"] = {"error": repr(e)[:200]}
            line["kernels_ms"] = kernel_breakdown(dev, wl["B"], wl["N"])
            if world == 1:
                line["dropin_ops_ms"] = dropin_ops_line(dev)
                line["roofline_step"] = step_roofline(args.workload, serial_ms=serial["ms_per_step"],
                                                      in_flight_ms=ms if pipelined else None,
                                                      depth=args.inflight if pipelined else None)
    if extras and world == 1:
        others = []
        for other in ("local", "global", "cfg5", "train"):
            if other == args.workload:
                continue
            ov, oms, oinfo = measure(other, batch=WORKLOADS[other]["B"])
            rec = {"workload": WORKLOADS[other]["name"], "key": other, "value": ov, "unit": "point-clouds/sec",
                   "ms_per_step": oms, "steps_in_flight": 1}
            rec.update(oinfo)
            if pipelined and other != "train":  # the same definition as the line's `value`
                rec["one_step_at_a_time"] = {"value": ov, "ms_per_step": oms}
                odepth = args.inflight if explicit_inflight else WORKLOADS[other]["inflight"]
                rec["value"], rec["ms_per_step"] = fresh_process_in_flight(other, odepth, args.steps)
                rec["steps_in_flight"] = odepth
            if other in ("local", "global"):
                rec["roofline_step"] = step_roofline(other, serial_ms=oms,
                                                     in_flight_ms=rec["ms_per_step"] if rec["steps_in_flight"] > 1 else None,
                                                     depth=rec["steps_in_flight"] if rec["steps_in_flight"] > 1 else None)
            if other == "cfg5":
                with torch.no_grad():
                    rec["kernel_roofline"] = cfg5_kernel_line(dev)
                    rec["kernels_ms"] = kernel_breakdown(dev, WORKLOADS[other]["B"], WORKLOADS[other]["N"])
            others.append(rec)
        line["other_workloads"] = others
        line["other_workload"] = others[0]  # round-1 key, kept for the driver's diff
        if args.workload != "train":
            depths = {"1": serial["value"]}
            for dpt in (2, 4):
                # (the sweep times 25 steps per step in flight: a deep pipeline's fill and drain weigh on K = 20)
                depths[str(dpt)] = value if (pipelined and dpt == args.inflight) else fresh_process_in_flight(args.workload, dpt, max(args.steps, 25 * dpt))[0]
            line["throughput_by_steps_in_flight"] = depths
        # per-GPU throughput against the local batch: what `--scaling strong` gives each GPU at 8 / 4 / 2 GPUs
        # (SURVEY 8e "Expected scaling": the FPS / kNN latency chain does not shrink with the batch)
        sweep = []
        for b in (wl["B"], wl["B"] // 2, wl["B"] // 4, wl["B"] // 8):
            if b >= 1:
                v, m_, _ = measure(args.workload, batch=b)
                sweep.append({"clouds_per_gpu": b, "value": v, "ms_per_step": m_})
        if args.workload in ("local", "global", "cfg5"):
            try:  # the same graphed step on scene-like clouds (informational: every other number is on the uniform cube)
                from dh3d_amd import pm
                model = build_model(wl["preset"], dev, seed=0, num_points=wl["N"])
                sp = scene_like_clouds(wl["B"], wl["N"], wl["seed"], dev)
                up = synthetic_clouds(wl["B"], wl["N"], wl["seed"], dev, 0)
                with torch.no_grad():
                    run = model.graphed(sp, outputs=(wl["out"],))
                    run.static_input.copy_(sp)
                    dts = time_steps(run, run.static_input, args.steps, args.warmup, dev)
                    knn = {}
                    for nm, p_ in (("uniform_cube", up), ("scene_like", sp)):
                        srt, gbox, cells = pm.spatial_sort_cells(p_)
                        knn[nm] = {"dh3d_knn_grid_ms": event_time_ms(lambda: pm.knn_grid(srt, gbox, cells, 8), iters=20, warm=3),
                                   "pruned_scan_ms": event_time_ms(lambda: pm.knn_sorted(srt, gbox, 8), iters=20, warm=3),
                                   "clouds_sent_to_the_scan": int((cells[:, 4106] != 0).sum())}
                real = {}
                try:  # the reference's own demo clouds (fixture): the same step and the same kNN launches
                    rp = real_oxford_clouds(wl["B"], wl["N"], dev)
                    if rp is not None:
                        with torch.no_grad():
                            rrun = model.graphed(rp, outputs=(wl["out"],))
                            rrun.static_input.copy_(rp)
                            dtr = time_steps(rrun, rrun.static_input, args.steps, args.warmup, dev)
                            srt, gbox, cells = pm.spatial_sort_cells(rp)
                            real = {"input": "evaluate/local_eval/demo_data/{268,642}.bin (Oxford LiDAR, metres), the %d points "
                                             "nearest to the centroid, %d clouds" % (wl["N"], wl["B"]),
                                    "one_step_at_a_time": {"ms_per_step": dtr / args.steps * 1e3, "value": wl["B"] * args.steps / dtr,
                                                           "ratio_to_uniform_cube": (dtr / args.steps * 1e3) / serial["ms_per_step"]},
                                    "knn_K8": {"dh3d_knn_grid_ms": event_time_ms(lambda: pm.knn_grid(srt, gbox, cells, 8), iters=20, warm=3),
                                               "pruned_scan_ms": event_time_ms(lambda: pm.knn_sorted(srt, gbox, 8), iters=20, warm=3),
                                               "clouds_sent_to_the_scan": int((cells[:, 4106] != 0).sum())}}
                except Exception as e:  # noqa: BLE001
                    real = {"error": repr(e)[:200]}
                line["data_sensitivity_real_oxford"] = real
                line["data_sensitivity"] = {
                    "input": "scene-like clouds (bench.scene_like_clouds: ground plane + walls + clutter in [-1, 1], z extent a tenth of x / y)",
                    "one_step_at_a_time": {"ms_per_step": dts / args.steps * 1e3, "value": wl["B"] * args.steps / dts,
                                           "ratio_to_uniform_cube": (dts / args.steps * 1e3) / serial["ms_per_step"]},
                    "knn_K8": knn,
                    "note": "FPS box pruning, kNN and three_nn are data dependent; results are bit-equal on either input "
                            "(tests/test_ops_gpu.py), DESIGN.md 3.0 'Data dependence'"}
            except Exception as e:  # noqa: BLE001 -- informational key
                line["data_sensitivity"] = {"error": repr(e)[:200]}
        line["batch_sweep_1gpu"] = {"workload": args.workload, "points": sweep,
                                    "predicted_strong_scaling_8gpu": (8 * sweep[-1]["value"] / sweep[0]["value"])
                                    if len(sweep) == 4 else None}
        if args.workload != "global":
            # BASELINE's scaling target (>= 6.5x at 8 GPUs) is quoted on the GLOBAL path: the same sweep for it, one step at
            # a time and with its steps in flight (fresh processes), and what it predicts.  No 8-GPU node has been
            # available in any round: these are 1-GPU measurements + arithmetic, not a scaling curve.
            gw = WORKLOADS["global"]
            gsweep = []
            for b in (gw["B"], gw["B"] // 2, gw["B"] // 4, gw["B"] // 8):
                v, m_, _ = measure("global", batch=b)
                rec = {"clouds_per_gpu": b, "one_step_at_a_time": {"value": v, "ms_per_step": m_}}
                if pipelined:
                    fv, fm = fresh_process_in_flight("global", gw["inflight"], args.steps, batch=b)
                    rec["in_flight"] = {"value": fv, "ms_per_step": fm, "steps_in_flight": gw["inflight"]}
                gsweep.append(rec)
            pred = {"weak_8gpu": 8.0,
                    "strong_8gpu_one_step_at_a_time": 8 * gsweep[-1]["one_step_at_a_time"]["value"] / gsweep[0]["one_step_at_a_time"]["value"]}
            if pipelined:
                pred["strong_8gpu_in_flight"] = 8 * gsweep[-1]["in_flight"]["value"] / gsweep[0]["in_flight"]["value"]
            line["batch_sweep_global_1gpu"] = {
                "workload": "global", "points": gsweep, "predicted_scaling": pred,
                "note": "weak scaling (32 clouds per GPU, the bench's --scaling weak) has no data-path collective: 8x by "
                        "construction, minus whatever eight processes on one host cost; strong scaling of ONE 32-cloud batch "
                        "(4 clouds per GPU) is bounded by the per-cloud latency chain, not by communication.  UNMEASURED on "
                        "more than one GPU in rounds 1-5 (no 8-GPU node was available to the driver)"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload in ("local", "global"):
        line["cpu_baseline"] = cpu_baseline(args.workload)
    D.barrier()
    if rank == 0:
        path = write_side_file(line, args.extras_file, args.workload)
        out = compact_line(line, extras_path=path)
        sys.stdout.flush()
        os.write(json_fd, (out + "\n").encode())
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
