"""Dev: what a serving loop around engine.Pipeline costs, piece by piece (local workload, four slots)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, bench
dev = torch.device("cuda")
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "local"]
depth = wl["inflight"]
model = bench.build_model(wl["preset"], dev, seed=0, num_points=wl["N"])
pts = bench.synthetic_clouds(wl["B"], wl["N"], wl["seed"], dev, 0)
name = wl["out"]
with torch.no_grad():
    pipe = model.pipeline(pts, depth=depth, outputs=(name,))
    shape = tuple(pipe._runs[0].outputs[name].shape)
    host = [torch.from_numpy(np.random.default_rng(i).random((wl["B"], wl["N"], 3), dtype=np.float32)).pin_memory() for i in range(2 * depth + 1)]
    print("pinned:", host[0].is_pinned(), "out shape", shape)
    devb = [h.to(dev) for h in host]
    sink = [torch.empty(shape, device=dev) for _ in range(depth)]
    hout = [torch.empty(shape).pin_memory() for _ in range(depth)]
    small = [torch.empty((shape[0], 64)).pin_memory() for _ in range(depth)]

    def timed(fn, n=40, warm=8):
        for i in range(warm): fn(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n): fn(i)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    def zero(i): pipe.submit()
    def h2d(i): pipe.submit(host[i % len(host)])
    def h2d_manual(i):
        k = pipe.next_slot
        with torch.cuda.stream(pipe.stream(k)):
            pipe.input_buffer(k).copy_(host[i % len(host)], non_blocking=True)
        pipe.submit()
    def d2d_in(i): pipe.submit(devb[i % len(devb)])
    def d2d_in_nowait(i): pipe.submit(devb[i % len(devb)], after_current=False)
    def sink_copy_plain(i):
        k = pipe.next_slot
        pipe.submit()
        with torch.cuda.stream(pipe.stream(k)):
            sink[k].copy_(pipe._runs[k].outputs[name], non_blocking=True)
    def h2d_sink_stage(i):
        k = pipe.next_slot
        pipe.submit(host[i % len(host)], fetch_to={name: sink[k]})
    def sink_copy(i):
        k = pipe.next_slot
        pipe.submit(fetch_to={name: sink[k]})
    def sink_kernel(i):
        k = pipe.next_slot
        pipe.submit()
        with torch.cuda.stream(pipe.stream(k)):
            torch.mul(pipe._runs[k].outputs[name], 1.0, out=sink[k])
    def d2h_small(i):
        k = pipe.next_slot
        pipe.submit()
        with torch.cuda.stream(pipe.stream(k)):
            small[k].copy_(pipe._runs[k].outputs[name].reshape(shape[0], -1)[:, :64], non_blocking=True)
    def d2h_full(i):
        k = pipe.next_slot
        pipe.submit(fetch_to={name: hout[k]})
    def h2d_sink_kernel(i):
        k = pipe.next_slot
        pipe.submit(host[i % len(host)])
        with torch.cuda.stream(pipe.stream(k)):
            torch.mul(pipe._runs[k].outputs[name], 1.0, out=sink[k])
    for nm, fn in (("zero-copy submit()", zero), ("H2D pinned via submit(host)", h2d), ("H2D pinned, manual copy_ on slot stream", h2d_manual),
                   ("device batch via submit(dev) [waits current]", d2d_in), ("device batch, after_current=False", d2d_in_nowait),
                   ("zero-copy + fetch_to device sink (copy_)", sink_copy), ("zero-copy + sink by a kernel (mul out=)", sink_kernel),
                   ("zero-copy + D2H of a small slice", d2h_small), ("zero-copy + D2H of the whole output", d2h_full),
                   ("H2D + sink by kernel", h2d_sink_kernel), ("H2D + fetch_to device sink (staging kernels both ways)", h2d_sink_stage), ("zero-copy + torch copy_ sink", sink_copy_plain), ("zero-copy submit() again", zero)):
        try:
            print("%-52s %.4f ms per step" % (nm, timed(fn)), flush=True)
        except Exception as e:
            print(nm, "ERROR", repr(e)[:200])
