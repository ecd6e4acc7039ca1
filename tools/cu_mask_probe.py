"""Dev: steps in flight with every slot's stream confined to its own part of the chip (hipExtStreamCreateWithCUMask): do the
latency-bound kernels of a step then run beside another step's instead of behind them?  Local workload, four slots."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, bench
hip = ctypes.CDLL("libamdhip64.so")
dev = torch.device("cuda")
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "local"]
depth = wl["inflight"]

def masked_stream(words):
    arr = (ctypes.c_uint32 * len(words))(*words)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), len(words), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value, device=dev)

def run(streams, tag):
    model = bench.build_model(wl["preset"], dev, seed=0, num_points=wl["N"])
    pts = bench.synthetic_clouds(wl["B"], wl["N"], wl["seed"], dev, 0)
    with torch.no_grad():
        pipe = model.pipeline(pts, depth=depth, outputs=(wl["out"],), streams=streams)
        for _ in range(40): pipe.submit()
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(5):
            t0 = time.perf_counter()
            for _ in range(100): pipe.submit()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 100)
    print("%-44s %.4f ms per step" % (tag, best * 1e3), flush=True)

run(None, "plain streams")
# 256 CUs = 8 words of 32 bits; how the bits map to XCDs / CUs is the driver's: try interleaved and blocked quarters
full = [0xFFFFFFFF] * 8
run([masked_stream(full) for _ in range(depth)], "masked streams, full masks")
blocked = [[0xFFFFFFFF if w // 2 == k else 0 for w in range(8)] for k in range(4)]
run([masked_stream(blocked[k % 4]) for k in range(depth)], "blocked quarters (64 CUs each)")
inter = [[(0x11111111 << k) & 0xFFFFFFFF for w in range(8)] for k in range(4)]
run([masked_stream(inter[k % 4]) for k in range(depth)], "interleaved quarters (every 4th CU)")
halves = [[0xFFFFFFFF if (w // 4) == (k % 2) else 0 for w in range(8)] for k in range(4)]
run([masked_stream(halves[k % 4]) for k in range(depth)], "blocked halves (128 CUs, two slots each)")
