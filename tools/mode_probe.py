"""Dev: the in-flight figure has two modes per process (0.233 / 0.245 ms per local step).  Does the mode change when the
pipeline (streams + graph instances) is rebuilt INSIDE one process?  Prints the median of five 20-step blocks per rebuild."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
dev = torch.device("cuda")
model = bench.build_model("basic_config", dev, seed=0, num_points=8192)
pts = bench.synthetic_clouds(8, 8192, 2002, dev)
keep = []
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
    with torch.no_grad():
        pipe = model.pipeline(pts, depth=4, outputs=("xyz_feat",), streams=streams)
        for _ in range(100):
            pipe.submit()
        torch.cuda.synchronize()
        blocks = []
        for _ in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20):
                pipe.submit()
            torch.cuda.synchronize(); blocks.append((time.perf_counter() - t0) / 20 * 1e3)
    print("rebuild %d: blocks %s  median %.4f ms/step" % (trial, np.round(blocks, 4).tolist(), float(np.median(blocks))), flush=True)
    if trial % 2 == 0:
        keep.append((pipe, streams))   # every other pipeline stays alive: the next one's streams land on other queues
