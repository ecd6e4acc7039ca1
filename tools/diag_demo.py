import numpy as np, torch, sys
sys.path.insert(0, "tests")
from test_demo_clouds import _weights
demo = np.load("tests/golden/demo_clouds.npz")
dev = torch.device("cuda")
for name in ("local_268", "local_642", "dso_9000"):
    c = demo[name]
    m, w = _weights("detection_config", demo)
    m.config.num_points = len(c)
    m = m.to(dev).eval().prepare()
    with torch.no_grad():
        outs = m(torch.from_numpy(c[None]).to(dev))
    got = outs["xyz_feat_att"][0].cpu().numpy()[::32]
    exp = demo[name + "/rows"]
    err = np.abs(got - exp)
    r, col = np.unravel_index(err.argmax(), err.shape)
    print(name, "max err", err.max(), "row", r, "col", col, "got", got[r, col], "exp", exp[r, col])
    print("  per-block max: xyz", err[:, :3].max(), "desc", err[:, 3:131].max(), "att", err[:, 131].max())
    feat = outs["feat"][0].cpu().numpy()[::32]
    print("  |feat| max", np.abs(feat).max(), " rows with desc err>5e-5:", int((err[:, 3:131].max(1) > 5e-5).sum()), " att err>5e-5:", int((err[:,131]>5e-5).sum()))
    if err[:, 131].max() > 5e-5:
        bad = np.argsort(-err[:, 131])[:5]
        print("  att bad rows", bad, got[bad, 131], exp[bad, 131])
