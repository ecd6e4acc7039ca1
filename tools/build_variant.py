"""Dev: build a variant of libdh3d_hip.so with extra -D flags into tools/lib<name>.so without touching the product's objects.
    python tools/build_variant.py <name> "<extra flags>" [file.hip ...]      (flags apply to the listed files, default: all)
Objects of files WITHOUT extra flags are taken from dh3d_amd/csrc/*.o (build the product first); flagged files are compiled
into /tmp/dh3d_variant_<name>/.  Use with DH3D_HIP_LIB=tools/lib<name>.so (dh3d_amd/_lib.py)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "dh3d_amd", "csrc")
EXACT = {"knn", "fps", "pointnet2", "spatial"}
EXTRA = {"fps": ["-fno-honor-nans", "-mno-amdgpu-ieee"], "flex_x6": ["-fno-slp-vectorize"]}
name, flags = sys.argv[1], sys.argv[2].split()
files = [f[:-4] for f in os.listdir(CSRC) if f.endswith(".hip")]
flagged = [f[:-4] if f.endswith(".hip") else f for f in sys.argv[3:]] or files
tmp = "/tmp/dh3d_variant_%s" % name
os.makedirs(tmp, exist_ok=True)
base = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall",
        "-Wno-unused-function", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
procs, objs = [], []
for f in files:
    if f in flagged:
        o = os.path.join(tmp, f + ".o")
        cmd = base + (["-ffp-contract=off"] if f in EXACT else []) + EXTRA.get(f, []) + flags + ["-c", os.path.join(CSRC, f + ".hip"), "-o", o]
        procs.append(subprocess.Popen(cmd, cwd=CSRC))
    else:
        o = os.path.join(CSRC, f + ".o")
    objs.append(o)
assert all(p.wait() == 0 for p in procs)
out = os.path.join(ROOT, "tools", "lib%s.so" % name)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out])
print(out)
