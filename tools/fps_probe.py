import ctypes, torch, numpy as np
lib = ctypes.CDLL("tools/libfps_probe.so")
dev = torch.device("cuda")
xyz = torch.rand(8, 8192, 3, device=dev)
out = torch.empty(8, 1024, dtype=torch.int32, device=dev)
for _ in range(2):
    lib.dh3d_farthest_point_sample(8, 8192, 1024, ctypes.c_void_p(xyz.data_ptr()), None, ctypes.c_void_p(out.data_ptr()), None)
torch.cuda.synchronize()
h = (ctypes.c_longlong * 16)()
lib.dh3d_fps_probe_read(h)
t = list(h)[:7]
print("stamps (cycles from round start):", [x - t[0] for x in t])
print("phases: coord-read %d, update %d, wave_max %d, key+ballot %d, lds-write+barrier %d, cross-wave %d" % tuple(t[i + 1] - t[i] for i in range(6)))
