"""Dev: per-kernel instruction statistics of hipcc -S output (/tmp/isa/*.s): global loads, full memory waits
(s_waitcnt vmcnt(0)), scratch use, 64-bit integer division sequences -- the quick screen for serialised loads."""
import re, sys
for path in sys.argv[1:]:
    txt = open(path).read()
    for m in re.finditer(r"^(_Z\w+):.*?\n(.*?)s_endpgm", txt, re.S | re.M):
        name, body = m.group(1), m.group(2)
        dem = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", name)[:70]
        n = len(body.splitlines())
        gl = len(re.findall(r"\bglobal_load|\bbuffer_load", body))
        gs = len(re.findall(r"\bglobal_store|\bbuffer_store", body))
        w0 = len(re.findall(r"s_waitcnt vmcnt\(0\)", body))
        wall = len(re.findall(r"s_waitcnt.*vmcnt", body))
        scr = len(re.findall(r"scratch_", body))
        div = len(re.findall(r"v_rcp_iflag_f32|v_mul_hi_u32", body))
        mf = len(re.findall(r"v_mfma", body))
        bar = len(re.findall(r"s_barrier", body))
        print("%-72s lines %6d gload %4d gstore %4d vmcnt0 %3d/%3d scratch %3d div %3d mfma %4d barrier %3d" % (dem, n, gl, gs, w0, wall, scr, div, mf, bar))
