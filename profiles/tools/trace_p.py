"""Where a wave of k_paths4 spends its cycles inside the pipelined column loop (heads_P): shader-clock stamps at the
segment boundaries of body(), summed over all waves (library built with -DQ_PIPE -DP_TRACE).
    FILE=stage_b profiles/tools/a_variants.sh build "P_TRACE"        (here)
    python profiles/tools/trace_p.py [lib]                             (GPU box)"""
import sys, os, ctypes as C, numpy as np, torch
os.environ["XMAP_HIP_LIB"] = sys.argv[1] if len(sys.argv) > 1 else "x-map_amd/_variants/libxmap_P_TRACE.so"
sys.path.insert(0, '.'); sys.path.insert(0, 'x-map_amd')
from xmap.engine import synth, device as dev
from xmap.engine.hipabi import lib
r = synth.config_c2(); attrs = r.item_attrs()
eng = dev.Engine(dev.DeviceRatings(r.user_ptr, r.item, r.rating, r.time, r.n_items, attrs))
S = eng.item_sim("adjust_cosine", 50)
lib.xmap_debug_ptrace.argtypes = [C.c_void_p, C.c_int]
names = ["outside the column loop (direct adds, finalisation, unit set-up) + loop overhead", "ends: wait e, stage through LDS", "rows requested, records prepared",
         "(after the fetch)", "record loop + slice exchange", "wait for the row entries", "flush (adds, stores)",
         "slow path (columns with several rounds / chunks)", "fetch: wait for the advanced heads' directory entries", "fetch: merge step + lane assignment",
         "fetch: the eight loads issued", "-"]
for it in range(3):
    lib.xmap_debug_ptrace(None, 1)
    eng.timers = {}
    E = eng.extend(S, 50); torch.cuda.synchronize()
    buf = np.zeros(16, np.uint64)
    assert lib.xmap_debug_ptrace(buf.ctypes.data, 0) == 0
    tot, waves = float(buf[12]), int(buf[13])
    print("pass %d: paths %.1f ms, %d waves, %.3g cycles per wave (s_memtime ticks)" % (it, eng.timer_ms()["paths"][0], waves, tot / max(waves, 1)))
    for i in range(11):
        print("   %-46s %5.1f %%" % (names[i], 100.0 * float(buf[i]) / tot))
    print("   %-46s %5.1f %%" % ("unaccounted", 100.0 * (tot - float(buf[:11].sum())) / tot))
