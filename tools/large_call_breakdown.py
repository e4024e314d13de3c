"""Where the wall time of one uvs_large_solve_fused() call on the configs[3] window goes (run on the GPU box): host packing alone (uvs_debug_pack_layout,
same code path incl. the inner threads), the whole call, and the resident LM loop the call reports."""
import ctypes as C, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
uvs = importlib.import_module("uv-slam_amd")
w = uvs.synth.make_window(70, n_points=20000, n_lines=5000, n_tagged=3750)
L = uvs.api.lib()
o = uvs.abi.default_options()
wc, keep = w.to_c()
info = (C.c_int32 * 12)()
L.uvs_debug_pack_layout.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
os.environ["UVS_DEBUG_CHUNK_GRID"] = "255"
for nt in ("1", "4", "8", "16"):
    os.environ["UVS_PACK_THREADS"] = nt
    for _ in range(3): L.uvs_debug_pack_layout(C.byref(o), C.byref(wc), info)
    t = time.perf_counter()
    for _ in range(10): L.uvs_debug_pack_layout(C.byref(o), C.byref(wc), info)
    print("pack only, %2s inner threads: %.2f ms" % (nt, (time.perf_counter() - t) / 10 * 1e3))
del os.environ["UVS_PACK_THREADS"]
s = uvs.api.Solver(max_batch=1, max_points=20008, max_point_obs=240000, max_lines=5008, max_line_obs=60000)
s.large_comm_init(None)
for _ in range(3): s.large_solve_fused(w)
calls, loops = [], []
for _ in range(10):
    st, rep, ms = s.large_solve_fused(w); calls.append(s.last_solve_ms); loops.append(ms)
print("whole call %.2f ms (min %.2f), resident LM loop %.2f ms" % (sum(calls) / 10, min(calls), sum(loops) / 10))
os.environ["UVS_PACK_PROFILE"] = "1"
s.large_solve_fused(w)
s.close()
