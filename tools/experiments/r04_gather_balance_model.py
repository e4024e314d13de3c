import os, sys, re, subprocess, importlib, ctypes as C
sys.path.insert(0, '/root/repo')
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    os.environ['UVS_DEBUG_LISTS'] = '1'
    uvs = importlib.import_module('uv-slam_amd'); abi, synth, api = uvs.abi, uvs.synth, uvs.api
    import numpy as np
    lib = api.lib()
    seed = int(sys.argv[2])
    w = synth.make_window(seed, with_prior=False)
    wc, keep = w.to_c()
    info = (C.c_int32 * 16)()
    o = abi.default_options()
    rc = lib.uvs_debug_pack_layout(C.byref(o), C.byref(wc), info)
    print("rc", rc, list(info)[:12])
    sys.exit(0)
tot_ideal = tot_actual = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    p = subprocess.run([sys.executable, __file__, 'child', str(seed)], capture_output=True, text=True)
    chunks = []; cur = None
    for ln in p.stderr.splitlines():
        m = re.match(r'chunk (\d+) type (\d+)', ln)
        if m: cur = dict(type=int(m.group(2)), waves=[]); chunks.append(cur); continue
        m = re.match(r'\s+wave (\d+):(.*)', ln)
        if m and cur is not None:
            gs = [(int(a), int(b)) for a, b in re.findall(r'\((\d+),(\d+)\)', m.group(2))]
            cur['waves'].append(gs)
    ideal = actual = 0
    for c in chunks:
        ws_, wd_ = (18, 35) if c["type"] == 0 else (18, 63)
        tw = []; work = 0
        for gs in c['waves']:
            tw.append(max(ws_ * s for s, d in gs) + max(wd_ * d for s, d in gs))
            work += sum(ws_ * s + wd_ * d for s, d in gs)
        t = max(tw); n = sum(len(g) for g in c['waves'])
        ideal += work / n; actual += t
        print("seed %d chunk type %d: wave times %s, mean group work %.0f -> utilisation %.2f" % (seed, c['type'], tw, work / n, work / n / t))
    print("seed %d: gather model time %d, perfectly balanced %d -> utilisation %.2f" % (seed, actual, ideal, ideal / actual))
    tot_ideal += ideal; tot_actual += actual
print("overall utilisation %.3f" % (tot_ideal / tot_actual))
