"""Two landmark shards stepped in lockstep inside ONE process (manual all-reduce) vs the single-shard large solve."""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helpers import uvs, abi, synth, pose_deltas
import torch
api = uvs.api
L = api.lib()
w = synth.make_window(53, n_points=400, n_lines=100, n_tagged=75)
s = api.Solver(max_batch=2)
st, rep = s.large_solve(w)
print("single:", rep.num_iterations, list(rep.accepted[:11]), rep.final_cost)
G = 2
shards = [synth.shard_landmarks(w, r, G)[0] for r in range(G)]
sol = [api.Solver(max_batch=2) for _ in range(G)]
keep = [sh.to_c() for sh in shards]
for r in range(G):
    assert L.uvs_large_begin(sol[r]._h, C.byref(keep[r][0])) == 0
x2 = sum(L.uvs_large_local_x2(sol[r]._h) for r in range(G))
for r in range(G): L.uvs_large_set_landmark_x2(sol[r]._h, x2)
n = C.c_int(0)
red = [api._device_tensor(L.uvs_large_reduced(sol[r]._h, C.byref(n)), n.value) for r in range(G)]
nr = n.value
scal = [api._device_tensor(L.uvs_large_scalars(sol[r]._h, C.byref(n)), n.value) for r in range(G)]
it = 0
while not L.uvs_large_done(sol[0]._h):
    if L.uvs_large_need_linearize(sol[0]._h):
        for r in range(G): assert L.uvs_large_linearize(sol[r]._h) == 0
        torch.cuda.synchronize()
        tot = sum(t.clone() for t in red); mx = torch.max(torch.stack([t[-7] for t in red]))
        for t in red: t.copy_(tot); t[-7] = mx
        torch.cuda.synchronize()
        print("  reduced checksum", float(tot.sum()), "alias check", float(red[0].sum()))
    for r in range(G): assert L.uvs_large_step(sol[r]._h) == 0
    torch.cuda.synchronize()
    tot = sum(t.clone() for t in scal)
    for t in scal: t.copy_(tot)
    torch.cuda.synchronize()
    for r in range(G): assert L.uvs_large_decide(sol[r]._h) == 0
    it += 1
outs = []
for r in range(G):
    stt = abi.State(len(shards[r].inv_depth), len(shards[r].line_orth)); sc = stt.alloc_c(); rp = abi.Report()
    L.uvs_large_finish(sol[r]._h, C.byref(sc), C.byref(rp)); stt.from_c(sc); outs.append((stt, rp))
    print("rank", r, rp.num_iterations, list(rp.accepted[:11]), rp.final_cost, "pose delta", pose_deltas(stt.pose, st.pose))
    print("   mcc", [float(rp.model_cost_change[k]) for k in range(1, 6)], "single", [float(rep.model_cost_change[k]) for k in range(1, 6)])
