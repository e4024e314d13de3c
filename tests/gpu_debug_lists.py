import sys, os
sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo')
from helpers import uvs, synth
s = uvs.api.Solver(max_batch=2)
w = synth.make_window(0, with_prior=True, marginalize_fn=lambda win, flag: s.marginalize(win, flag))
os.environ["UVS_DEBUG_LISTS"] = "1"
s.upload([w])
