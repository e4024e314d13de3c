"""Dumps the gather-group assignment of the bench window (UVS_DEBUG_LISTS): per chunk and wave, block.part(Schur entries, direct entries).
Run by hand on the GPU box: python tests/gpu_debug_lists.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helpers import uvs, synth
s = uvs.api.Solver(max_batch=2)
w = synth.make_window(0, with_prior=True, marginalize_fn=lambda win, flag: s.marginalize(win, flag))
os.environ["UVS_DEBUG_LISTS"] = "1"
s.upload([w])
