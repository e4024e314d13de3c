"""Malformed windows through every entry point of the C ABI (run on the GPU box; tests/test_gpu_edge_cases.py runs it in a child process so that a crash
fails ONE test instead of ending the session): each call must come back with an error status -- never a crash, never UVS_OK -- and the handle must
still solve a good window afterwards.   python tests/gpu_fuzz_bad_windows.py"""
import sys, os, re
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from helpers import uvs, abi, synth

good = synth.make_window(0, n_points=40, n_lines=12, n_tagged=6)
s_marg = uvs.api.Solver(max_batch=2)
good_p = synth.make_window(1, n_points=40, n_lines=12, n_tagged=6, with_prior=True, marginalize_fn=lambda win, flag: s_marg.marginalize(win, flag))
good_r = synth.add_relocalization(synth.make_window(2, n_points=40, n_lines=12, n_tagged=6), seed=2)

def mutants():
    def m(name, base, fn):
        w = base.copy(); fn(w); return name, w
    yield m("pt_lm out of range", good, lambda w: w.pt_lm.__setitem__(3, 10 ** 6))
    yield m("pt_lm negative", good, lambda w: w.pt_lm.__setitem__(0, -1))
    yield m("pt_lm decreasing", good, lambda w: w.pt_lm.__setitem__(len(w.pt_lm) - 1, 0))
    yield m("pt_fj == NUM_FRAMES", good, lambda w: w.pt_fj.__setitem__(5, abi.NUM_FRAMES))
    yield m("pt_fj huge", good, lambda w: w.pt_fj.__setitem__(5, 2 ** 30))
    yield m("pt_fi >= pt_fj", good, lambda w: w.pt_fi.__setitem__(0, int(w.pt_fj[0])))
    yield m("pt_fi negative", good, lambda w: w.pt_fi.__setitem__(0, -5))
    yield m("anchor changes inside a landmark", good, lambda w: w.pt_fi.__setitem__(1, int(w.pt_fi[1]) + 1) if w.pt_lm[0] == w.pt_lm[1] else w.pt_lm.__setitem__(0, -1))
    yield m("ln_lm out of range", good, lambda w: w.ln_lm.__setitem__(2, 10 ** 6))
    yield m("ln_fj out of range", good, lambda w: w.ln_fj.__setitem__(2, 99))
    yield m("ln_fj negative", good, lambda w: w.ln_fj.__setitem__(2, -1))
    yield m("ln_fj repeated", good, lambda w: w.ln_fj.__setitem__(1, int(w.ln_fj[0])))
    yield m("imu frame_i out of range", good, lambda w: w.imu[0].__setitem__("frame_i", abi.NUM_FRAMES - 1))
    yield m("imu frame_i negative", good, lambda w: w.imu[0].__setitem__("frame_i", -2))
    yield m("relo_lm decreasing", good_r, lambda w: setattr(w, "relo_lm", w.relo_lm[::-1].copy()))
    yield m("relo_lm out of range", good_r, lambda w: w.relo_lm.__setitem__(len(w.relo_lm) - 1, 10 ** 6))
    yield m("prior n too large", good_p, lambda w: setattr(w, "prior", _prior(w.prior, n=10 ** 6)))
    yield m("prior n_blocks too large", good_p, lambda w: setattr(w, "prior", _prior(w.prior, n_blocks=10 ** 6)))
    yield m("prior block frame out of range", good_p, lambda w: setattr(w, "prior", _prior(w.prior, frame0=77)))
    yield m("prior block idx out of range", good_p, lambda w: setattr(w, "prior", _prior(w.prior, idx0=10 ** 6)))
    yield m("non-finite state", good, lambda w: w.pose.__setitem__((2, 0), float("nan")))

def _prior(p, n=None, n_blocks=None, frame0=None, idx0=None):
    q = p.copy()
    if n is not None: q.n = n
    if n_blocks is not None: q.n_blocks = n_blocks
    if frame0 is not None: q.block_frame[0] = frame0
    if idx0 is not None: q.block_idx[0] = idx0
    return q

def entry_points(s):
    yield "uvs_solve_window", lambda w: s.solve(w)
    yield "uvs_batch_upload", lambda w: s.upload([good, w])
    yield "uvs_batch_stream", lambda w: s.stream([good, w, good, good], 2)
    yield "uvs_evaluate", lambda w: s.evaluate(w)
    yield "uvs_marginalize(0)", lambda w: s.marginalize(w, 0)
    yield "uvs_marginalize(1)", lambda w: s.marginalize(w, 1)
    yield "uvs_large_solve (step-wise)", lambda w: s.large_solve(w)
    yield "uvs_large_solve_fused", lambda w: s.large_solve_fused(w)

s = uvs.api.Solver(max_batch=4)
bad = 0; calls = 0
for name, w in mutants():
    for ep, fn in entry_points(s):
        calls += 1
        if os.environ.get("UVS_FUZZ_VERBOSE"): print("...", name, "|", ep, flush=True)
        try:
            out = fn(w)
        except RuntimeError as e:
            if not re.search(r"uvs error -?\d+", str(e)): bad += 1; print("UNEXPECTED EXCEPTION", name, ep, e)
            continue
        except Exception as e:
            bad += 1; print("UNEXPECTED EXCEPTION TYPE", name, ep, type(e), e); continue
        # a call that returns must not have claimed success on garbage: the only mutant that may pass validation is the non-finite state, which has to end as a failed solve
        if name == "non-finite state":
            rep = out[1] if isinstance(out, tuple) and len(out) > 1 and hasattr(out[1], "termination") else None
            if rep is not None and rep.status == 0 and rep.termination not in (5, 6) and np.isfinite(rep.final_cost): bad += 1; print("NON-FINITE STATE SOLVED?", ep, rep.termination, rep.final_cost)
        elif ep in ("uvs_marginalize(0)", "uvs_marginalize(1)", "uvs_evaluate") and name.startswith("relo_lm"):
            pass      # relocalization blocks are solve-only: these entry points do not read them
        else:
            bad += 1; print("ACCEPTED", name, ep)
    st, rep = s.solve(good)      # the handle survives
    if rep.status != 0 or not np.isfinite(rep.final_cost): bad += 1; print("HANDLE BROKEN AFTER", name)
print("%d calls with malformed windows, %d wrong outcomes" % (calls, bad))
sys.exit(1 if bad else 0)
