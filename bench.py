#!/usr/bin/env python
"""bench.py -- sliding-window solves/s on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path (the persistent LM solve kernel) over one
batch of synthetic W10-P150-L40-V3 windows that is ALREADY RESIDENT in HBM
(uvs_batch_upload is outside the timed region; the PCIe-inclusive single-window
rate is reported separately and in DESIGN.md).  With --gpus N every rank owns
its own batch (independent windows => replicas, no data-path collective,
"scaling": "weak"); timing is barrier + synchronize on both sides, max over ranks.

Rank 0 prints ONE JSON line with the contract fields plus `roofline` and
`cpu_baseline` (the CPU oracle -- a port of the reference solve -- timed on a
bounded sample of the same windows on this box's host cores: one thread, which
is what the reference gives Ceres (estimator.cpp:985 leaves num_threads = 1), and
one window per core as the throughput comparator; built -O3 -march=native ON
THE BOX for this leg so that the ratio is not quoted against a slow build).

Also in the line (never `value`): `single_window` = BASELINE configs[1] (one
resident window, its own roofline fraction, the PCIe-inclusive rate),
`batch_pack_upload_ms` (host packing + H2D of the 256 windows, outside the timed
region), `large_window` = configs[3] (20 000 points + 5 000 lines) through the
fused loop -- with --gpus N the landmarks are sharded k mod N over the ranks and
the reduced system is all-reduced over RCCL by the library's own communicator.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6      # MI355X FP64 vector = FP64 matrix (datasheet; not listed in MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


K_SOLVE_SOURCES = ("uvs_solve_kernel.h", "uvs_factors.h", "uvs_layout.h", "uvs_solver.hip", "uvs_solve512.hip")      # k_solve's device code + the packing that lays its inputs out


def kernel_source_tag():
    """sha1 over the sources that determine k_solve's memory traffic (its device code and pack_window): profiles/pmc_traffic.json carries the
    tag of the build it was measured on (profiles/summarize.py).  The marginalization / evaluation / large-window sources are left out: they
    do not run in the profiled launch."""
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(ROOT, "uv-slam_amd", "csrc")
    for f in K_SOLVE_SOURCES:
        h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def large_source_tag():
    """Like kernel_source_tag(), for the landmark-sharded kernels (their own header on top of the shared device code)."""
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(ROOT, "uv-slam_amd", "csrc")
    for f in K_SOLVE_SOURCES + ("uvs_large_kernel.h",):
        h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def large_traffic(n_iterations):
    """HBM-side bytes of one configs[3] solve = PMC bytes per LM iteration (profiles/collect.sh: FETCH_SIZE / WRITE_SIZE passes over the k_large_* kernels,
    profiles/summarize.py) x iterations -- only while the kernel sources still hash to the profiled build (the large-window kernels share the tag's file list)."""
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic_large.json")
    try:
        tj = json.load(open(tpath))
        if tj.get("kernel_source_tag") == large_source_tag():
            return float(tj["hbm_bytes_per_iteration"]) * n_iterations
    except Exception:
        pass
    return None


def algorithmic_flops(w, n_iterations, n_successful=None):
    """SURVEY.md section 8d flop model: (1 + it) linearizations + it cost-only evaluations."""
    n_po, n_lo = len(w.pt_lm), len(w.ln_lm)
    n_vp = int(np.sum(w.ln_has_vp))
    n_imu = len(w.imu)
    n = w.prior.n if w.prior is not None else 0
    D = 165
    acc = lambda r, c: 2 * r * c + r * c * (c + 1)
    ev = 600 * n_po + 800 * n_lo + 400 * n_vp + (2 * 15 * 15 * 30 + 2000) * n_imu + 4 * n * n
    ac = acc(2, 13) * n_po + acc(2, 10) * n_lo + acc(1, 10) * n_vp + acc(15, 30) * n_imu
    def schur(d, q):
        return d ** 3 / 3 + 2 * q * d * d + q * (q + 1) * d + 2 * q * d
    # frames touched per landmark from the CSR structure
    sch = 0.0
    pt_cnt = np.bincount(w.pt_lm, minlength=len(w.inv_depth)) if n_po else np.zeros(0)
    for c in pt_cnt:
        sch += schur(1, 6 * (int(c) + 1))
    ln_cnt = np.bincount(w.ln_lm, minlength=len(w.line_orth)) if n_lo else np.zeros(0)
    for c in ln_cnt:
        sch += schur(4, 6 * int(c))
    chol = D ** 3 / 3 + 2 * D * D
    lin = ev + ac + sch + chol
    cost_only = 0.3 * (600 * n_po + 800 * n_lo + 400 * n_vp) + 2000 * n_imu + 2 * n * n
    if n_successful is not None:
        # what the kernel EXECUTES since the re-damping change: factor evaluation + Schur products only at the start and after accepted steps; after a
        # rejected step the stored linearization is re-damped (the Schur accumulation and the factorization run again, the factors do not)
        n_lin = 1 + n_successful
        return n_lin * (ev + ac) + (1 + n_iterations) * (sch + chol) + n_iterations * cost_only
    return (1 + n_iterations) * lin + n_iterations * cost_only


def project_large_window(measured_n):
    """UNMEASURED projection of the configs[3] LM iteration on 2 / 4 / 8 GPUs from the per-kernel times of the committed 1-GPU rocprofv3 trace
    (the newest profiles/rNN_kernel_stats_large.csv): the landmark-sharded kernels divide by N, the reduced solve and the trust-region step are replicated, and every
    iteration pays two in-place RCCL all-reduces (40 KB and 64 B: latency bound, 15 us each ASSUMED -- no multi-GPU node was available to measure them)."""
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_kernel_stats_large.csv")))
    if not cands:
        return None
    path = cands[-1]
    import csv
    avg = {}
    for row in csv.DictReader(open(path)):
        for key in ("k_large_chunks", "k_large_backsub", "k_large_reduce", "k_large_solve", "k_large_decide"):
            if "::" + key + "(" in row["Name"]:      # (uvsdev:: or uvsdev512::, whichever instantiation the profiled build launched)
                avg[key] = float(row["AverageNs"]) * 1e-3
    if len(avg) < 5:
        return None
    sharded = avg["k_large_chunks"] + avg["k_large_backsub"] + avg["k_large_reduce"]
    replicated = avg["k_large_solve"] + avg["k_large_decide"]
    allreduce_us = 15.0
    out = {"status": "UNMEASURED projection from 1-GPU kernel times (profiles/%s); the all-reduce latency is an assumption" % os.path.basename(path),
           "per_iteration_us_1gpu": {"sharded (chunks + backsub + reduce)": sharded, "replicated (reduced solve + decide)": replicated},
           "assumed_allreduce_us_each": allreduce_us, "iteration_us": {}, "speedup_vs_1gpu": {}}
    t1 = sharded + replicated
    for n in (1, 2, 4, 8):
        t = sharded / n + replicated + (2 * allreduce_us if n > 1 else 0.0)
        out["iteration_us"][str(n)] = t; out["speedup_vs_1gpu"][str(n)] = t1 / t
    out["measured_on_n_gpus"] = measured_n
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="independent windows per GPU per step (BASELINE configs[2])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="windows timed on the CPU oracle (0 = auto ~15 s)")
    ap.add_argument("--no-prior", action="store_true")
    ap.add_argument("--no-stream", action="store_true", help="skip the end-to-end stream of fresh batches (profiling runs)")
    ap.add_argument("--no-replay", action="store_true", help="skip the closed-loop sequence replay (profiling runs)")
    ap.add_argument("--no-large", action="store_true", help="skip the configs[3] large-window timing (profiling runs)")
    ap.add_argument("--no-scale-point", action="store_true", help="skip the 200 000-point window of the large_window leg (not a BASELINE config; ~30 s of generation and packing)")
    ap.add_argument("--no-fused-single", action="store_true", help="skip the multi-workgroup form of the single window (keeps a kernel trace of the configs[3] leg clean)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0")); local_rank = int(os.environ.get("LOCAL_RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    import torch
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # one rank per GPU over RCCL (backend "nccl" on ROCm).  UVS_BENCH_BACKEND=gloo is a dry-run hook for a box with fewer GPUs than
        # ranks: the ranks then share the visible devices round-robin and the barrier / max-reduce go through gloo on the host.
        backend = os.environ.get("UVS_BENCH_BACKEND", "nccl")
        local_rank = local_rank % max(torch.cuda.device_count(), 1) if backend != "nccl" else local_rank
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    uvs = importlib.import_module("uv-slam_amd")
    synth, abi, api = uvs.synth, uvs.abi, uvs.api

    solver = api.Solver(device=local_rank, max_batch=max(args.batch, 1))
    # ---- synthetic inputs (seed = 1000 + global window index); prior from the PRODUCT's own marginalization
    marg = None if args.no_prior else (lambda win, flag: solver.marginalize(win, flag))
    windows = [synth.make_window(rank * args.batch + i, with_prior=marg is not None, marginalize_fn=marg) for i in range(args.batch)]
    solver.upload(windows)                    # host packing + H2D: OUTSIDE the timed region, reported as batch_pack_upload_ms
    pack_upload_ms = solver.last_upload_ms    # uvs_batch_upload() alone (pack_window on the host threads + one H2D copy), without the ctypes conversion

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    for _ in range(args.warmup):
        solver.solve_resident()
    sync()
    t0 = time.perf_counter()
    kernel_ms = []
    for _ in range(args.steps):
        kernel_ms.append(solver.solve_resident())        # HIP events on the solver's own stream around the launch
    sync()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    states, reps = solver.download()
    n_total = args.batch * world * args.steps
    value = n_total / elapsed

    # ---- end to end (never `value`): a STREAM of fresh batches through uvs_batch_stream -- host packing (threads), H2D, solve, D2H of consecutive
    # batches overlapped on two buffer sets.  The same 256 windows are packed and uploaded again for every batch (nothing is cached between batches).
    end_to_end = None
    if rank == 0 and not args.no_stream:
        nb = 32
        try:
            _, sreps, _ = solver.stream(windows * 6, args.batch, want_states=False)      # warm-up: every buffer set twice (the first batch of a set sizes its pinned staging buffer, the second packs in place)
            walls = []
            for _ in range(5):      # five streams, the median reported (the packing threads share the host with whatever else runs on it: single streams scatter by +-10 %,
                _, sreps, wall_ms = solver.stream(windows * nb, args.batch, want_states=False)      # and a busy second of the host took two of three streams in one round-6 run)
                walls.append(wall_ms)
            wall_ms = sorted(walls)[2]
            ok = all(r.status == 0 for r in sreps) and all(sreps[i].final_cost == reps[i % args.batch].final_cost for i in range(len(sreps)))
            end_to_end = {"what": "uvs_batch_stream: %d batches of %d windows, packing + H2D + solve + results overlapped (three buffer sets); wall time of the C-ABI call" % (nb, args.batch),
                          "solves_per_s": nb * args.batch / (wall_ms * 1e-3), "ms_per_batch": wall_ms / nb, "solves_per_s_of_the_streams": [nb * args.batch / (w * 1e-3) for w in walls], "bitwise_equal_to_resident_solves": bool(ok),
                          "serial_reference": {"what": "upload (pack + H2D) then solve then download of ONE batch, nothing overlapped",
                                               "ms_per_batch": None}}
            solver.upload(windows); up = solver.last_upload_ms; kms = solver.solve_resident(); solver.download()      # C-ABI call times only (no ctypes conversion)
            end_to_end["serial_reference"]["ms_per_batch"] = up + kms + solver.last_download_ms; end_to_end["serial_reference"]["pack_upload_ms"] = up
        except Exception as e:
            end_to_end = {"error": repr(e)}

    # ---- BASELINE configs[3]: ONE window with 20 000 points + 5 000 lines, landmarks sharded k mod N over the ranks (all ranks take part)
    large = None
    if not args.no_large:
        wl = synth.make_window(70, n_points=20000, n_lines=5000, n_tagged=3750)
        shard = synth.shard_landmarks(wl, rank, world)[0] if world > 1 else wl
        sl = api.Solver(device=local_rank, max_batch=1, max_points=20008, max_point_obs=240000, max_lines=5008, max_line_obs=60000)
        try:
            sl.large_comm_init(dist if world > 1 and dist.get_backend() == "nccl" else None)
            comm_error = None
        except Exception as e:      # e.g. no RCCL in the process: every rank fails here alike, the headline line must not die with this leg
            comm_error = repr(e)
        if comm_error is not None:
            large = {"error": "uvs_large_comm_init: " + comm_error}
        elif world > 1 and dist.get_backend() != "nccl":
            large = {"skipped": "the fused loop all-reduces over RCCL; dry runs on another backend skip it"}
        else:
            loop_ms, wall_ms, cold_ms = [], [], []
            for rep_i in range(7):
                # calls 1..3: the window's STRUCTURE is new to the handle every time (UVS_NO_PACK_CACHE: chunking, work split, gather lists rebuilt, the whole
                # blob uploaded); calls 4..6: same structure as the call before (only the value sections are rewritten and uploaded)
                if rep_i < 4: os.environ["UVS_NO_PACK_CACHE"] = "1"
                else: os.environ.pop("UVS_NO_PACK_CACHE", None)
                sync()
                stl, repl, ms = sl.large_solve_fused(shard)
                wall = sl.last_solve_ms                      # the C-ABI call alone: pack + H2D + LM loop + D2H
                if rep_i >= 1 and rep_i < 4: cold_ms.append(max_over_ranks(wall))      # (the first pass warms the code objects / the communicator)
                if rep_i >= 5: loop_ms.append(max_over_ranks(ms)); wall_ms.append(max_over_ranks(wall))
            os.environ.pop("UVS_NO_PACK_CACHE", None)
            fl = algorithmic_flops(wl, int(repl.num_iterations))
            lm = float(np.median(loop_ms))
            large = {"workload": f"configs[3]: 10-KF window, 20000 points / 100000 obs, 5000 lines / 35000 obs, landmarks sharded k mod {world} over {world} GPU(s), "
                                 "reduced system all-reduced over RCCL by the library's communicator" if world > 1 else
                                 "configs[3]: 10-KF window, 20000 points / 100000 obs, 5000 lines / 35000 obs, 1 GPU (fused loop: control on the device)",
                     "n_gpus": world, "lm_iterations": int(repl.num_iterations), "final_cost": float(repl.final_cost),
                     "resident_lm_loop_ms": lm, "wall_ms_pack_upload_loop_download": float(np.median(cold_ms)), "wall_ms_same_structure_cached": float(np.median(wall_ms)),
                     "wall_note": "uvs_large_solve_fused() call alone (host packing + H2D + LM loop + D2H): wall_ms_pack_upload_loop_download = every call packs chunking / work split / gather lists afresh and "
                                  "uploads the whole blob (the meaning this key had in rounds 1 - 2; round 3 reported the cached case under it and this number as wall_ms_new_structure); "
                                  "wall_ms_same_structure_cached = a window whose index structure equals the previous call's (per-handle structure cache: values rewritten and uploaded only)",
                     "solves_per_s_resident": 1e3 / lm,
                     "collectives_per_iteration": 2 if world > 1 else 0, "allreduce_payload_bytes": [5016 * 8, 64],
                     "roofline": {"bound": "mfma", "kernels": "uvsdev::k_large_chunks / k_large_solve / k_large_backsub", "achieved": fl / (lm * 1e-3) / 1e12,
                                  "peak": FP64_PEAK_TFLOPS * world, "unit": "TFLOP/s", "frac": fl / (lm * 1e-3) / 1e12 / (FP64_PEAK_TFLOPS * world),
                                  "algorithmic_flops_per_solve": fl, "algorithmic_bytes_per_solve": float(synth.algorithmic_bytes(wl)), "traffic": large_traffic(int(repl.num_iterations)),
                                  "note": "whole resident LM loop (all kernels + collectives) against N x the FP64 roof; SURVEY.md 8d flop model"}}
            large["projected"] = project_large_window(world)
        sl.close()
        # ---- NOT a BASELINE config: the same window shape with TEN times the landmarks (200 000 points, 50 000 lines), the size from which the builder's own model says
        # landmark sharding pays (at configs[3]'s size the replicated one-workgroup reduced solve is as long as the sharded work: DESIGN.md section 6).  One GPU measures the
        # loop; N ranks shard it like configs[3]; the 2 / 4 / 8-GPU figures next to a 1-GPU run are a PROJECTION (sharded part / N + the replicated part of the committed
        # configs[3] profile + two assumed all-reduces).
        if large is not None and "error" not in large and "skipped" not in large and not args.no_scale_point:
            try:
                wb = synth.make_window(71, n_points=200000, n_lines=50000, n_tagged=37500)
                shard_b = synth.shard_landmarks(wb, rank, world)[0] if world > 1 else wb
                sb = api.Solver(device=local_rank, max_batch=1, max_points=200016, max_point_obs=1200000, max_lines=50016, max_line_obs=400000)
                sb.large_comm_init(dist if world > 1 and dist.get_backend() == "nccl" else None)
                lms = []
                for rep_i in range(3):
                    sync()
                    _, repb, ms = sb.large_solve_fused(shard_b)
                    if rep_i >= 1: lms.append(max_over_ranks(ms))
                sb.close()
                lmb = float(np.median(lms)); itb = max(int(repb.num_iterations), 1)
                flb = algorithmic_flops(wb, int(repb.num_iterations))
                big = {"workload": f"NOT a BASELINE config: 10-KF window, 200000 points / 1000000 obs, 50000 lines / 350000 obs, landmarks sharded k mod {world} over {world} GPU(s)",
                       "n_gpus": world, "lm_iterations": int(repb.num_iterations), "final_cost": float(repb.final_cost), "resident_lm_loop_ms": lmb,
                       "roofline_frac": flb / (lmb * 1e-3) / 1e12 / (FP64_PEAK_TFLOPS * world)}
                pr = large.get("projected")
                if world == 1 and pr:
                    rep_us = pr["per_iteration_us_1gpu"]["replicated (reduced solve + decide)"]; ar = pr["assumed_allreduce_us_each"]
                    per_it = lmb * 1e3 / itb; shard_us = max(per_it - rep_us, 0.0)
                    big["projected"] = {"status": "UNMEASURED projection: sharded part (measured loop / iterations - replicated part of the committed configs[3] profile) / N + replicated part + two assumed all-reduces",
                                        "per_iteration_us_1gpu": per_it, "replicated_us": rep_us, "assumed_allreduce_us_each": ar,
                                        "speedup_vs_1gpu": {str(n): per_it / (shard_us / n + rep_us + (2 * ar if n > 1 else 0.0)) for n in (1, 2, 4, 8)}}
                large["ten_times_the_landmarks"] = big
            except Exception as e:
                large["ten_times_the_landmarks"] = {"error": repr(e)}

    if rank == 0:
        its = np.array([r.num_iterations for r in reps])
        bytes_per_launch = float(sum(synth.algorithmic_bytes(w) for w in windows))
        flops_per_launch = float(sum(algorithmic_flops(w, int(r.num_iterations)) for w, r in zip(windows, reps)))
        flops_executed = float(sum(algorithmic_flops(w, int(r.num_iterations), int(r.num_successful)) for w, r in zip(windows, reps)))
        k_ms = float(np.mean(kernel_ms))
        ach_tflops = flops_per_launch / (k_ms * 1e-3) / 1e12
        ach_gbs = bytes_per_launch / (k_ms * 1e-3) / 1e9
        # HBM traffic per launch from the PMC passes of profiles/collect.sh -- only if it was measured on THIS kernel build
        traffic, traffic_note = None, "no profiles/pmc_traffic.json"
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("kernel_source_tag") == kernel_source_tag():
                    traffic, traffic_note = tj.get("hbm_bytes_per_launch"), "rocprofv3 PMC, " + str(tj.get("source"))
                else:
                    traffic_note = "profiles/pmc_traffic.json was measured on another kernel build (tag %s, this build %s): not reported" % (tj.get("kernel_source_tag"), kernel_source_tag())
            except Exception as e:
                traffic_note = "unreadable profiles/pmc_traffic.json: %s" % e
        ksolve_512 = os.environ.get("UVS_KSOLVE_NT", "512") != "256"      # which instantiation of the persistent kernel the library launches (csrc/uvs_solve512.hip / the 256-thread one)
        roofline = {"bound": "mfma", "kernel": "uvsdev512::k_solve (512 threads: two waves per SIMD, evaluator / gatherer wave roles)" if ksolve_512 else "uvsdev::k_solve (256 threads, UVS_KSOLVE_NT=256)", "achieved": ach_tflops, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": ach_tflops / FP64_PEAK_TFLOPS, "traffic": traffic, "traffic_note": traffic_note,
                    "kernel_ms_per_launch": k_ms, "algorithmic_flops_per_launch": flops_per_launch,
                    "frac_as_executed": flops_executed / (k_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, "flops_as_executed_per_launch": flops_executed,
                    "algorithmic_bytes_per_launch": bytes_per_launch, "hbm_achieved_GBps": ach_gbs, "hbm_frac": ach_gbs / HBM_PEAK_GBS,
                    "note": "'mfma' bound = FP64 FLOP roof 78.6 TF/s (FP64 MFMA rate = FP64 vector rate on MI355X); flop/byte model of SURVEY.md 8d (frac: every LM iteration credited with a full linearization; "
                            "frac_as_executed: factor evaluation credited only where it runs -- at the start and after accepted steps, a rejected step is followed by a re-damping of the stored linearization); "
                            "dense reduced solve + IMU blocks run on v_mfma_f64_16x16x4_f64, the sparse 6x6 Schur gather on VALU; the kernel is latency bound "
                            "(one wavefront per SIMD, see DESIGN.md section 5); traffic = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 of the same command (8 B/lane calibration: profiles/fetch_calibration.txt)"}
        # single-window latency mode (BASELINE configs[1]): one window resident, one launch per solve
        solver.upload(windows[:1])
        for _ in range(3):
            solver.solve_resident()
        lat = [solver.solve_resident() for _ in range(10)]
        solver.solve(windows[0])                                   # warms the pinned staging buffers of the single-window path
        pc = []
        for _ in range(5):
            solver.solve(windows[0]); pc.append(solver.last_solve_ms * 1e-3)      # uvs_solve_window() alone: host packing + H2D + kernel + D2H
        pcie = float(np.median(pc))
        sw_ms = float(np.median(lat)); sw_fl = float(algorithmic_flops(windows[0], int(reps[0].num_iterations)))
        # the same window through the multi-workgroup fused loop (landmark chunks on many CUs, reduced solve in one workgroup, control on the device)
        solver.large_comm_init(None)
        fl_loop, fl_call = [], []
        for i in range(0 if args.no_fused_single else 7):
            _, frep, lms = solver.large_solve_fused(windows[0])
            if i >= 2: fl_loop.append(lms); fl_call.append(solver.last_solve_ms)
        single = {"workload": "BASELINE configs[1]: one resident W10-P150-L40-V3 window (with the n = 75 prior), one launch per solve",
                  "ms": sw_ms, "solves_per_s": 1e3 / sw_ms, "pcie_inclusive_ms": pcie * 1e3, "pcie_inclusive_solves_per_s": 1.0 / pcie,
                  "multi_workgroup": None if args.no_fused_single else {"what": "uvs_large_solve_fused on the same window: the LM loop as a stream of launches over many compute units (an otherwise idle GPU), what the host mirror's Estimator::optimization() uses",
                                      "lm_loop_ms": float(np.median(fl_loop)), "call_ms_with_packing_and_pcie": float(np.median(fl_call)), "solves_per_s_call": 1e3 / float(np.median(fl_call)),
                                      "lm_iterations": int(frep.num_iterations), "final_cost": float(frep.final_cost)},
                  "roofline": {"bound": "mfma", "kernel": "uvsdev::k_solve (1 workgroup on 1 of 256 CUs)", "achieved": sw_fl / (sw_ms * 1e-3) / 1e12, "peak": FP64_PEAK_TFLOPS,
                               "unit": "TFLOP/s", "frac": sw_fl / (sw_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, "frac_of_one_cu": sw_fl / (sw_ms * 1e-3) / 1e12 / (FP64_PEAK_TFLOPS / 256),
                               "note": "a single window occupies one compute unit: the whole-chip fraction is bounded by 1/256, the per-CU fraction is the meaningful one"}}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from oracle_binding import Oracle     # CPU oracle: baseline leg only
            import subprocess, tempfile
            # the same oracle sources built -O3 -march=native on THIS host (the checker build is -O2 and portable)
            native_dir = tempfile.mkdtemp(); native = os.path.join(native_dir, "liboracle_native.so"); build = "-O3 -march=native (built on this host)"
            try:
                subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "native", "OUT=" + native_dir], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                orc = Oracle(native)
            except Exception:
                orc = Oracle(); build = "-O2 checker build (the native build failed on this host)"
            t1 = time.perf_counter(); orc.solve(windows[0]); one = time.perf_counter() - t1
            ns = args.cpu_sample or int(max(8, min(args.batch, 10.0 / max(one, 1e-3))))
            t1 = time.perf_counter()
            for w in windows[:ns]:
                orc.solve(w)
            tc = time.perf_counter() - t1
            # one window per PHYSICAL core, one PINNED PROCESS per core: the throughput comparator of BASELINE.md section 3.  (Rounds 2 - 4 ran a thread pool over ctypes
            # on every hardware thread the affinity mask names: 8 x the single-thread rate on 256 threads -- SMT siblings share one FP64 pipe, a container's mask lists
            # cores other tenants are using, and the pool's workers migrated.  A forked child touches no HIP state: it solves prepared windows with the CPU oracle and
            # leaves through os._exit.)
            cores = os.cpu_count() or 1
            try:
                usable = sorted(os.sched_getaffinity(0))              # what this process may actually run on (containers: < cpu_count)
            except AttributeError:
                usable = list(range(cores))
            phys, seen = [], set()
            for cpu_i in usable:                                       # first hardware thread of every physical core in the mask
                try:
                    sib = open(f"/sys/devices/system/cpu/cpu{cpu_i}/topology/thread_siblings_list").read().strip()
                except OSError:
                    sib = str(cpu_i)
                if sib not in seen:
                    seen.add(sib); phys.append(cpu_i)
            jobs = [orc.prepare(w) for w in windows]                   # struct conversion: outside the timed region, inherited by the children
            budget_s = 8.0
            def child(k, cpu_i, wfd):
                try:
                    try: os.sched_setaffinity(0, {cpu_i})
                    except OSError: pass
                    orc.solve_prepared(jobs[k % len(jobs)])            # warm-up (page faults of this process)
                    n_done, j = 0, k % len(jobs)
                    t_start = time.perf_counter(); deadline = t_start + budget_s
                    while time.perf_counter() < deadline:
                        orc.solve_prepared(jobs[j]); n_done += 1
                        j = (j + len(phys)) % len(jobs)
                    os.write(wfd, ("%d %.6f\n" % (n_done, time.perf_counter() - t_start)).encode())
                finally:
                    os._exit(0)
            pipes, pids = [], []
            sys.stdout.flush(); sys.stderr.flush()
            for k, cpu_i in enumerate(phys):
                rfd, wfd = os.pipe()
                pid = os.fork()
                if pid == 0:
                    os.close(rfd); child(k, cpu_i, wfd)
                os.close(wfd); pipes.append(rfd); pids.append(pid)
            rates, nmt = [], 0
            for rfd, pid in zip(pipes, pids):
                buf = b""
                while True:
                    chunk = os.read(rfd, 256)
                    if not chunk: break
                    buf += chunk
                os.close(rfd); os.waitpid(pid, 0)
                try:
                    n_done, dt_c = buf.split(); nmt += int(n_done); rates.append(int(n_done) / float(dt_c))
                except ValueError:
                    pass
            mp_rate = float(sum(rates)); eff = mp_rate / max(len(phys) * (ns / tc), 1e-9)
            cpu = {"value": ns / tc, "unit": "solves/s", "cores": 1, "kind": "port",
                   "sample": f"first {ns} windows of the same batch, single-thread C++ oracle (oracle/uvs_oracle.cpp, {build}), {tc:.1f} s",
                   "host_cpus": cores}
            mp = {"value": mp_rate, "unit": "solves/s", "cores": len(phys), "parallel_efficiency_vs_single_thread": eff,
                  "sample": f"{nmt} solves over the same batch in {budget_s:.0f} s, one pinned process per physical core ({len(phys)} of os.cpu_count() = {cores} hardware threads, affinity mask = {len(usable)}), sum of the per-process rates"}
            # ONE stable key with a validity flag beside it (round-5 advisor: a reader of cpu.multithread must not silently fall back to the single-thread figure on a loaded host)
            mp["valid_comparator"] = bool(eff >= 0.5)
            if eff < 0.5: mp["note"] = "under half of linear scaling: this host's cores are shared or throttled, so the figure understates an idle host's multi-core rate (baseline only either way)"
            cpu["multithread"] = mp
        # closed-loop replay through the product host library (ATE half of BASELINE.json's metric; configs[4] as far as this image allows):
        # processIMU / processImage / optimization (HIP) / marginalization (HIP) / slideWindow over the WHOLE MH_05_difficult ground-truth
        # trajectory the reference holds (benchmark_publisher/config/MH_05_difficult/data.csv -> tests/golden/mh05_groundtruth.npz), with
        # IMU samples and point / line / vanishing-point messages synthesised along it (uv-slam_amd/sequence.py: make_groundtruth_sequence);
        # scored against the recorded rows with the reference's association rule (uv-slam_amd/trajectory.py, tools/ate.py)
        replay = None
        if world == 1 and not args.no_replay:
            try:
                import ctypes as C, tempfile
                seqm, traj = uvs.sequence, uvs.trajectory
                gt = traj.load_groundtruth_fixture(os.path.join(ROOT, "tests", "golden", "mh05_groundtruth.npz"))
                tmpd = tempfile.mkdtemp()

                def run_replay(lib_path, seq, tag):
                    pin, pout, pres = os.path.join(tmpd, tag + "_seq.bin"), os.path.join(tmpd, tag + "_out.bin"), os.path.join(tmpd, tag + "_vins_result.txt")
                    seqm.save(seq, pin)
                    lib = C.CDLL(lib_path)
                    lib.uvs_host_replay_sequence.argtypes = [C.c_char_p, C.c_char_p]; lib.uvs_host_replay_sequence.restype = C.c_int
                    os.environ["UVS_VINS_RESULT_PATH"] = pres
                    t1 = time.perf_counter(); rc = lib.uvs_host_replay_sequence(pin.encode(), pout.encode()); dt = time.perf_counter() - t1
                    del os.environ["UVS_VINS_RESULT_PATH"]
                    if rc != 0:
                        raise RuntimeError("sequence replay failed: %d" % rc)
                    r = seqm.load_result(pout)
                    tm = (C.c_double * 4)()
                    if hasattr(lib, "uvs_host_replay_timing"):
                        lib.uvs_host_replay_timing.argtypes = [C.POINTER(C.c_double)]; lib.uvs_host_replay_timing.restype = None
                        lib.uvs_host_replay_timing(tm)
                    return r, traj.ate(pres, gt), dt / len(r["frame"]) * 1e3, list(tm)

                seq = seqm.make_groundtruth_sequence(gt)
                r, a, ms_frame, tm = run_replay(os.path.join(ROOT, "uv-slam_amd", "libuvs_host.so"), seq, "hip")
                kinds = np.bincount(r["flag"], minlength=2)
                replay = {"workload": "MH_05_difficult GT trajectory, synthetic measurements",
                          "detail": f"{seq.n_frames} keyframe candidates at 10 Hz over {seq.stamps[-1] - seq.stamps[0]:.1f} s of the recorded ground truth (rows {int(seq.gt_rows[0])}..{int(seq.gt_rows[-1])} of 22212), "
                                    f"200 Hz IMU with the recorded biases, <= 150 points / <= 40 lines per image, 0.5 px noise; {len(r['frame'])} chained windows "
                                    f"({int(kinds[0])} MARGIN_OLD, {int(kinds[1])} MARGIN_SECOND_NEW), failureDetection never fired",
                          "ate_vs_recorded_groundtruth_m": a["rmse_m"], "ate_mean_m": a["mean_m"], "ate_max_m": a["max_m"], "n_matched": a["n_matched"],
                          "unaligned_drift_max_m": float(np.linalg.norm(r["P"] - seq.truth_pose[r["frame"], :3], axis=1).max()),
                          "points_per_window_median": float(np.median(r["n_points"])), "lines_per_window_median": float(np.median(r["n_lines"])),
                          "ms_per_frame_whole_replay": ms_frame, "frames_solved": len(r["frame"]),
                          "note": "ms_per_frame_whole_replay = the whole uvs_host_replay_sequence() call / frames (file parsing, handle creation, IMU integration, "
                                  "triangulation, solve, marginalization); the three entries below are wall-clock inside Estimator::optimization() per call",
                          "optimization_ms_per_call": tm[0], "solve_ms_per_call": tm[1], "marginalize_ms_per_call": tm[2], "optimization_calls": int(tm[3])}
                if cpu is not None:       # the same state machine with the CPU oracle behind the C ABI (baseline leg only), on a bounded PREFIX of the same trajectory
                    n_pre = 200
                    pre = seqm.make_groundtruth_sequence(gt, t_end=3.0 + 0.1 * n_pre + 0.05)
                    ro, ao, ms_o, tmo = run_replay(os.path.join(ROOT, "oracle", "libuvs_host_oracle.so"), pre, "oracle")
                    m = len(ro["frame"])
                    Pt = pre.truth_pose[ro["frame"], :3]
                    cpu["replay_sample"] = f"the first {pre.n_frames} frames ({m} chained windows) of the same sequence"
                    cpu["replay_ate_vs_recorded_groundtruth_m"] = ao["rmse_m"]; cpu["replay_ms_per_frame"] = ms_o; cpu["replay_optimization_ms_per_call"] = tmo[0]
                    replay["prefix_vs_oracle_backend"] = {"windows": m, "ate_hip_m": seqm.ate(r["P"][:m], Pt), "ate_oracle_m": seqm.ate(ro["P"], Pt),
                                                          "max_dP_m": float(np.linalg.norm(r["P"][:m] - ro["P"], axis=1).max()),
                                                          "same_marginalization_flags": bool(np.array_equal(r["flag"][:m], ro["flag"]))}
            except Exception as e:      # (a secondary leg must never cost the bench line; the error is reported instead of the numbers)
                replay = {"workload": "MH_05_difficult GT trajectory, synthetic measurements", "error": repr(e)}
        # the marginalization that follows every solve of a replay, for the WHOLE batch in one call (uvs_marginalize_batch, ABI v7): the post-solve windows of the timed batch
        marg_batch = None
        if world == 1 and not args.no_prior and not args.no_replay:
            try:
                import ctypes as C
                post = [w.with_state(st) for w, st in zip(windows, states)]
                keeps = [w.to_c() for w in post]
                n = len(post)
                arr = (C.POINTER(abi.WindowC) * n)(*[C.pointer(k[0]) for k in keeps])
                pri = (abi.Prior * n)(); stc = (C.c_int * n)()
                L = uvs.api.lib(); L.uvs_marginalize_batch.restype = C.c_int
                marg_batch = {"windows": n}
                for flag, name in ((0, "margin_old"), (1, "margin_second_new")):
                    fl = (C.c_int * n)(*([flag] * n))
                    ts = []
                    for _ in range(4):
                        t1 = time.perf_counter(); rc = L.uvs_marginalize_batch(solver._h, n, arr, fl, pri, stc); ts.append(time.perf_counter() - t1)
                    t1 = time.perf_counter()
                    for k in range(16): L.uvs_marginalize(solver._h, C.byref(keeps[k][0]), flag, C.byref(pri[k]))
                    one = (time.perf_counter() - t1) / 16
                    # (the last 16 one-window calls overwrote pri[0..15]: compare them with the batch's priors of the same windows, in the information form the next solve reads)
                    one_pri = [(np.array(pri[k].J0()), np.array(pri[k].r0())) for k in range(16)]
                    rc2 = L.uvs_marginalize_batch(solver._h, n, arr, fl, pri, stc)
                    worst = 0.0
                    for k in range(16):
                        J1, r1 = one_pri[k]; Jb, rb = np.array(pri[k].J0()), np.array(pri[k].r0())
                        H1, Hb = J1.T @ J1, Jb.T @ Jb
                        worst = max(worst, float(np.abs(Hb - H1).max() / max(np.abs(H1).max(), 1e-300)), float(np.abs(Jb.T @ rb - J1.T @ r1).max() / max(1.0, np.abs(J1.T @ r1).max())))
                    marg_batch[name] = {"batch_call_ms": float(np.median(ts[1:])) * 1e3, "us_per_window": float(np.median(ts[1:])) / n * 1e6, "status": int(rc) | int(rc2),
                                        "windows_ok": int(sum(1 for x in stc if x == 0)), "one_window_call_ms": one * 1e3, "max_rel_diff_vs_one_window_call_16_samples": worst}
                marg_batch["note"] = ("C-ABI call times. One launch linearizes the sub-windows of all MARGIN_OLD windows (k_marg_linearize_batch), one launch eliminates every window's dropped frame block "
                                      "and factors its n x n Schur complement (k_marg_finish: parallel cyclic Jacobi, csrc/uvs_marg_kernel.h); the one-window call finishes on a host core")
            except Exception as e:      # (a secondary leg must never cost the bench line)
                marg_batch = {"error": repr(e)}
        out = {
            "metric": "sliding-window solves/sec (10 KF, 150 pts, 40 lines, 3 VP)", "value": value, "unit": "solves/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"W10-P150-L40-V3 x {args.batch} independent windows per GPU (BASELINE configs[2]); "
                                   "single window (configs[1]) in single_window, the 20k-point window (configs[3]) in large_window",
                       "windows_per_gpu": args.batch, "frames": 11, "points": 150, "lines": 40, "vp_tagged_lines": 30,
                       "prior": (not args.no_prior), "max_lm_iterations": 10, "parallelism": f"replicas x{world}"},
            "lm_iterations_mean": float(its.mean()), "lm_successful_steps_mean": float(np.mean([r.num_successful for r in reps])), "final_cost_mean": float(np.mean([r.final_cost for r in reps])),
            "batch_pack_upload_ms": pack_upload_ms, "value_end_to_end": (end_to_end or {}).get("solves_per_s"), "end_to_end": end_to_end,
            "single_window": single, "single_window_ms": sw_ms, "single_window_solves_per_s": 1e3 / sw_ms, "single_window_pcie_inclusive_ms": pcie * 1e3,
            "replay": replay, "marginalization_batch": marg_batch, "large_window": large, "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    solver.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
