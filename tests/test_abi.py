"""The C ABI library loads and exports every function include/uvs_solver.h declares; ctypes mirrors match the C structs.

No compute call is made here (there is no GPU in the build container and the library has no CPU path).
"""
import ctypes as C
import numpy as np
import os
import re
import subprocess
import tempfile

import pytest

from helpers import uvs, abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "uvs_solver.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(uvs_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = uvs.api.lib()
    names = declared_functions()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/uvs_solver.h but not exported by libuvs_solver.so"
    assert lib.uvs_abi_version() == 7


def test_default_options_match_python_mirror():
    lib = uvs.api.lib()
    o = abi.Options(); lib.uvs_default_options(C.byref(o))
    p = abi.default_options()
    for name, _ in abi.Options._fields_:
        a, b = getattr(o, name), getattr(p, name)
        if hasattr(a, "__len__"):
            assert list(a) == list(b), name
        else:
            assert a == b, name
    assert lib.uvs_reduced_dim(C.byref(o)) == 165


def test_struct_layouts_match_the_header():
    code = r'''
#include <stdio.h>
#include <stddef.h>
#include "uvs_solver.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu\n", sizeof(uvs_options), sizeof(uvs_imu_block), sizeof(uvs_prior), sizeof(uvs_window), sizeof(uvs_state), sizeof(uvs_report), sizeof(uvs_eval));
  printf("%zu %zu %zu %zu\n", offsetof(uvs_window, n_points), offsetof(uvs_window, imu), offsetof(uvs_prior, linearized_jacobians), offsetof(uvs_report, accepted));
  return 0; }
'''
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "s.c"); exe = os.path.join(d, "s")
        open(src, "w").write(code)
        subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), src, "-o", exe])    # the header is plain C
        out = subprocess.check_output([exe]).decode().split()
    sizes = [C.sizeof(t) for t in (abi.Options, abi.ImuBlock, abi.Prior, abi.WindowC, abi.StateC, abi.Report, abi.EvalC)]
    assert [int(v) for v in out[:7]] == sizes
    offs = [abi.WindowC.n_points.offset, abi.WindowC.imu.offset, abi.Prior.linearized_jacobians.offset, abi.Report.accepted.offset]
    assert [int(v) for v in out[7:]] == offs


def test_create_fails_loudly_without_a_gpu():
    """The product has no CPU path: on a machine without a HIP device uvs_create must refuse (on the GPU box it succeeds)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        uvs.api.Solver()


def test_status_strings():
    lib = uvs.api.lib()
    assert b"no HIP device" in lib.uvs_status_string(abi.UVS_ERR_NO_DEVICE)
    assert lib.uvs_status_string(0) == b"ok"


def test_pack_layout_host_only(monkeypatch):
    """uvs_debug_pack_layout: the packing of uvs_batch_upload() without a device.  Every option mix must keep its fullest chunk inside
    the LDS staging area (the td + extrinsic capacity estimate was once an entry short per observation: only the GPU saw it)."""
    lib = uvs.api.lib()
    lib.uvs_debug_pack_layout.argtypes = [C.POINTER(abi.Options), C.POINTER(abi.WindowC), C.POINTER(C.c_int32)]
    synth = uvs.synth
    rng = np.random.default_rng(5)
    for i in range(24):
        npt, nln = int(rng.integers(20, 300)), int(rng.integers(0, 80))
        w = synth.make_window(24000 + i, n_points=npt, n_lines=nln, n_tagged=int(rng.integers(0, nln + 1)), pt_track=int(rng.integers(3, 10)), ln_track=int(rng.integers(5, 10)))
        mode = i % 4
        o = abi.default_options()
        if mode == 1: o.estimate_td = 1; w = synth.add_time_offset(w)
        if mode == 2: o.estimate_extrinsic = 1; o.estimate_td = 1; w = synth.add_time_offset(w)
        if mode == 3: w = synth.add_relocalization(w, relo_frame=int(rng.integers(0, 10)), fraction=1.0, seed=i)
        wc, keep = w.to_c(); info = (C.c_int32 * 12)()
        assert lib.uvs_debug_pack_layout(C.byref(o), C.byref(wc), info) == abi.UVS_OK
        blob, wsd, nch, npo, nrelo, prec, xs, mx, cap = list(info)[:9]
        assert 0 < mx <= cap == 17952 and nch >= 1 + (nln > 0) and blob % 256 == 0 and wsd > 0
        assert npo == len(w.pt_lm) + len(w.relo_lm) and nrelo == len(w.relo_lm)
        assert prec == (46 if mode == 2 else 34 if mode == 1 else 30) and xs == 1 + (mode in (1, 2)) + (mode == 2)
    # relocalization blocks beside a free extrinsic: relo_Pose as a second-level block -- packed for the persistent kernel and (since round 6) for the chunk
    # grids of the landmark-sharded forms as well, with the same observations and records
    w = synth.add_relocalization(synth.make_window(3), seed=3)
    o = abi.default_options(); o.estimate_extrinsic = 1
    wc, keep = w.to_c(); info = (C.c_int32 * 12)()
    assert lib.uvs_debug_pack_layout(C.byref(o), C.byref(wc), info) == abi.UVS_OK and list(info)[4] == len(w.relo_lm)
    monkeypatch.setenv("UVS_DEBUG_CHUNK_GRID", "64")
    first = list(info)
    assert lib.uvs_debug_pack_layout(C.byref(o), C.byref(wc), info) == abi.UVS_OK and list(info)[3:7] == first[3:7] and list(info)[2] >= first[2]
    monkeypatch.delenv("UVS_DEBUG_CHUNK_GRID")


def test_large_window_chunking_fills_the_grid(monkeypatch):
    """uvs_large_begin packs with the grid of its persistent kernels (compute units - 1): the chunk count becomes a multiple of the grid (every
    workgroup carries the same number of chunks), a small window or a shard gets one chunk per workgroup (>= 64 observations each), and
    every chunk still fits the LDS staging area.  Host-only through UVS_DEBUG_CHUNK_GRID."""
    lib = uvs.api.lib(); synth = uvs.synth
    lib.uvs_debug_pack_layout.argtypes = [C.POINTER(abi.Options), C.POINTER(abi.WindowC), C.POINTER(C.c_int32)]
    o = abi.default_options()
    def chunks(w, grid):
        if grid: monkeypatch.setenv("UVS_DEBUG_CHUNK_GRID", str(grid))
        else: monkeypatch.delenv("UVS_DEBUG_CHUNK_GRID", raising=False)
        wc, keep = w.to_c(); info = (C.c_int32 * 12)()
        assert lib.uvs_debug_pack_layout(C.byref(o), C.byref(wc), info) == abi.UVS_OK
        assert info[7] <= info[8]      # fullest chunk <= staging area
        return info[2]
    big = synth.make_window(70, n_points=4000, n_lines=1000, n_tagged=700)
    n_min = chunks(big, 0)
    assert n_min > 64
    for grid in (64, 255):
        n = chunks(big, grid)
        assert n % grid == 0 and n >= n_min and n < n_min + grid
    shard = synth.shard_landmarks(big, 0, 8)[0]
    n_obs = len(shard.pt_lm) + len(shard.ln_lm)
    assert chunks(shard, 0) < 64 and chunks(shard, 255) == min(255, n_obs // 64)      # one chunk per workgroup, but not under 64 observations per chunk
    small = synth.make_window(3)
    assert chunks(small, 0) == 5 and chunks(small, 255) == (750 + 280) // 64
    monkeypatch.delenv("UVS_DEBUG_CHUNK_GRID", raising=False)


def test_malformed_prior_is_rejected_not_dereferenced():
    """validate_window() checked block_idx / block_frame of the prior but not x0_off, block_size or block_kind, and pack_window
    then read x0[x0_off[b] + k] unchecked (a prior with x0_off = 1e8 crashed the process).  Host-only: no device is touched."""
    lib = uvs.api.lib()
    lib.uvs_debug_pack_layout.argtypes = [C.POINTER(abi.Options), C.POINTER(abi.WindowC), C.POINTER(C.c_int32)]
    o = abi.default_options()

    def status_with(mutate):
        w = uvs.synth.make_window(11, n_points=30, n_lines=8, n_tagged=4)
        p = abi.Prior(); p.n = 15; p.n_blocks = 2
        p.block_kind[0] = abi.BLOCK_POSE; p.block_frame[0] = 0; p.block_size[0] = 7; p.block_idx[0] = 0; p.x0_off[0] = 0
        p.block_kind[1] = abi.BLOCK_SPEEDBIAS; p.block_frame[1] = 0; p.block_size[1] = 9; p.block_idx[1] = 6; p.x0_off[1] = 7
        p.x0[6] = 1.0
        for k in range(15): p.linearized_jacobians[k * 15 + k] = 1.0
        mutate(p)
        w.prior = p
        wc, keep = w.to_c(); info = (C.c_int32 * 12)()
        return lib.uvs_debug_pack_layout(C.byref(o), C.byref(wc), info)

    assert status_with(lambda p: None) == abi.UVS_OK
    assert status_with(lambda p: p.x0_off.__setitem__(0, 100000000)) == abi.UVS_ERR_INVALID_ARG
    assert status_with(lambda p: p.x0_off.__setitem__(1, -3)) == abi.UVS_ERR_INVALID_ARG
    assert status_with(lambda p: p.x0_off.__setitem__(1, 144 - 8)) == abi.UVS_ERR_INVALID_ARG      # 9 doubles do not fit behind offset 136
    assert status_with(lambda p: p.block_kind.__setitem__(0, 7)) == abi.UVS_ERR_INVALID_ARG
    assert status_with(lambda p: p.block_size.__setitem__(0, 9)) == abi.UVS_ERR_INVALID_ARG         # a pose block is 7 wide
    assert status_with(lambda p: p.block_size.__setitem__(1, 1000)) == abi.UVS_ERR_INVALID_ARG


def test_window_file_loader_rejects_corrupt_headers(tmp_path):
    """Window.load trusted the header counts (prior_n > 96 overran the fixed-size prior arrays)."""
    import struct
    w = uvs.synth.make_window(12, n_points=20, n_lines=6, n_tagged=3)
    path = str(tmp_path / "w.bin"); w.save(path)
    assert len(abi.Window.load(path).pt_lm) == len(w.pt_lm)
    raw = bytearray(open(path, "rb").read())
    for slot, value in ((5, 97), (6, 17), (4, 11), (0, -1)):       # prior_n, prior blocks, imu blocks, points
        bad = bytearray(raw); struct.pack_into("<i", bad, 8 + 4 * slot, value)
        open(path, "wb").write(bad)
        with pytest.raises(ValueError):
            abi.Window.load(path)


def test_prior_that_keeps_a_block_twice_is_refused(oracle):
    """Two kept blocks on the same parameter block (or overlapping prior columns) would map two prior columns to one index of the reduced system; the device's (H0 entry,
    S offset) table has one slot per pair of S indices (setup_window), so the packing refuses such a prior instead of letting the table overrun (round-4 advisor finding)."""
    lib = uvs.api.lib()
    lib.uvs_debug_pack_layout.argtypes = [C.POINTER(abi.Options), C.POINTER(abi.WindowC), C.POINTER(C.c_int32)]
    w = uvs.synth.make_window(3, with_prior=True, marginalize_fn=lambda win, flag: oracle.marginalize(win, flag))
    o = abi.default_options(); info = (C.c_int32 * 12)()
    wc, keep = w.to_c()
    assert lib.uvs_debug_pack_layout(C.byref(o), C.byref(wc), info) == abi.UVS_OK
    poses = [b for b in range(w.prior.n_blocks) if w.prior.block_kind[b] == abi.BLOCK_POSE]
    assert len(poses) >= 2
    dup = w.copy(); dup.prior.block_frame[poses[1]] = dup.prior.block_frame[poses[0]]      # the same pose block twice
    wc, keep = dup.to_c()
    assert lib.uvs_debug_pack_layout(C.byref(o), C.byref(wc), info) == abi.UVS_ERR_INVALID_ARG
    ov = w.copy(); ov.prior.block_idx[poses[1]] = ov.prior.block_idx[poses[0]] + 3        # overlapping columns of J0
    wc, keep = ov.to_c()
    assert lib.uvs_debug_pack_layout(C.byref(o), C.byref(wc), info) == abi.UVS_ERR_INVALID_ARG
