"""The C ABI library loads and exports every function include/uvs_solver.h declares; ctypes mirrors match the C structs.

No compute call is made here (there is no GPU in the build container and the library has no CPU path).
"""
import ctypes as C
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
    assert lib.uvs_abi_version() == 3


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
