#!/bin/sh
# One command on a machine that has Ceres <= 2.1 and Eigen (the build image has neither): writes the window files of the three canonical goldens
# (tests/golden/*.npz: canonical_prior, canonical_vp_heavy, small_relo is left out -- the harness has no relocalization blocks) plus the small ones,
# builds the harness against the real library and runs it on each: tests/golden/ceres/<name>.bin + <name>.trc.  `git add tests/golden/ceres` then
# turns tests/test_ceres_traces.py green or red -- for the CPU oracle (-m "not gpu") and for the HIP solver (-m gpu) alike.
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT="$ROOT/tests/golden/ceres"
mkdir -p "$OUT"
python3 - "$ROOT" "$OUT" <<'PY'
import importlib.util, os, sys
root, out = sys.argv[1], sys.argv[2]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
spec = importlib.util.spec_from_file_location("make_golden", os.path.join(root, "tests", "golden", "make_golden.py"))
mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
import numpy as np
for name in ("canonical_prior", "canonical_vp_heavy", "small_prior", "small_noprior", "points_only"):
    d = dict(np.load(os.path.join(root, "tests", "golden", name + ".npz")))
    mg.dict_to_window(d).save(os.path.join(out, name + ".bin"))
    print("wrote", name + ".bin")
PY
make -C "$ROOT/oracle"
BUILD=${BUILD:-/tmp/uvs_ceres_harness}
cmake -S "$ROOT/tools/ceres_harness" -B "$BUILD" -DCMAKE_BUILD_TYPE=Release
cmake --build "$BUILD" -j
for w in "$OUT"/*.bin; do "$BUILD/ceres_harness" "$w" "${w%.bin}.trc"; done
ls -l "$OUT"
