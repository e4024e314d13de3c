"""Turns the rocprofv3 CSVs under gpurun_out/ into the small summaries committed under profiles/.

Collected on MI355X with (each PMC group in its own pass, per MI355X_MICROARCH.md):
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline
  rocprofv3 --pmc FETCH_SIZE --kernel-trace ...   rocprofv3 --pmc WRITE_SIZE --kernel-trace ...   rocprofv3 --pmc SQ_... --kernel-trace ...
"""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
out = {}
def _find(d, suffix):
    fs = glob.glob(f"{ROOT}/gpurun_out/{d}/**/*{suffix}", recursive=True)
    return fs[0] if fs else None
rows = list(csv.DictReader(open(_find("prof_kt", "_kernel_trace.csv"))))
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
big = [dur(r) for r in rows if "k_solve" in r["Kernel_Name"] and int(r["Grid_Size_X"]) > 1024]      # batch launches (256 windows x 256 threads)
one = [dur(r) for r in rows if "k_solve" in r["Kernel_Name"] and int(r["Grid_Size_X"]) <= 1024]    # single-window launches
out.update(k_solve_batch_launches=len(big), k_solve_batch_avg_ms=sum(big) / len(big), k_solve_batch_min_ms=min(big), k_solve_batch_max_ms=max(big),
           k_solve_single_window_avg_ms=sum(one) / max(len(one), 1))
for name in ("prof_fetch", "prof_write", "prof_sq", "prof_sq2"):
    f = _find(name, "_counter_collection.csv")
    if not f: continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_solve" in r["Kernel_Name"] and int(r["Grid_Size"]) > 1024:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            out.update(vgpr=r["VGPR_Count"], sgpr=r["SGPR_Count"], lds_block_size=r["LDS_Block_Size"], scratch_size=r["Scratch_Size"], grid=r["Grid_Size"])
    for k, v in agg.items():
        out[k + "_per_launch_mean"] = sum(v) / len(v)
# rocprofv3 FETCH_SIZE / WRITE_SIZE are KiB.  MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports half of the bytes of a coalesced
# 16 B/lane read, other widths and WRITE_SIZE uncalibrated -> calibrated here for this kernel's 8 B/lane accesses with tools/calib_fetch.hip
# (profiles/fetch_calibration.txt): FETCH_SIZE x2.000, WRITE_SIZE x1.000.
if "FETCH_SIZE_per_launch_mean" in out:
    out["hbm_bytes_per_launch"] = (2 * out["FETCH_SIZE_per_launch_mean"] + out["WRITE_SIZE_per_launch_mean"]) * 1024
    out["hbm_bytes_per_launch_uncorrected"] = (out["FETCH_SIZE_per_launch_mean"] + out["WRITE_SIZE_per_launch_mean"]) * 1024
# ---- the landmark-sharded kernels of the configs[3] fused loop (prof_large_*): per kernel, mean per launch over the launches that did work (a pass after termination returns at once)
large = {}
for name in ("prof_large_fetch", "prof_large_write", "prof_large_sq"):
    f = _find(name, "_counter_collection.csv")
    if not f: continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        if "k_large_" not in kn: continue
        short = kn.split("(")[0].split("::")[-1]
        agg[(short, r["Counter_Name"])].append(float(r["Counter_Value"]))
        large.setdefault(short, {}).update(vgpr=r["VGPR_Count"], lds_block_size=r["LDS_Block_Size"], scratch_size=r["Scratch_Size"], grid=r["Grid_Size"], workgroup=r["Workgroup_Size"])
    for (short, cn), v in agg.items():
        nz = [x for x in v if x > 0] or v
        large[short][cn + "_per_launch_mean"] = sum(nz) / len(nz); large[short][cn + "_launches"] = len(v)
if large:
    # FETCH_SIZE / WRITE_SIZE in KiB; same calibration as above (x2 for the 8 B / lane reads).  Per LM iteration = one launch of each kernel.
    tot = 0.0
    for k, d in large.items():
        if "FETCH_SIZE_per_launch_mean" in d and "WRITE_SIZE_per_launch_mean" in d:
            d["hbm_bytes_per_launch"] = (2 * d["FETCH_SIZE_per_launch_mean"] + d["WRITE_SIZE_per_launch_mean"]) * 1024; tot += d["hbm_bytes_per_launch"]
            d["hbm_bytes_per_launch_uncorrected"] = (d["FETCH_SIZE_per_launch_mean"] + d["WRITE_SIZE_per_launch_mean"]) * 1024      # (k_large_reduce re-reads the 10.2 MB of partials k_large_chunks has just written: its uncorrected figure is the plausible one -- the x2 of the streaming-read calibration is an upper bound there)
    out["large_window_kernels"] = large
    out["large_window_hbm_bytes_per_iteration"] = tot
json.dump(out, open(f"{ROOT}/profiles/{tag}_pmc_summary.json", "w"), indent=1)
if out.get("large_window_hbm_bytes_per_iteration"):
    sys.path.insert(0, ROOT)
    import bench as _b
    json.dump({"hbm_bytes_per_iteration": out["large_window_hbm_bytes_per_iteration"], "source": f"profiles/{tag}_pmc_summary.json", "kernel_source_tag": _b.large_source_tag()},
              open(f"{ROOT}/profiles/pmc_traffic_large.json", "w"))
if "hbm_bytes_per_launch" in out:
    # bench.py reports this figure only while the kernel sources still hash to the tag recorded here (run summarize.py on the build that was profiled)
    sys.path.insert(0, ROOT)
    import bench
    json.dump({"hbm_bytes_per_launch": out["hbm_bytes_per_launch"], "source": f"profiles/{tag}_pmc_summary.json", "kernel_source_tag": bench.kernel_source_tag()},
              open(f"{ROOT}/profiles/pmc_traffic.json", "w"))
print(json.dumps(out, indent=1))
