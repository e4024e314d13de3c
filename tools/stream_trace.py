"""Timeline of uvs_batch_stream from a rocprofv3 run with --kernel-trace --memory-copy-trace --hip-runtime-trace (csv) in the directory given: GPU kernels and copies beside the
host's HIP calls, last 40 ms.  `python tools/stream_trace.py gpurun_out/prof_stream2`"""
import csv, glob, os, sys
d = sys.argv[1]
f = lambda pat: glob.glob(os.path.join(d, "**", pat), recursive=True)[0]
ev = []
for r in csv.DictReader(open(f("*kernel_trace.csv"))):
    nm = r["Kernel_Name"].split("(")[0].split("::")[-1]
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "GPU " + nm[:30] + " q" + r["Queue_Id"]))
for r in csv.DictReader(open(f("*memory_copy_trace.csv"))):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "GPU copy " + r["Direction"].replace("MEMORY_COPY_", "")))
for r in csv.DictReader(open(f("*hip_api_trace.csv"))):
    n = r["Function"]
    if n.startswith("hip") and not n.startswith(("hipGet", "hipSetDevice", "hipPeek", "hipHostGetDevice", "__hip")):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "      host " + n + " t" + r["Thread_Id"][-3:]))
ev.sort()
win = float(sys.argv[2]) if len(sys.argv) > 2 else 25.0
big = [e for e in ev if "k_solve" in e[2]]
tend = big[-1][1]
t0 = tend - int(win * 1e6)
for s, e, n in ev:
    if s >= t0 and s <= tend + 200000:
        print("%9.3f %9.3f %7.3f  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, n))
