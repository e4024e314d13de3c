"""Copy / kernel timeline of bench.py's end-to-end leg from a rocprofv3 --kernel-trace --memory-copy-trace run (profiles/collect.sh writes it to gpurun_out/prof_stream):
every k_solve launch and every copy longer than 0.3 ms of the LAST stream of batches, in time order.  What it shows on this platform: the host-to-device copy of batch k + 1
starts when k_solve of batch k ends, although it was enqueued while that kernel ran -- copies and the persistent kernel do not overlap."""
import csv, glob, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
kt = glob.glob(os.path.join(ROOT, "gpurun_out", "prof_stream", "**", "*kernel_trace.csv"), recursive=True)[0]
mc = glob.glob(os.path.join(ROOT, "gpurun_out", "prof_stream", "**", "*memory_copy_trace.csv"), recursive=True)[0]
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "k_solve (queue %s, grid %s)" % (r["Queue_Id"], r["Grid_Size_X"])) for r in csv.DictReader(open(kt)) if "k_solve" in r["Kernel_Name"] and int(r["Grid_Size_X"]) > 1024]
for r in csv.DictReader(open(mc)):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e - s > 300000: ev.append((s, e, r["Direction"]))
ev.sort()
# the last run of launches that alternate queues = the timed stream
stream = [i for i, (_, _, n) in enumerate(ev) if n.startswith("k_solve")]
qs = [ev[i][2] for i in stream]
last = len(stream) - 1
while last > 0 and qs[last] == qs[last - 1]: last -= 1      # (the resident timing loop behind the stream runs on one queue)
first = last
while first > 0 and qs[first] != qs[first - 1]: first -= 1
t0 = ev[stream[first]][0]
print("# start [ms]  end [ms]  duration [ms]  what")
for s, e, n in ev[stream[first]: stream[last] + 1]:
    print("%10.3f %10.3f %8.3f   %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, n))
