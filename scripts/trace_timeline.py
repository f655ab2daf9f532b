"""Timeline of one host-pointer call from a `rocprofv3 --kernel-trace --memory-copy-trace --output-format csv` run of harness/trcbench:
kernels and copies of the LAST encode (or decode) call, microseconds since its first event.  usage: trace_timeline.py DIR [enc|dec]"""
import csv
import glob
import sys

d = sys.argv[1]
which = sys.argv[2] if len(sys.argv) > 2 else "enc"
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", r["Kernel_Name"].split("(")[0][:44], r.get("Queue_Id", "")))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C", r.get("Direction", r.get("Name", "copy"))[:44], ""))
ev.sort()
pat = "enc" if which == "enc" else "dec"
ks = [e for e in ev if e[2] == "K" and ("_%s" % pat) in e[3]]
if not ks:
    sys.exit("no %s kernels in the trace" % pat)
# the last call: walk back from the last coder kernel while gaps between coder kernels stay below 20 ms
last = ks[-1]
first = last
for e in reversed(ks):
    if first[0] - e[1] > 20_000_000:
        break
    first = e
lo, hi = first[0] - 3_000_000, last[1] + 3_000_000
sel = [e for e in ev if lo <= e[0] <= hi]
t0 = sel[0][0]
print("# %s call: events within 3 ms of its coder kernels; us since the first; K = kernel (queue id), C = copy" % which)
for s, e, k, name, q in sel:
    print("%9.1f %9.1f  %7.1f us  %s %-44s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, k, name, q))
