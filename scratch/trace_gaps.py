"""Gap analysis of a rocprofv3 --kernel-trace csv: per-queue busy time and the union over queues, over the last frames."""
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0"), r.get("Stream_Id", "")) for r in rows]
ev.sort()
nfr = int(sys.argv[2]) if len(sys.argv) > 2 else 20
# frames delimited by the voxelizer's first kernel
marks = [i for i, e in enumerate(ev) if "vox_hash" in e[2] or "voxel_hash" in e[2] or "vox_insert" in e[2]]
if not marks:
    names = collections.Counter(e[2][:40] for e in ev); print(names.most_common(40)); sys.exit()
lo = marks[-nfr - 1]; hi = marks[-1]
sel = ev[lo:hi]
t0, t1 = sel[0][0], ev[hi][0]
print("frames", nfr, "wall/frame us %.1f" % ((t1 - t0) / nfr / 1e3))
perq = collections.defaultdict(float)
for s, e, n, q, st in sel:
    perq[(q, st)] += e - s
for k, v in sorted(perq.items()):
    print("queue/stream", k, "busy/frame us %.1f" % (v / nfr / 1e3))
# union busy
iv = sorted((s, e) for s, e, *_ in sel)
busy = 0; cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce:
        busy += ce - cs; cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
print("union busy/frame us %.1f  idle/frame us %.1f" % (busy / nfr / 1e3, (t1 - t0 - busy) / nfr / 1e3))
# largest idle gaps (union) with neighbours
gaps = []
iv2 = sorted(sel)
ce = iv2[0][1]; last = iv2[0]
for x in iv2[1:]:
    if x[0] > ce:
        gaps.append((x[0] - ce, last[2][:50], x[2][:50]))
    if x[1] > ce:
        ce = x[1]; last = x
agg = collections.defaultdict(lambda: [0, 0])
for g, a, b in gaps:
    agg[(a, b)][0] += g; agg[(a, b)][1] += 1
for (a, b), (g, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:25]:
    print("%8.1f us/frame  n=%5.1f/frame  %s -> %s" % (g / nfr / 1e3, c / nfr, a, b))
