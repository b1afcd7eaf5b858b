"""Idle time between consecutive kernels of one hardware queue (kernel trace of rocprofv3): what a HIP graph of the sweep
could win at most.  usage: trace_gaps.py <dir with *_kernel_trace.csv> [skip fraction at both ends, default 0.2]"""
import csv, glob, sys, collections, statistics
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.2
rows = list(csv.DictReader(open(f)))
byq = collections.defaultdict(list)
for r in rows:
    byq[r.get("Queue_Id", "0")].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
print(f"{len(rows)} dispatches on {len(byq)} queues")
for q, v in sorted(byq.items()):
    v.sort()
    a, b = int(len(v) * skip), int(len(v) * (1 - skip))
    w = v[a:b]
    if len(w) < 100: continue
    gaps = [(w[i + 1][0] - w[i][1]) / 1e3 for i in range(len(w) - 1)]
    dur = [(e - s) / 1e3 for s, e, _ in w]
    pos = [g for g in gaps if g > 0]
    span = (w[-1][1] - w[0][0]) / 1e3
    print(f"queue {q}: {len(w)} kernels, span {span / 1e3:.1f} ms, kernel time {sum(dur) / 1e3:.1f} ms ({100 * sum(dur) / span:.1f} %), "
          f"gaps: median {statistics.median(gaps):.2f} us, mean {statistics.mean(gaps):.2f}, p90 {sorted(gaps)[int(0.9 * len(gaps))]:.2f}, "
          f"max {max(gaps):.1f}; kernels median {statistics.median(dur):.1f} us")
    # gap by the kernel that FOLLOWS it
    byk = collections.defaultdict(list)
    for i, g in enumerate(gaps):
        byk[w[i + 1][2].replace("sqphip::", "").replace("void ", "").split("(")[0][:28]].append(g)
    for k, g in sorted(byk.items(), key=lambda kv: -sum(kv[1]))[:8]:
        print(f"      before {k:30s} n {len(g):6d} median {statistics.median(g):6.2f} us  mean {statistics.mean(g):6.2f}")
