"""Average duration of every launch position of a sweep (kernel trace of rocprofv3): which level launches are slow.
usage: trace_by_position.py <dir with *_kernel_trace.csv> [first sweep] [last sweep]"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 60
hi = int(sys.argv[3]) if len(sys.argv) > 3 else 160
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_qp_finish" in r["Kernel_Name"]]
acc = collections.defaultdict(lambda: [0.0, 0, "", ""])
nsw = 0
ref_len = None
for a, b in zip(idx[lo:hi], idx[lo + 1:hi + 1]):
    seq = rows[a:b]
    if ref_len is None: ref_len = len(seq)
    if len(seq) != ref_len: continue
    nsw += 1
    for k, r in enumerate(seq):
        e = acc[k]; e[0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; e[1] += 1
        e[2] = r["Kernel_Name"].replace("sqphip::", "").replace("void ", "").split("(")[0][:30]; e[3] = f'{r["Grid_Size_X"]}x{r["Grid_Size_Y"]}/{r["Workgroup_Size_X"]}'
print(f"{nsw} sweeps of {ref_len} launches")
tot = 0
for k in range(ref_len or 0):
    e = acc[k]; tot += e[0] / max(1, e[1])
    print(f"{k:3d} {e[0] / max(1, e[1]):7.1f} us  {e[2]:32s} {e[3]}")
print("sum", round(tot, 1), "us")
