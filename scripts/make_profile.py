"""Post-process scripts/gpu_profile.sh output into the files kept under profiles/ (round tag TAG, workload WL):
  <TAG>_bench_<WL>_kernel_stats.csv     rocprofv3 --kernel-trace --stats of the bench command
  <TAG>_bench_<WL>_under_rocprof.json   the bench line of that run
  <TAG>_pmc_traffic_<WL>.json           per kernel: dispatches, FETCH_SIZE / WRITE_SIZE bytes per launch (separate --pmc passes)
  mf_traffic_<WL>.json                  memory-side bytes of the multifrontal kernels per instance-factorisation, with the
                                        nnz(L) of the plan it was measured on (bench.py reads it for this workload only)
FETCH_SIZE / WRITE_SIZE are KiB.  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half the bytes of wide (16 B / lane)
streaming reads; these kernels read 8 B per lane, a width the guide calls uncalibrated -- the raw figure is kept and the
doubled one is given beside it as an upper bound.   python make_profile.py DIR TAG WL"""
import csv, glob, json, os, shutil, sys, collections
root, tag, wl = sys.argv[1], sys.argv[2], sys.argv[3]

def per_kernel(sub, counter):
    tot = collections.defaultdict(float); n = collections.defaultdict(set)
    for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            tot[k] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    return {k: (tot[k], len(n[k])) for k in tot}

fe, wr = per_kernel("fetch", "FETCH_SIZE"), per_kernel("write", "WRITE_SIZE")
out = {}
for k in sorted(set(fe) | set(wr)):
    if "sqphip" not in k:
        continue
    f, nf = fe.get(k, (0.0, 0)); w, nw = wr.get(k, (0.0, 0))
    out[k] = {"dispatches": nf or nw, "fetch_bytes_total_raw": 1024.0 * f, "write_bytes_total": 1024.0 * w,
              "fetch_bytes_per_launch_raw": 1024.0 * f / max(1, nf), "write_bytes_per_launch": 1024.0 * w / max(1, nw)}
json.dump(out, open(os.path.join(root, f"{tag}_pmc_traffic_{wl}.json"), "w"), indent=1)
bj = json.loads(open(os.path.join(root, "fetch.json")).read().strip().splitlines()[-1])
nfac = bj["config"]["kkt_factorisations"]
mf = {k: v for k, v in out.items() if "k_mf_" in k}
if mf and nfac:
    fr = sum(v["fetch_bytes_total_raw"] for v in mf.values()); w = sum(v["write_bytes_total"] for v in mf.values())
    rf = bj["roofline"]
    json.dump({"kernels": " + ".join(sorted({k.split("<")[0].replace("sqphip::", "") for k in mf})),
               "workload": wl, "nnz_l": rf.get("nnz_l"), "nnz_k_lower": rf.get("nnz_k_lower"),
               "instance_factorisations": nfac,
               "fetch_bytes_per_instance_factorisation_raw": fr / nfac,
               "write_bytes_per_instance_factorisation": w / nfac,
               "hbm_bytes_per_instance_factorisation": (fr + w) / nfac,
               "hbm_bytes_per_instance_factorisation_fetch_doubled": (2 * fr + w) / nfac,
               "algorithmic_bytes_per_instance_factorisation": rf.get("bytes_per_instance_factorisation"),
               "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over bench.py --steps 2 --warmup 0 --quick "
                      "of this workload; KiB -> bytes, summed over every dispatch of the multifrontal kernels and divided by the "
                      "instance-factorisations of the run.  Memory-side (fabric) requests: Infinity-Cache hits are counted.  "
                      "8-byte-per-lane accesses: FETCH_SIZE uncalibrated on gfx950 (MI355X_MICROARCH.md, HBM section); the doubled "
                      "figure is the guide's correction for 16-byte-per-lane reads, an upper bound here."},
              open(os.path.join(root, f"mf_traffic_{wl}.json"), "w"), indent=1)
for f in glob.glob(os.path.join(root, "stats", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(root, f"{tag}_bench_{wl}_kernel_stats.csv"))
shutil.copy(os.path.join(root, "bench_under_rocprof.json"), os.path.join(root, f"{tag}_bench_{wl}_under_rocprof.json"))
p = os.path.join(root, f"mf_traffic_{wl}.json")
print(open(p).read() if os.path.exists(p) else "no mf kernels")
