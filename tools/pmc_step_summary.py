"""Sum FETCH_SIZE / WRITE_SIZE (KiB units) over all kernels of the profiled bench run; per step = total / (steps+warmup)."""
import csv, glob, sys, collections
steps = int(sys.argv[1])
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/pmc_step_{c}/**/*counter_collection.csv", recursive=True)
    if not f: print(c, "missing"); continue
    tot = 0.0; per = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == c:
            v = float(r["Counter_Value"]); tot += v; per[r["Kernel_Name"][:60]] += v
    print(f"{c}: total {tot/1e6:.3f} GB(KiB-units*1e-6) over {steps} steps = {tot*1024/steps/1e9:.3f} GB/step (raw counter; FETCH_SIZE under-reports wide streaming reads by 2x on gfx950)")
    for k, v in per.most_common(6): print(f"    {v*1024/steps/1e6:9.1f} MB/step  {k}")
