"""Average PMC counters per kernel name + grid from rocprofv3 --pmc csv output. usage: python tools/pmc_summary.py gpurun_out/pmc_*"""
import csv, glob, sys, collections, re
for d in sys.argv[1:]:
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs: print(d, "no counter csv"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        n = r["Kernel_Name"]
        m = re.search(r"GemmCfgI(DF16b|f)(?:DF16b|f)Lb(\d)ELb(\d)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d)ELi(\d)ELi(\d+)", n)
        key = (f"gemm p{m.group(2)}q{m.group(3)} {m.group(4)}x{m.group(5)}x{m.group(6)} epi{m.group(9)}" if m else n[:40], r["Grid_Size"])
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", d)
    for key, cs in agg.items():
        print("  ", key, {k: round(sum(v) / len(v), 1) for k, v in cs.items()}, "n=", len(next(iter(cs.values()))))
