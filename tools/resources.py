import re,sys,subprocess,glob
for f in sorted(glob.glob("vit4hep_amd/_build/*.resources.txt")):
    txt=open(f).read()
    blocks=re.split(r"remark: Function Name: ",txt)[1:]
    print("==",f.split("/")[-1], "kernels:",len(blocks))
    for b in blocks:
        n=b.split()[0]
        g=lambda k: int(re.search(k+r": (\d+)",b).group(1))
        v,sp,sc,occ,lds=g(" VGPRs"),g("VGPRs Spill"),g(r"ScratchSize \[bytes/lane\]"),g(r"Occupancy \[waves/SIMD\]"),g(r"LDS Size \[bytes/block\]")
        d=subprocess.run(["c++filt",n],capture_output=True,text=True).stdout.strip()
        d=re.sub(r"\(anonymous namespace\)::|v4h::","",d)
        if sp or sc or v>120 or "-a" in sys.argv: print(f"  vgpr={v:3d} spill={sp} scratch={sc} occ={occ} lds={lds:6d}  {d[:130]}")
