import csv, sys, collections
for f in sys.argv[1:]:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        if "gemm" not in kn and "attn" not in kn and "ln_" not in kn: continue
        agg[kn[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for kn, cs in agg.items():
        print(kn)
        for c, v in cs.items():
            v = v[len(v)//2:]  # skip warm-up
            print(f"   {c:34s} {sum(v)/len(v):16.0f}  (n={len(v)})")
