import csv, collections, sys
for d in sys.argv[1:]:
    rows=list(csv.DictReader(open(d+"/p_counter_collection.csv")))
    acc=collections.defaultdict(list)
    for r in rows:
        if "gemm_fp8" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    kt=list(csv.DictReader(open(d+"/p_kernel_trace.csv")))
    ds=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"])) for r in kt if "gemm_fp8" in r["Kernel_Name"]]
    print(d, "kernel dur us mean", sum(ds[5:])/len(ds[5:])/1e3)
    for k,v in sorted(acc.items()):
        v=v[5:]
        print(f"  {k:32s} mean={sum(v)/len(v):.5g}")
