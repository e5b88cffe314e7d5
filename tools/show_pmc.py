import csv, collections, glob, sys
d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
for f in sorted(glob.glob(d + "/p*/p*_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "zstd_service" in r.get("Kernel_Name", "") and r.get("Grid_Size") == "524288":
            continue                                                  # (the calibration launch of tsx_init)
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print("%-30s launches=%d mean=%.5g" % (k, len(v), sum(v) / len(v)))
