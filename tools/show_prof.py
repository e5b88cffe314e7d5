import json, sys
for f in sys.argv[1:]:
    r = json.load(open(f))
    print(f, {k: (round(v, 1) if isinstance(v, float) else v) for k, v in r.items() if k != 'buckets'})
    for k, v in r["buckets"].items():
        print("  %-45s %14.0f %s" % (k, v["mean"], "" if v["frac_of_total"] is None else "%.3f" % v["frac_of_total"]))
