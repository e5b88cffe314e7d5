import json, sys
r = json.load(open(sys.argv[1]))
print("zstd_ms", r["zstd_ms_1"])
names = ["0 between steps (post-match, control)", "1 window upkeep + position bytes", "2 hash + table round trip", "3 scoreboard", "4 candidates + verdict",
         "5 inserts+extension+seq store"]
b = list(r["buckets"].values())
tot = 0
for i in range(6):
    print("%-40s %12.0f  per seq %7.0f" % (names[i], b[i]["mean"], b[i]["mean"] / 175358)); tot += b[i]["mean"]
print("sum per seq %.0f" % (tot / 175358))
