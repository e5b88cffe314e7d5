#!/usr/bin/env python3
"""Which kernel of a fetch waited, and next to what?  Reads a rocprofv3 --kernel-trace CSV (the run of tools/mixed_load_notorch.py or any other
process that fetches next to the compressor service) and prints: the service kernel's launches, and every dispatch of another kernel that
STARTED more than --ms after the previous dispatch of its own stream had ended (a fetch queues its kernels back to back: such a gap is a
kernel that was not placed) - with the three dispatches in front of it, its resources, and what else started inside the gap.
  python tools/trace_gaps.py <kernel_trace.csv> [--ms 20]"""
import argparse
import csv
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--ms", type=float, default=20.0)
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.csv)))
    for r in rows:
        r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"]); r["name"] = r["Kernel_Name"].split("(")[0]
    rows.sort(key=lambda r: r["s"])
    if not rows:
        print("empty trace"); return
    t0 = rows[0]["s"]
    svc = [r for r in rows if r["name"] == "zstd_service_kernel"]
    print("service launches: %d" % len(svc))
    for r in svc:
        print("   start %10.1f ms  dur %9.1f ms  grid %s  queue %s" % ((r["s"] - t0) / 1e6, (r["e"] - r["s"]) / 1e6, r["Grid_Size_X"], r["Queue_Id"]))
    others = [r for r in rows if r["name"] != "zstd_service_kernel"]
    by_stream = {}
    for r in others:
        by_stream.setdefault(r["Stream_Id"], []).append(r)
    names = {}
    for r in others:
        names[r["name"]] = names.get(r["name"], 0) + 1
    print("other dispatches: %d  (%s)" % (len(others), ", ".join("%s x %d" % kv for kv in sorted(names.items(), key=lambda kv: -kv[1])[:12])))
    found = 0
    for st, lst in by_stream.items():
        for i in range(1, len(lst)):
            gap = (lst[i]["s"] - lst[i - 1]["e"]) / 1e6
            # a gap between two fetches (the probe sleeps 50 ms between them) is not a stall: only gaps INSIDE a chain count - the chain begins with begin_batch_kernel
            if gap < a.ms or lst[i]["name"] == "begin_batch_kernel":
                continue
            found += 1
            r = lst[i]
            print("\nSTALL stream %s queue %s: %s started %.1f ms after the previous dispatch of its stream ended (at %.1f ms; ran %.3f ms)" % (st, r["Queue_Id"], r["name"], gap, (r["s"] - t0) / 1e6, (r["e"] - r["s"]) / 1e6))
            print("      grid %s x wg %s, LDS %s, scratch %s, VGPR %s" % (r["Grid_Size_X"], r["Workgroup_Size_X"], r["LDS_Block_Size"], r["Scratch_Size"], r["VGPR_Count"]))
            for p in lst[max(0, i - 3):i]:
                print("      before: %-28s start %10.3f ms dur %8.3f ms grid %s wg %s lds %s" % (p["name"], (p["s"] - t0) / 1e6, (p["e"] - p["s"]) / 1e6, p["Grid_Size_X"], p["Workgroup_Size_X"], p["LDS_Block_Size"]))
            inside = [o for o in rows if lst[i - 1]["e"] <= o["s"] <= r["s"] and o is not r]
            print("      dispatches that started inside the gap: %d%s" % (len(inside), "".join("\n         %-28s stream %s queue %s start %10.3f ms dur %9.3f ms" % (o["name"], o["Stream_Id"], o["Queue_Id"], (o["s"] - t0) / 1e6, (o["e"] - o["s"]) / 1e6) for o in inside[:8])))
            ends = [o for o in svc if lst[i - 1]["e"] <= o["e"] <= r["s"] + 2000000]
            print("      service launches that ENDED inside the gap (or within 2 ms of its end): %d" % len(ends))
    print("\nstalls inside a chain (> %.0f ms): %d" % (a.ms, found))
    # longest kernels that are not the service's (a slow kernel is not a stall, but it is what a fetch's latency is made of)
    top = sorted(others, key=lambda r: r["s"] - r["e"])[:6]
    print("longest other kernels: " + ", ".join("%s %.2f ms" % (r["name"], (r["e"] - r["s"]) / 1e6) for r in top))


if __name__ == "__main__":
    main()
