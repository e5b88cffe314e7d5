#!/usr/bin/env python3
"""Regenerates profiles/pmc_traffic.json from the PMC passes of tools/pmc_zstd.sh / tools/pmc_dec.sh (gpurun_out/pmc, gpurun_out/pmc_dec)
and stamps every record with the sha256 of the kernel source it was measured on: bench.py quotes `roofline.traffic` only while
that source is unchanged (a kernel change without a new PMC pass must not leave the driver line quoting stale counters).
usage: python tools/pmc_traffic.py [--enc gpurun_out/pmc] [--dec gpurun_out/pmc_dec] [--tag r03]"""
import argparse
import collections
import csv
import glob
import hashlib
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "tiered-storage-for-apache-kafka_amd", "csrc")


def source_sha(names):
    h = hashlib.sha256()
    for n in names:
        h.update(open(os.path.join(CSRC, n), "rb").read())
    return h.hexdigest()[:16]


ENC_SOURCES = ["zstd_enc.hip", "zstd_common.h", "gcm_dev.h", "crc_dev.h"]
CRC_SOURCES = ["crc32c.hip", "crc_dev.h"]
GCM_SOURCES = ["gcm.hip", "gcm_dev.h"]
DEC_SOURCES = ["zstd_dec.hip", "zstd_dec_dev.h", "zstd_common.h"]


def counters(d):
    agg = collections.defaultdict(list)
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            if "zstd_service" in r.get("Kernel_Name", "") and r.get("Grid_Size") == "524288":
                continue                                              # the calibration launch of tsx_init (32 workgroups per CU that just wait)
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}


def counters_prefixed(d, prefix):
    agg = collections.defaultdict(list)
    for sub in sorted(glob.glob(os.path.join(d, prefix + "*"))):
        if not os.path.isdir(sub):
            continue
        for f in sorted(glob.glob(os.path.join(sub, "**", "*counter_collection.csv"), recursive=True)):
            for r in csv.DictReader(open(f)):
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--enc", default=os.path.join(ROOT, "gpurun_out", "pmc"))
    ap.add_argument("--dec", default=os.path.join(ROOT, "gpurun_out", "pmc_dec"))
    ap.add_argument("--small", default=os.path.join(ROOT, "gpurun_out", "pmc_small"))
    ap.add_argument("--tag", default="r05")
    ap.add_argument("--chunks", type=int, default=2048)
    ap.add_argument("--sequences-per-chunk", type=float, default=175358.0)
    args = ap.parse_args()
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    rec = json.load(open(path))
    for key, d, srcs, kernel in (("full/K/%d" % args.chunks, args.enc, ENC_SOURCES, "zstd_service_kernel (whole chain: CRC32C head, compressor, AES-256-GCM tail), one launch = the chunks of one batch"),
                                 ("detransform/K/%d" % args.chunks, args.dec, DEC_SOURCES, "zstd_decompress_kernel"),
                                 ("crc/K/256", (args.small, "crc_p"), CRC_SOURCES, "crc32c_partial_kernel"),
                                 ("gcm_crc/K/256", (args.small, "gcm_p"), GCM_SOURCES, "gcm_ctr_ghash_kernel")):
        if isinstance(d, tuple):
            c, cnt = counters_prefixed(*d)
        else:
            c, cnt = counters(d)
        if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
            print("no FETCH_SIZE / WRITE_SIZE under", d, "- record", key, "left as it is")
            continue
        out = os.path.join(ROOT, "profiles", "%s_%s_pmc.txt" % (args.tag, {"full": "zstd_service", "detransform": "zstd_decompress", "crc": "crc32c", "gcm_crc": "gcm"}[key.split("/")[0]]))
        with open(out, "w") as f:
            for k in sorted(c):
                f.write("%-30s launches=%d mean=%.5g\n" % (k, cnt[k], c[k]))
        fetch, write = c["FETCH_SIZE"] * 1000.0, c["WRITE_SIZE"] * 1000.0       # reported in KB
        streaming = key.startswith("crc") or key.startswith("gcm")
        if streaming:
            # wide coalesced streaming reads (16 B per lane): on gfx950 FETCH_SIZE reports exactly half of the bytes (MI355X_MICROARCH.md, HBM:
            # 128-B requests tallied at 64 B) - doubled.  Calibration on these kernels' own known byte count: the doubled figure is 0.98 of
            # the 1 GiB each of them reads; WRITE_SIZE of the GCM kernel is 0.98 of the 1 GiB it writes as it stands.
            fetch *= 2.0
        r = {"hbm_bytes_per_launch": int(fetch + write), "fetch_bytes": int(fetch), "write_bytes": int(write), "kernel": kernel, "chunks": int(key.split("/")[-1]),
             "kernel_source_sha": source_sha(srcs), "kernel_sources": srcs, "source": os.path.relpath(out, ROOT), "tag": args.tag,
             "note": ("FETCH_SIZE / WRITE_SIZE are reported in KB (x1000 here); wide coalesced 16-byte-per-lane streaming reads: FETCH_SIZE doubled (the gfx950 correction of "
                      "MI355X_MICROARCH.md, HBM section; checked against the kernel's known byte count)") if streaming else
                     ("FETCH_SIZE / WRITE_SIZE are reported in KB (x1000 here); they derive from TCC_EA0_RDREQ / WRREQ and include Infinity-Cache hits "
                      "(MI355X_MICROARCH.md, HBM section); random 4-byte probes / short match copies, so the gfx950 x2 correction for wide streaming reads does not apply")}
        if "TCC_EA0_RDREQ_sum" in c:
            r["tcc_ea_rdreq"] = int(c["TCC_EA0_RDREQ_sum"]); r["tcc_ea_wrreq"] = int(c["TCC_EA0_WRREQ_sum"])
            if key.startswith("full"):
                nseq = args.sequences_per_chunk * args.chunks
                r["requests_per_sequence"] = {"read": round(c["TCC_EA0_RDREQ_sum"] / nseq, 2), "write": round(c["TCC_EA0_WRREQ_sum"] / nseq, 2)}
        for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "GRBM_GUI_ACTIVE"):
            if k in c:
                r.setdefault("sq", {})[k] = int(c[k])
        rec[key] = r
        print(key, json.dumps(r)[:300])
    json.dump(rec, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
