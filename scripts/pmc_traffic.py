#!/usr/bin/env python
"""Summarise the rocprofv3 --pmc passes of scripts/gpu_check.sh: per-kernel HBM traffic per launch.

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB.  On gfx950 FETCH_SIZE counts 64 B per 128-B
request for wide coalesced reads (MI355X_MICROARCH.md §HBM): the read side is DOUBLED here before it
is compared with a byte count; WRITE_SIZE is taken as is (uncalibrated, see the guide).  Working sets
below 256 MiB live in the Infinity Cache, so these are fabric-side bytes, not DRAM bytes.
Writes profiles-ready JSON to <dir>/pmc_traffic.json (copy to profiles/pmc_traffic.json to have
bench.py report it as roofline.traffic)."""
import collections
import csv
import glob
import json
import os
import sys

out_dir = sys.argv[1]


def per_kernel(sub, counter):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(out_dir, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"]
            a = agg[k]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in agg.items() if v[1]}


fetch = per_kernel("pmc_fetch", "FETCH_SIZE")
write = per_kernel("pmc_write", "WRITE_SIZE")
sq = {}
for c in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
          "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_LDS"):
    for k, (v, n) in per_kernel("pmc_sq", c).items():
        sq.setdefault(k, {})[c] = round(v)
rows = []
for k in sorted(set(fetch) | set(write)):
    if "dmpnn" not in k:
        continue
    f = fetch.get(k, (0.0, 0))
    w = write.get(k, (0.0, 0))
    rows.append(dict(kernel=k[:110], launches=f[1] or w[1], fetch_KiB_raw=round(f[0], 1), write_KiB=round(w[0], 1),
                     hbm_bytes_per_launch=round((2.0 * f[0] + w[0]) * 1024), sq=sq.get(k, {})))
res = dict(note="FETCH_SIZE doubled (gfx950 counts 64 B per 128-B request); per-launch averages", kernels=rows)
# the dominant kernel: fused update with the message epilogue
bench = {}
try:
    bench = json.loads(open(os.path.join(out_dir, "bench.json")).read().strip().splitlines()[-1])
except Exception:
    pass
upd = [r for r in rows if "k_gemmILi3ELi5ELi4ELb0ELi1" in r["kernel"] or ("k_gemm<3, 5, 4, false, 1>" in r["kernel"])]
meg = [r for r in rows if "k_mpnn_tile16" in r["kernel"]]
if meg:
    res["mega_kernel_bytes_per_launch"] = meg[0]["hbm_bytes_per_launch"]
if upd:
    res["update_kernel_bytes_per_launch"] = upd[0]["hbm_bytes_per_launch"]
if bench:
    res["directed_edges"] = bench.get("config", {}).get("directed_edges_per_gpu")
    res["hidden"] = 300
json.dump(res, open(os.path.join(out_dir, "pmc_traffic.json"), "w"), indent=1)
for r in rows:
    print(f"{r['kernel'][:70]:70s} n={r['launches']:4d} fetch(raw KiB)={r['fetch_KiB_raw']:10.1f} write(KiB)={r['write_KiB']:10.1f} "
          f"bytes/launch={r['hbm_bytes_per_launch']:12d}")
    if r["sq"]:
        print("    ", r["sq"])
