#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) for the decode kernel.

FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch.  On gfx950 FETCH_SIZE
counts 128-byte read requests at 64 bytes, i.e. exactly half of a wide coalesced
stream, so the read side is doubled; WRITE_SIZE is exact for 16-byte stores.
Writes <dir>/traffic.json with the per-launch average of the decode kernel."""
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
res = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(os.path.join(root, ctr, "**", "*counter_collection.csv"), recursive=True)
    vals = []
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if "decode_mfma_kernel" in row.get("Kernel_Name", "") and row.get("Counter_Name") == ctr:
                    vals.append(float(row["Counter_Value"]))
    res[ctr] = {"dispatches": len(vals), "avg_kib": sum(vals) / len(vals) if vals else None}
fetch, write = res["FETCH_SIZE"]["avg_kib"], res["WRITE_SIZE"]["avg_kib"]
if fetch is not None and write is not None:
    res["hbm_bytes_per_launch"] = int(2 * fetch * 1024 + write * 1024)
    res["note"] = "2 x FETCH_SIZE (gfx950 counts 128-B read requests as 64 B) + WRITE_SIZE, KiB -> bytes"
json.dump(res, open(os.path.join(root, "traffic.json"), "w"), indent=1)
print(json.dumps(res))
