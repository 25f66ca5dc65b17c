#!/bin/bash
# Matrix-pipe utilisation of the extend kernel from rocprofv3 PMC counters (one pass: 8 SQ + 1 GRBM).
# Usage: bash tools/pmc_extend.sh <tag> [bench_prefill args...]
set -u
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d "$out/pmc" -- python3 "$GRAFT_REPO_ROOT/tools/bench_prefill.py" --reps 30 "$@" > "$out/bench.txt" 2> "$out/pmc.err"
rc=$?
echo "pmc pass rc=$rc"
if [ $rc -ne 0 ]; then tail -5 "$out/pmc.err"; exit $rc; fi
cat "$out/bench.txt" | cut -c1-200
python3 - "$out" <<'PY'
import csv, glob, json, os, sys
root = sys.argv[1]
acc = {}
dur = {}
rows = []
for f in glob.glob(os.path.join(root, "pmc", "**", "*counter_collection.csv"), recursive=True):
    rows += [r for r in csv.DictReader(open(f)) if "prefill_mfma" in r.get("Kernel_Name", "")]
# (the first launches after the set-up run slower: only the second half of the dispatches is averaged)
ids = sorted({int(r["Dispatch_Id"]) for r in rows})
late = set(ids[len(ids) // 2:])
for row in rows:
    if int(row["Dispatch_Id"]) not in late:
        continue
    acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    if "Start_Timestamp" in row and "End_Timestamp" in row:
        dur[row["Dispatch_Id"]] = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
avg = {k: sum(v) / len(v) for k, v in acc.items()}
res = {"dispatches": len(next(iter(acc.values()))) if acc else 0, "per_dispatch": avg}
if dur:
    res["kernel_ns"] = sum(dur.values()) / len(dur)
    if "GRBM_GUI_ACTIVE" in avg:
        res["clock_ghz"] = avg["GRBM_GUI_ACTIVE"] / 8.0 / res["kernel_ns"]  # (reads high on dispatches under 0.3 ms)
if "GRBM_GUI_ACTIVE" in avg and "SQ_VALU_MFMA_BUSY_CYCLES" in avg:
    cycles = avg["GRBM_GUI_ACTIVE"] / 8.0  # the counter is summed over the 8 XCDs
    res["kernel_cycles"] = cycles
    res["mfma_busy_fraction"] = avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (cycles * 1024.0)  # 256 CUs x 4 SIMDs
    res["note"] = "SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the SIMDs; kernel cycles = GRBM_GUI_ACTIVE / 8"
json.dump(res, open(os.path.join(root, "mfma_util.json"), "w"), indent=1)
print(json.dumps(res))
PY
