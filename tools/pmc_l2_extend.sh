#!/bin/bash
# L2 hit rate and request mix of the extend kernel (one --pmc pass: 4 TCC counters).
# Usage: bash tools/pmc_l2_extend.sh <tag> [bench_prefill args...]
set -u
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d "$out/pmc" -- python3 "$GRAFT_REPO_ROOT/tools/bench_prefill.py" --reps 10 "$@" > "$out/bench.txt" 2> "$out/pmc.err"
rc=$?
echo "pmc pass rc=$rc"
if [ $rc -ne 0 ]; then tail -5 "$out/pmc.err"; exit $rc; fi
python3 - "$out" <<'PY'
import csv, glob, json, os, sys
root = sys.argv[1]
acc = {}
for f in glob.glob(os.path.join(root, "pmc", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "prefill_mfma" in r.get("Kernel_Name", ""):
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
avg = {k: sum(v) / len(v) for k, v in acc.items()}
if "TCC_HIT_sum" in avg and "TCC_MISS_sum" in avg:
    avg["l2_hit_rate"] = avg["TCC_HIT_sum"] / max(1.0, avg["TCC_HIT_sum"] + avg["TCC_MISS_sum"])
json.dump(avg, open(os.path.join(root, "l2.json"), "w"), indent=1)
print(json.dumps(avg))
PY
