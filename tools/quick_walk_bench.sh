#!/bin/bash
# Traceback-heavy bench lines only (after a change to the walker): gpurun -- bash tools/quick_walk_bench.sh
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/walk
mkdir -p "$OUT"
run() {
  local name=$1 secs=$2
  shift 2
  timeout -k 10 "$secs" python3 -u "$ROOT/bench.py" "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"
  local rc=$?
  echo "$name rc=$rc" | tee -a "$OUT/status.txt"
  return $rc
}
run nw1m 200 --no-batch --no-cpu || exit 1
run nw1m_unrelated 200 --no-batch --unrelated --no-cpu || exit 1
run lcs1m 200 --kind lcs --no-batch --no-cpu || exit 1
run nw100k_unrelated 120 --size 100000 --steps 5 --no-batch --unrelated --no-cpu || exit 1
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/walk/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], d["value"], d["ms_per_step"], d.get("breakdown_ms_per_step"))
PY
