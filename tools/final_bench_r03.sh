#!/bin/bash
# Round-3 bench lines on the GPU box (each the one JSON line bench.py prints), kept under profiles/r03_bench_*.json.
#   gpurun -- bash tools/final_bench_r03.sh
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r3f
mkdir -p "$OUT"
run() {  # name, seconds, bench args...
  local name=$1 secs=$2
  shift 2
  echo "== $name" >&2
  timeout -k 10 "$secs" python3 -u "$ROOT/bench.py" "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"
  local rc=$?
  echo "$name rc=$rc" | tee -a "$OUT/status.txt"
  return $rc
}
run default 400 || exit 1   # configs 4 (headline pair), 5 (nested batch), 1 and 2 (nested small_pairs)
run nw100k 120 --size 100000 --steps 5 --no-batch || exit 1
run nw100k_unrelated 120 --size 100000 --steps 5 --no-batch --unrelated || exit 1
run nw1m_unrelated 200 --no-batch --unrelated --no-cpu || exit 1
run lcs1m 200 --kind lcs --no-batch || exit 1
run lcs100k 120 --kind lcs --size 100000 --steps 5 --no-batch || exit 1
run sw1m 200 --kind sw --no-batch || exit 1
run nw2m_checkpointed 300 --size 2000000 --steps 1 --warmup 1 --no-batch --no-cpu --no-small || exit 1   # (warm: the first call allocates 250 GB, 5 s)
run lcs2m_checkpointed 300 --kind lcs --size 2000000 --steps 2 --warmup 1 --no-batch --no-cpu --no-small || exit 1
run batch_align 200 --workload batch-align || exit 1
run batch_global 200 --workload batch-global || exit 1
cat "$OUT/status.txt"
