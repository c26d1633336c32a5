#!/bin/bash
# Round-3 profiles on the GPU box: rocprofv3 kernel-trace stats and PMC passes (each pass a run of its own, no trace
# domains beside --pmc), for the headline pair (1 Mb), config 3 (100 kb) and the config-5 batch.  Run from the repo root:
#   gpurun -- bash tools/profile_r03.sh
# Outputs under gpurun_out/r3p/; tools/summarize_r03.py turns them into profiles/r03_*.
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r3p
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {  # name, seconds, rocprof args..., -- program
  local name=$1 secs=$2
  shift 2
  echo "== $name" >&2
  timeout -s KILL "$secs" rocprofv3 "$@" > "$OUT/$name.out" 2>&1
  local rc=$?
  echo "$name rc=$rc" >> "$OUT/status.txt"
  return $rc
}
B="python3 $ROOT/bench.py --no-cpu --no-small"
PART=${1:-all}   # a | b | all | s (kernel-trace stats only): gpurun limits one call to 20 minutes
SQ="SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_WAVES"
if [ "$PART" != b ]; then
run stats_1m   240 --kernel-trace --stats --output-format csv -d "$OUT/stats_1m"   -- $B --steps 3 --warmup 1 --no-batch || exit 1
run stats_100k 120 --kernel-trace --stats --output-format csv -d "$OUT/stats_100k" -- $B --steps 5 --warmup 1 --no-batch --size 100000 || exit 1
run stats_100k_unrel 120 --kernel-trace --stats --output-format csv -d "$OUT/stats_100k_unrel" -- $B --steps 5 --warmup 1 --no-batch --size 100000 --unrelated || exit 1
run stats_lcs1m 240 --kernel-trace --stats --output-format csv -d "$OUT/stats_lcs1m" -- $B --steps 3 --warmup 1 --no-batch --kind lcs || exit 1
run stats_sw1m 240 --kernel-trace --stats --output-format csv -d "$OUT/stats_sw1m" -- $B --steps 3 --warmup 1 --no-batch --kind sw || exit 1
run stats_2m 300 --kernel-trace --stats --output-format csv -d "$OUT/stats_2m" -- $B --steps 1 --warmup 0 --no-batch --size 2000000 || exit 1
run stats_lcs2m 300 --kernel-trace --stats --output-format csv -d "$OUT/stats_lcs2m" -- $B --steps 1 --warmup 0 --no-batch --size 2000000 --kind lcs || exit 1
run stats_1m_unrel 240 --kernel-trace --stats --output-format csv -d "$OUT/stats_1m_unrel" -- $B --steps 3 --warmup 1 --no-batch --unrelated || exit 1
run stats_batch 240 --kernel-trace --stats --output-format csv -d "$OUT/stats_batch" -- $B --steps 2 --warmup 1 --workload batch || exit 1
WL="1m 100k lcs1m"
fi
if [ "$PART" = s ]; then WL=""; fi
if [ "$PART" = b ]; then WL="sw1m batch"; fi
if [ "$PART" = all ]; then WL="1m 100k lcs1m sw1m batch"; fi
for w in $WL; do
  case $w in
    lcs1m) ARGS="--steps 1 --warmup 0 --no-batch --kind lcs" ;;
    1m) ARGS="--steps 1 --warmup 0 --no-batch" ;;
    sw1m) ARGS="--steps 1 --warmup 0 --no-batch --kind sw" ;;
    100k) ARGS="--steps 1 --warmup 0 --no-batch --size 100000" ;;
    batch) ARGS="--steps 1 --warmup 0 --workload batch" ;;
  esac
  run pmc_write_$w 240 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write_$w" -- $B $ARGS || exit 1
  run pmc_fetch_$w 240 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch_$w" -- $B $ARGS || exit 1
  run pmc_sq_$w    240 --pmc $SQ GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq_$w" -- $B $ARGS || exit 1
done
if [ "$PART" != a ] && [ "$PART" != s ]; then
# the second wave per SIMD: fill_group with one and with two workgroups per CU (the library reads
# SA_GROUP / SA_WPS when the context is created; exported here, in front of rocprofv3, not inside its command)
for wps in 1 2; do
  export SA_GROUP=1 SA_WPS=$wps
  w=1m_group_wps$wps
  ARGS="--steps 1 --warmup 0 --no-batch"
  run stats_$w 240 --kernel-trace --stats --output-format csv -d "$OUT/stats_$w" -- $B $ARGS || exit 1
  run pmc_write_$w 240 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write_$w" -- $B $ARGS || exit 1
  run pmc_fetch_$w 240 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch_$w" -- $B $ARGS || exit 1
  run pmc_sq_$w    240 --pmc $SQ GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq_$w" -- $B $ARGS || exit 1
done
unset SA_GROUP SA_WPS
fi
# keep what is needed, drop the bulky per-dispatch traces
find "$OUT" -name "*kernel_trace.csv" -size +2M -delete
find "$OUT" -name "*.db" -delete
du -sh "$OUT" >&2
cat "$OUT/status.txt"
