set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3k; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
SQ="SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_WAVES"
timeout -s KILL 200 rocprofv3 --pmc $SQ --output-format csv -d $OUT/pmc_sq_1m -- python3 $ROOT/bench.py --no-cpu --no-small --steps 1 --warmup 0 --no-batch > $OUT/a.out 2>&1 || exit 1
SQ2="SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_INSTS_VALU"
timeout -s KILL 200 rocprofv3 --pmc $SQ2 --output-format csv -d $OUT/pmc_sq2_1m -- python3 $ROOT/bench.py --no-cpu --no-small --steps 1 --warmup 0 --no-batch > $OUT/b.out 2>&1 || exit 1
find $OUT -name "*.db" -delete
echo done
