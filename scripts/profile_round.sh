#!/bin/bash
# Runs on the GPU box: bench + rocprofv3 kernel-trace stats + PMC passes; outputs under gpurun_out/<tag>/
set -e
TAG=${1:-r01_v2}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_prof.json 2> $OUT/prof.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_$c.err
done
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_SQ -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_SQ.err
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_SQ2 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_SQ2.err || true
# the reference's own vehicle (N = 15, 16 thrusters, two faults, B = 4096): kernel 8
python3 bench.py --horizon 15 --thrusters 16 --batch 4096 --no-cpu-baseline > $OUT/bench_refvehicle.json 2> $OUT/bench_refvehicle.err || true
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_refvehicle -- python3 bench.py --horizon 15 --thrusters 16 --batch 4096 --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> $OUT/prof_refvehicle.err || true
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmcrv_SQ -- python3 bench.py --horizon 15 --thrusters 16 --batch 4096 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmcrv_SQ.err || true
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $OUT/pmcrv_$c -- python3 bench.py --horizon 15 --thrusters 16 --batch 4096 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmcrv_$c.err || true
done
cat $OUT/bench.json
