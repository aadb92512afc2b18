#!/bin/bash
# Runs on the GPU box: for every BASELINE shape (and the reference vehicle) one bench line, a rocprofv3 kernel-trace stats pass
# and PMC passes (FETCH_SIZE / WRITE_SIZE in separate passes, SQ counters); outputs under gpurun_out/<tag>/<shape>/.
# scripts/summarize_profile.py <tag> turns them into the committed profiles/<tag>_* files.
set -e
TAG=${1:-r03_v1}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY"
SQ2="SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD"
shape() {   # name, bench arguments...
  local name=$1; shift
  local D=$OUT/$name
  mkdir -p $D
  python3 bench.py "$@" > $D/bench.json 2> $D/bench.err || true
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -- python3 bench.py "$@" --steps 5 --warmup 2 --no-cpu-baseline > $D/bench_prof.json 2> $D/prof.err || true
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $D/pmc_$c -- python3 bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $D/pmc_$c.err || true
  done
  timeout -k 10 200 rocprofv3 --pmc $SQ1 --output-format csv -d $D/pmc_SQ -- python3 bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $D/pmc_SQ.err || true
  timeout -k 10 200 rocprofv3 --pmc $SQ2 --output-format csv -d $D/pmc_SQ2 -- python3 bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $D/pmc_SQ2.err || true
  echo "$name done" >> $OUT/progress.log
}
script_shape() {   # name, script + arguments (no bench line: kernel stats and counters only)
  local name=$1; shift
  local D=$OUT/$name
  mkdir -p $D
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -- python3 "$@" > $D/run.log 2> $D/prof.err || true
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $D/pmc_$c -- python3 "$@" > /dev/null 2> $D/pmc_$c.err || true
  done
  timeout -k 10 200 rocprofv3 --pmc $SQ1 --output-format csv -d $D/pmc_SQ -- python3 "$@" > /dev/null 2> $D/pmc_SQ.err || true
  timeout -k 10 200 rocprofv3 --pmc $SQ2 --output-format csv -d $D/pmc_SQ2 -- python3 "$@" > /dev/null 2> $D/pmc_SQ2.err || true
  echo "$name done" >> $OUT/progress.log
}
shape headline
shape config2 --batch 4096 --faults 1 --no-cpu-baseline
shape config4shard --batch 32768 --no-cpu-baseline
shape config5shard --horizon 40 --thrusters 16 --batch 2048 --dtype f64 --no-cpu-baseline
shape refvehicle --horizon 15 --thrusters 16 --batch 4096 --no-cpu-baseline
shape refvehicle_f64 --horizon 15 --thrusters 16 --batch 4096 --dtype f64 --no-cpu-baseline
shape n20nt16 --horizon 20 --thrusters 16 --batch 4096 --no-cpu-baseline
shape nominal8 --faults 0 --no-cpu-baseline
script_shape wrench_hull32 scripts/wrench_trace.py f32       # the two-stage step: linearise, kernel 11, allocation (N = 15, 16 thrusters, B = 16 384)
script_shape wrench_f64 scripts/wrench_trace.py f64      # the same on a float64 handle: kernel 13 (Riccati recursion)
script_shape wrench_f64_n20 scripts/wrench_trace.py f64 20      # ... at BASELINE's horizon
shape config5_16k --horizon 40 --thrusters 16 --batch 16384 --dtype f64 --no-cpu-baseline      # config 5's whole-node batch on one GPU (eight instances per resident wave)
cat $OUT/headline/bench.json
