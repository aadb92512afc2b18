#!/bin/bash
# Diagnostic (GPU box): SQ / TCC counters of the headline kernel for two builds of the library.
# usage: scripts/pmc_compare.sh <outdir> <lib or ""> [<lib> ...]
OUT=$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for L in "$@"; do
  i=$((i+1))
  if [ "$L" != "default" ]; then export FTMPC_LIB=$L; else unset FTMPC_LIB; fi
  timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/sq_$i -- python3 scripts/quick_perf.py 65536 2 > $OUT/sq_$i.log 2>&1
  timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d $OUT/sq2_$i -- python3 scripts/quick_perf.py 65536 2 > $OUT/sq2_$i.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $OUT/${c}_$i -- python3 scripts/quick_perf.py 65536 2 > $OUT/${c}_$i.log 2>&1
  done
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for d in sorted(glob.glob(out + "/*_[0-9]")):
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "solve_f32_kernel<8>" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(d.split("/")[-1], {k: "%.4g" % (sum(v) / len(v)) for k, v in agg.items()})
PY
