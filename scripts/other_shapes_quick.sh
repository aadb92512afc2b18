#!/bin/bash
# Diagnostic (GPU box): bench lines of the shapes served by kernel 2's NB = 9 / 10 instantiations and the config-4 shard.
for args in "--batch 4096 --faults 1" "--faults 0" "--batch 32768"; do
  timeout -k 10 150 python3 bench.py $args --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('$args: value %.0f  ms %.3f  %s %.3f ms  iters %.2f' % (d['value'], d['ms_per_step'], r['kernel'], r['kernel_ms'], d['config']['ipm_iters_mean']))"
done
