#!/bin/bash
# Diagnostic (GPU box): the headline bench line for each variant library given (FTMPC_LIB selects the library the host side loads).
for lib in "$@"; do
  echo "== $lib"
  FTMPC_LIB=$GRAFT_REPO_ROOT/fault-tolerant-mpc_amd/ft_mpc_amd/$lib timeout -k 10 100 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('value %.0f  ms %.3f  kernel_ms %.3f  iters %.2f  notconv %d  other %s' % (d['value'], d['ms_per_step'], r['kernel_ms'], d['config']['ipm_iters_mean'], d['config']['not_converged'], r['other_kernels_ms']))"
done
