#!/bin/bash
# Same-box A/B of the training step: `bench.py` under different kernel switches (arguments: one quoted env list per run).
B="python bench.py --decode-tokens 0 --decode-full 0 --no-cpu-baseline --steps 8 --warmup 3"
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg timeout -k 10 200 $B > gpurun_out/step_ab_tmp.log 2>&1 || { tail -5 gpurun_out/step_ab_tmp.log; exit 1; }
  python tools/show_bench.py gpurun_out/step_ab_tmp.log | head -1
done
