#!/bin/bash
# Same-box A/B of the training step: each line is `bench.py` with one of the new GEMM kernels switched off.
B="python bench.py --decode-tokens 0 --decode-full 0 --no-cpu-baseline --steps 8 --warmup 3"
for cfg in "MGEN_GEMM_RS=1 MGEN_GEMM_OS=1" "MGEN_GEMM_RS=1 MGEN_GEMM_OS=0" "MGEN_GEMM_RS=0 MGEN_GEMM_OS=1" "MGEN_GEMM_RS=0 MGEN_GEMM_OS=0" "MGEN_GEMM_RS=1 MGEN_GEMM_OS=1"; do
  echo "== $cfg"
  env $cfg timeout -k 10 200 $B > gpurun_out/step_ab_tmp.log 2>&1 || { tail -5 gpurun_out/step_ab_tmp.log; exit 1; }
  python tools/show_bench.py gpurun_out/step_ab_tmp.log | head -1
done
