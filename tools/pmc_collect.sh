#!/bin/bash
# Counter and trace passes behind profiles/rNN_pmc_traffic.json and the kernel-stats files: `bash tools/pmc_collect.sh r02`
# (on the GPU box; results under gpurun_out/<tag>/). One rocprofv3 run per counter, never combined with a trace.
set -e
TAG=$1
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
TRAIN="$R/bench.py --steps 1 --warmup 1 --decode-tokens 0 --no-cpu-baseline"
DEC="$R/tools/profile_decode.py run 256 1792 17"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train_trace -o t -- python3 $R/bench.py --steps 4 --warmup 2 --decode-tokens 0 --no-cpu-baseline > $OUT/train_trace.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/train_fetch -o t -- python3 $TRAIN > $OUT/train_fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/train_write -o t -- python3 $TRAIN > $OUT/train_write.log 2>&1
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dec_trace -o t -- python3 $DEC > $OUT/dec_trace.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/dec_fetch -o t -- python3 $DEC > $OUT/dec_fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/dec_write -o t -- python3 $DEC > $OUT/dec_write.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/train_mfma -o t -- python3 $TRAIN > $OUT/train_mfma.log 2>&1
cd $R
python3 tools/pmc_traffic.py $OUT/pmc_traffic.json \
  train=$OUT/train_fetch/t_counter_collection.csv,$OUT/train_write/t_counter_collection.csv,$OUT/train_trace/t_kernel_trace.csv \
  decode=$OUT/dec_fetch/t_counter_collection.csv,$OUT/dec_write/t_counter_collection.csv,$OUT/dec_trace/t_kernel_trace.csv \
  decode_streams=256
# decode traffic at three more contexts (9 steps each): the line through them gives the literal run's average
SPECS=""
for CTX in 256 2048 3840; do
  D2="$R/tools/profile_decode.py run 256 $CTX 9"
  timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/d${CTX}_trace -o t -- python3 $D2 > $OUT/d_trace.log 2>&1
  timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/d${CTX}_fetch -o t -- python3 $D2 > $OUT/d_fetch.log 2>&1
  timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/d${CTX}_write -o t -- python3 $D2 > $OUT/d_write.log 2>&1
  SPECS="$SPECS decode@$CTX=$OUT/d${CTX}_fetch/t_counter_collection.csv,$OUT/d${CTX}_write/t_counter_collection.csv,$OUT/d${CTX}_trace/t_kernel_trace.csv"
done
python3 tools/pmc_traffic.py $OUT/pmc_decode_contexts.json decode_streams=256 $SPECS
rm -rf $OUT/d[0-9]*_trace $OUT/d[0-9]*_fetch $OUT/d[0-9]*_write
python3 tools/pmc_sum.py $OUT/train_mfma attn_ > $OUT/attn_mfma_busy.txt
python3 tools/profile_decode.py report $OUT/dec_trace/t_kernel_trace.csv > $OUT/decode_steps_only.txt
cp $OUT/train_trace/t_kernel_stats.csv $OUT/train_step_kernel_stats.csv
cp $OUT/dec_trace/t_kernel_stats.csv $OUT/decode_kernel_stats.csv
# the raw per-dispatch files are large: keep the summaries only
rm -rf $OUT/train_trace $OUT/train_fetch $OUT/train_write $OUT/dec_trace $OUT/dec_fetch $OUT/dec_write $OUT/train_mfma
ls -la $OUT
