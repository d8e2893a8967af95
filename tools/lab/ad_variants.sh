#!/bin/bash
# decode step time with variant libraries tools/lab/libmgen_ad_<U>_<OCC>.so of the attention kernel, interleaved repeats
cd /root/repo
for rep in 1 2 3; do
for v in 4_5 2_6 4_4; do
    echo "== $v parts 2 rep $rep"
    MGEN_LIB=$PWD/tools/lab/libmgen_ad_$v.so MGEN_DECODE_PARTS=2 LEGS=long timeout -k 10 120 python tools/bench_decode.py 2 || exit 1
done
done
