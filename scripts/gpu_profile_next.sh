#!/bin/bash
# rocprofv3 kernel traces of the section-8(f) rows (JDCNet, EMA_Predictor, HiFi-GAN generator): run on the GPU box through gpurun.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out
mkdir -p $OUT
for n in jdc ema vocoder; do
  rocprofv3 --kernel-trace --stats -d $OUT/prof_$n -o $n --output-format csv -- python3 $R/scripts/${n}_bench.py > $OUT/prof_$n.log 2>&1
  tail -2 $OUT/prof_$n.log
done
