#!/bin/bash
# rocprofv3 passes of the benchmark (run on the GPU box through gpurun; condensed into profiles/ by scripts/make_profiles.py).
#   prof_trace    the C3 call of the headline arrangement (two batches of 32 per call), eager, one merged chain: a kernel's duration is its own
#   prof_trace32  the same for ONE batch of 32 per call (rounds 1-4's step)
#   prof_graph    the command the driver times (hipGraph replay, branches on side streams), extras (MAS, C2, C5, transfers) included
#   prof_c5       the long-form configuration
#   prof_c2       batch 1 (one utterance, 150 frames), one merged chain, eager
#   prof_pmc_*    counter passes (kernel trace only, one counter group per run)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out
mkdir -p $OUT
# (--call-batches 2: a step of these eager passes is ONE call over two batches of 32 -- what a coalescing lane of the headline arrangement launches)
ARGS="--steps 10 --warmup 2 --no-graph --no-concurrency --cpu-utts 0 --no-extras --call-batches 2"
rocprofv3 --kernel-trace --stats -d $OUT/prof_trace -o bench --output-format csv -- python3 $R/bench.py $ARGS > $OUT/prof_trace.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/prof_trace32 -o bench --output-format csv -- python3 $R/bench.py --steps 10 --warmup 2 --no-graph --no-concurrency --cpu-utts 0 --no-extras > $OUT/prof_trace32.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/prof_graph -o bench --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --cpu-utts 0 > $OUT/prof_graph.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/prof_c5 -o bench --output-format csv -- python3 $R/bench.py --config C5 --no-extras --steps 5 --warmup 2 --no-graph --no-concurrency --cpu-utts 0 > $OUT/prof_c5.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/prof_c2 -o bench --output-format csv -- python3 $R/bench.py --config C2 --no-extras --steps 20 --warmup 2 --no-graph --no-concurrency --cpu-utts 0 > $OUT/prof_c2.log 2>&1
# the test.py chain around the path: the frozen extractors and the HiFi-GAN generator at the C3 batch, and the whole
# chain (bench.py's `surface` object)
rocprofv3 --kernel-trace --stats -d $OUT/prof_jdc -o bench --output-format csv -- python3 $R/scripts/jdc_bench.py > $OUT/prof_jdc.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/prof_ema -o bench --output-format csv -- python3 $R/scripts/ema_bench.py > $OUT/prof_ema.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/prof_vocoder -o bench --output-format csv -- python3 $R/scripts/vocoder_bench.py > $OUT/prof_vocoder.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/prof_surface -o bench --output-format csv -- python3 $R/bench.py --surface-only > $OUT/prof_surface.log 2>&1
PARGS="--steps 2 --warmup 1 --no-graph --no-concurrency --cpu-utts 0 --no-extras --call-batches 2"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/prof_pmc_sq -o bench --output-format csv -- python3 $R/bench.py $PARGS > $OUT/prof_pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/prof_pmc_fetch -o bench --output-format csv -- python3 $R/bench.py $PARGS > $OUT/prof_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/prof_pmc_write -o bench --output-format csv -- python3 $R/bench.py $PARGS > $OUT/prof_pmc_write.log 2>&1
# the library's own per-launch HIP-event log of the conv GEMM (class, shape / tile / split tag, ms, algorithmic flop and bytes)
rm -f $OUT/gemm_launches_events.csv
# (--call-batches 2: every instrumented call is ONE as_forward_test over two batches of 32 -- the headline arrangement's call width, whatever arrangement a run adopts)
AS_PROF_CSV=$OUT/gemm_launches_events.csv python3 $R/bench.py --steps 5 --warmup 2 --cpu-utts 0 --no-extras --call-batches 2 > $OUT/prof_events.log 2>&1
# the multi-rank launch path on this one-GPU box (two ranks on GPU 0, gloo for the barrier: AS_BENCH_TEST_ONE_GPU=1), weak scaling and C4
cd $R
for mode in weak c4; do
  if [ $mode = weak ]; then EXTRA="--steps 20 --warmup 5 --no-extras"; else EXTRA="--steps 3 --warmup 1 --global-batch 256"; fi
  AS_BENCH_TEST_ONE_GPU=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 2 --cpu-utts 0 $EXTRA > $OUT/two_ranks_$mode.log 2>&1
done
cd /tmp
# the stats CSVs are small; the raw traces are not: keep only what make_profiles.py reads
for d in prof_trace prof_trace32 prof_graph prof_c5 prof_c2 prof_jdc prof_ema prof_vocoder prof_surface; do rm -f $OUT/$d/bench_kernel_trace.csv; done
ls $OUT/prof_trace $OUT/prof_graph $OUT/prof_c5 $OUT/prof_pmc_sq | head -20
for f in prof_trace prof_graph prof_c5 two_ranks_weak two_ranks_c4; do grep '^{' $OUT/$f.log | head -1 | cut -c1-400; done
