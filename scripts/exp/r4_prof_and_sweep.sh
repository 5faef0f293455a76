for n in 2 3 5 6; do python bench.py --steps 60 --warmup 12 --no-extras --cpu-utts 0 --in-flight $n 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('in-flight', d['config']['in_flight'], d['ms_per_step'])" ; done > gpurun_out/r4_inflight_sweep.log 2>&1
bash scripts/gpu_profile.sh > gpurun_out/gpu_profile.log 2>&1
