for t in 1.5 4 8; do
  echo "thr $t"
  AS_MULTI_SLICE_THR=$t python bench.py --steps 80 --warmup 16 --cpu-utts 0 --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['ms_per_step_one_chain_alone'], d['roofline']['gemm_ms_per_step_by_events'])"
done
