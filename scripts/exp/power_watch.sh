#!/bin/bash
# what the chip's power management does during the benchmark's GEMM-heavy loop: socket power, power cap and shader clock sampled with rocm-smi
# while (a) the conv GEMM runs back to back on random operands, (b) on zero operands, (c) the C3 step replays
R=${GRAFT_REPO_ROOT:-/root/repo}
sample() { for i in 1 2 3 4 5 6; do rocm-smi --showpower --showclocks --showmaxpower 2>/dev/null | grep -E "Power|sclk|Max Graphics" | tr -s ' ' | tr '\n' ';'; echo; sleep 0.7; done; }
echo "== idle"; rocm-smi --showpower --showclocks --showmaxpower 2>/dev/null | grep -E "Power|sclk|Max" | tr -s ' '
for z in 0 1; do
  echo "== conv GEMM M1024 N6400 K1024 T3 back to back, ZERO=$z"
  ZERO=$z LOOP=1 python3 $R/scripts/exp/gemm_loop.py 6 & 
  sleep 2.0; sample; wait
done
