# per-workgroup loop ticks (s_memtime) of stamp + knockout builds (scripts/build_exp.sh s_NAME -DX6_EXP_STAMPS -DX6_EXP_...)
for f in artspeech_amd/lib/exp_x6_STAMPS.so artspeech_amd/lib/exp_s_*.so; do
  echo "== $f"
  AS_LIB_PATH=$PWD/$f timeout 120 python scripts/exp/x6_stamps.py 2>&1 | grep "tile22\|tile21" | head -2 | cut -c1-150
done
