cd $GRAFT_REPO_ROOT
for opts in "batch_blocks=64" "batch_blocks=64 wave_budget=20" "batch_blocks=64 wave_budget=24" "batch_blocks=64 wave_budget=28" "batch_blocks=64 wave_budget=32"; do
  echo "== peptides $opts"
  timeout -k 10 120 python tools/sweeps/r4_peptides.py 2000000 $opts 2>&1 | grep "lq" || exit 1
done
