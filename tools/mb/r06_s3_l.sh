# session 3, call L: split counts of the chained launch when the heads do not fill the chip at eight splits (24 / 16 / 8 / 4 heads)
cd $GRAFT_REPO_ROOT
for cfg in "24 0" "24 10" "24 9" "16 0" "16 12" "16 16" "8 0" "8 16" "8 32" "4 0" "4 16" "4 32" "4 64" "32 0"; do
  set -- $cfg
  CHAIN_SPLITS=$2 timeout 300 python tools/mb/chain_bench.py $1 2081 32 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-215
done
