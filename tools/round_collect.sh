# copy what tools/round_final.sh left in gpurun_out/<tag>/ into profiles/ (tracked): the two bench lines + their detail files, the
# kernel statistics of bench.py, the PMC digest + raw counter CSVs, the training histories.     usage: tools/round_collect.sh <tag>
set -e
TAG=${1:?usage: round_collect.sh <tag>}
O=gpurun_out/$TAG
cp $O/${TAG}_pmc.json profiles/${TAG}_pmc.json
cp $O/${TAG}_bench_kernel_stats.csv profiles/
[ -f $O/${TAG}_bench_sweep_kernel_stats.csv ] && cp $O/${TAG}_bench_sweep_kernel_stats.csv profiles/
rm -rf profiles/${TAG}_pmc && mkdir -p profiles/${TAG}_pmc
for d in $O/*/; do
  t=$(basename $d)
  f=$(find $d -name "*counter_collection.csv" | head -1); if [ -n "$f" ]; then cp $f profiles/${TAG}_pmc/${t}.csv; fi
done
cp $O/bench.json profiles/${TAG}_bench_n1.json
cp $O/bench_detail.json profiles/${TAG}_bench_n1_detail.json
cp $O/bench_s20.json profiles/${TAG}_bench_n1_steps20.json
cp $O/bench_s20_detail.json profiles/${TAG}_bench_n1_steps20_detail.json
for f in $O/train_*_CONFIG.json; do b=$(basename $f .json); b=${b#train_}; cp $f profiles/${TAG}_train_$(echo $b | tr 'A-Z' 'a-z')_8it.json; done
ls profiles | grep ${TAG}
