# copy the digests of the last tools/r03_profile.sh run (gpurun_out/r03) and the bench lines of gpurun_out/<run> into profiles/
#   usage: tools/r03_collect.sh r03h
R=${1:?run directory under gpurun_out}
cp gpurun_out/r03/r03_pmc.json profiles/r03_pmc.json
cp gpurun_out/r03/r03_bench_kernel_stats.csv profiles/
rm -rf profiles/r03_pmc && mkdir -p profiles/r03_pmc
for d in gpurun_out/r03/*/; do t=$(basename $d); f=$(ls -t $(find $d -name "*counter_collection.csv") 2>/dev/null | head -1); [ -n "$f" ] && cp $f profiles/r03_pmc/${t}.csv; done
cp gpurun_out/$R/bench.json profiles/r03_bench_n1.json
cp gpurun_out/$R/bench_s20.json profiles/r03_bench_n1_steps20.json
