# copy the digests of the last tools/r04_profile.sh run (gpurun_out/r04, gpurun_out/r04_scan_gather) and the bench lines / training
# histories of tools/r04_final.sh (gpurun_out/r04f) into profiles/            usage: tools/r04_collect.sh
set -e
cp gpurun_out/r04/r04_pmc.json profiles/r04_pmc.json
cp gpurun_out/r04_scan_gather/r04_scan_gather_pmc.json profiles/r04_scan_gather_form_pmc_digest.json
cp gpurun_out/r04/r04_bench_kernel_stats.csv profiles/
rm -rf profiles/r04_pmc && mkdir -p profiles/r04_pmc
for d in gpurun_out/r04/*/ gpurun_out/r04_scan_gather/*/; do
  t=$(basename $d); case $d in *r04_scan_gather*) t=gatherscan_$t;; esac
  f=$(find $d -name "*counter_collection.csv" | head -1); if [ -n "$f" ]; then cp $f profiles/r04_pmc/${t}.csv; fi
done
if [ -f gpurun_out/r04f/bench.json ]; then
  cp gpurun_out/r04f/bench.json profiles/r04_bench_n1.json
  cp gpurun_out/r04f/bench_s20.json profiles/r04_bench_n1_steps20.json
  for f in gpurun_out/r04f/train_*_CONFIG.json; do b=$(basename $f .json); b=${b#train_}; cp $f profiles/r04_train_$(echo $b | tr 'A-Z' 'a-z')_8it.json; done
fi
ls profiles | grep r04
