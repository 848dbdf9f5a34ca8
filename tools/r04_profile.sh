# Round-4 profiling pass (through gpurun): tools/r03_profile.sh with the round's workloads -- the drift step at the four sweep sizes
# (SQ passes for 65 536 and 1 M envs too), elevation with the LDS-patch scan (default) AND with the gather scan (WL_FLAGS = 8) at
# 262 144 envs, visual, depth, the visual-depth extension task -- + kernel statistics of bench.py.  Digest: gpurun_out/r04/r04_pmc.json.
R=$GRAFT_REPO_ROOT
export WL_PMC_TIMEOUT=200
WL_PMC_WORK="drift:4096:64 drift:65536:16 drift:1048576:8 drift:4194304:4 elev:4096:32 elev:262144:4 visual:4096:16 visual:262144:2 depth:4096:8 visual_depth:4096:8" bash $R/tools/r03_profile.sh r04
WL_FLAGS=8 WL_PMC_NO_STATS=1 WL_PMC_WORK="elev:262144:4" bash $R/tools/r03_profile.sh r04_scan_gather > /dev/null
ls $R/gpurun_out/r04 | head -80
