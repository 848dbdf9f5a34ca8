# per-kernel durations of the elevation launches at 262 144 envs (rocprofv3 --kernel-trace --stats), both scan forms
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04scanstats; mkdir -p $O
cat > /tmp/obs_loop.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from wheeledlab_amd.core import ElevBatch
n = int(sys.argv[1]); fl = int(os.environ.get("WL_FLAGS", "0"))
env = ElevBatch(n, device="cuda:0", seed=42); env.reset(); env.set_lanes(1); env.set_flags(fl)
a = torch.rand(4, n, 2, device="cuda:0") * 2 - 1
env.rollout(a)
for _ in range(6): env.observe()
torch.cuda.synchronize()
PY
for form in gather:8 lds:4; do
  name=${form%%:*}; export WL_FLAGS=${form#*:}
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -- python /tmp/obs_loop.py 262144 > $O/$name.log 2>&1
  f=$(find $O/$name -name "*kernel_stats.csv" | head -1); echo "== $name"; cut -d, -f1-4 $f | head -12
done
