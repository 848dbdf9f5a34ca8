# MFMA-pipe counters of the matrix-pipe kernels (policy step on wide observations, PPO gradient kernel): busy cycles of
# the matrix pipe against the launch's cycles.  Separate passes, kernel-trace / stats off (gpurun rule).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cat > /tmp/mfma_work.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from wheeledlab_amd.policy import ActorCritic
from wheeledlab_amd.rl.ppo import ActorCritic as TorchAC, PPO, FusedPpoStep
dev = "cuda:0"
for D, n, act in ((3208, 4096, "elu"), (689, 4096, "relu"), (3208, 16384, "elu")):
    ac = ActorCritic(D, 2, act, device=dev, seed=0)
    obs = torch.randn(n, D, device=dev)
    a, mu = torch.empty(n, 2, device=dev), torch.empty(n, 2, device=dev)
    lp, v = torch.empty(n, device=dev), torch.empty(n, device=dev)
    for i in range(6):
        ac.act(obs, a, mu, lp, v, 1, i)
torch.manual_seed(0)
B = 524288
tac = TorchAC(14, 14, 2).to(dev)
fz = FusedPpoStep(tac, PPO(tac))
flat = dict(obs=torch.randn(B, 14, device=dev), actions=torch.randn(B, 2, device=dev), mu=torch.randn(B, 2, device=dev),
            logp=torch.randn(B, device=dev) - 2, adv=torch.randn(B, device=dev), returns=torch.randn(B, device=dev),
            values=torch.randn(B, device=dev))
perm = torch.randperm(B, device=dev).to(torch.int32)
sig = torch.ones(2, device=dev)
for i in range(4):
    fz.minibatch(flat, perm, i * 131072, 131072, sig)
torch.cuda.synchronize()
PY
timeout 200 rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_mfma -- python /tmp/mfma_work.py > $R/gpurun_out/pmc_mfma.log 2>&1
python - <<'PY'
import csv, glob, json, os, collections
R = os.environ["GRAFT_REPO_ROOT"]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(R + "/gpurun_out/pmc_mfma/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "actor_critic_act" in k or "ppo_grad" in k:
            key = ("actor_critic_act grid " + r["Grid_Size"]) if "actor_critic_act" in k else "ppo_grad_kernel"
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            acc[key]["ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = {}
for k, d in acc.items():
    m = {c: sum(v[1:]) / max(len(v) - 1, 1) for c, v in d.items()}     # skip the first (cold) launch
    if m.get("GRBM_GUI_ACTIVE"):
        # matrix-pipe busy cycles summed over the 1024 SIMDs against the launch's cycles (GRBM counts per XCD: / 8)
        m["mfma_busy_fraction"] = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (m["GRBM_GUI_ACTIVE"] / 8 * 1024)
    out[k] = m
json.dump(out, open(R + "/gpurun_out/pmc_mfma/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
