"""Round 6: in the fused elevation launch and in the quad-form visual step on a heightfield the block's RESET draws are made by an
otherwise idle wavefront while the physics runs (wl_elev.hip FusedHooks::reset, wl_visual.hip HelperReset) and handed over through
LDS.  The lane-form kernels draw the same values on the spot.  A reset pose does not depend on the physics, so with episodes of three
steps -- a third of the batch resetting in every step -- the poses the two forms write for the resetting envs must agree bit for bit."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _run(make, n, steps, lanes):
    env = make()
    env.p.max_episode_length = 3
    env.set_lanes(lanes)
    env.reset()
    env.episode_len[:n] = torch.arange(n, device=DEV, dtype=torch.int32) % 3          # staggered: resets in every step
    g = torch.Generator(device=DEV).manual_seed(5)
    out = []
    for _ in range(steps):
        a = torch.rand(n, 2, device=DEV, generator=g) * 2 - 1
        env.step(a)
        torch.cuda.synchronize()
        out.append((env.state[:7, :n].clone(), env.truncated[:n].clone(), env.terminated[:n].clone()))
    return out


@pytest.mark.parametrize("task,n", [("elev", 4096), ("elev", 1000), ("visual_depth", 4096), ("visual_depth", 250)])
def test_reset_poses_from_the_helper_wavefront_equal_the_inline_draw(task, n):
    from wheeledlab_amd.core import ElevBatch, VisualDepthBatch
    make = (lambda: ElevBatch(n, device=DEV, seed=9)) if task == "elev" else (lambda: VisualDepthBatch(n, device=DEV, seed=9))
    quad, lane = _run(make, n, 6, 4), _run(make, n, 6, 1)
    seen = 0
    for (sq, tq, dq), (sl, tl, dl) in zip(quad, lane):
        # the time-outs are a function of the step count alone: the same envs reset in both forms whatever their physics did
        both = tq & tl
        assert int(both.sum()) >= n // 5          # (an env that terminated earlier re-starts its count: fewer than a third)
        seen += int(both.sum())
        assert torch.equal(sq[:, both], sl[:, both]), task      # position + quaternion of every resetting env: bit for bit
    assert seen >= n
