"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/wheeledlab_amd.h
declares, the ctypes structs match the header's layout, and the product's default parameters equal the oracle's."""
import ctypes as C
import os
import re
import subprocess

import pytest

from conftest import ROOT
from oracle import params as OP
from wheeledlab_amd import _abi as A
from wheeledlab_amd import params as PP

HEADER = os.path.join(ROOT, "include", "wheeledlab_amd.h")


def _ensure_built():
    import __graft_entry__ as g
    g.build()  # incremental: recompiles only when a source is newer than the library


def test_library_exports_every_declared_symbol():
    _ensure_built()
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)  # prose in comments also mentions the entry points
    declared = set(re.findall(r"\b(wl_[a-z0-9_]+)\s*\(", src))
    assert declared == set(A.SIGNATURES), (declared ^ set(A.SIGNATURES))
    lib = A.load()
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.wl_version() == A.WL_ABI_VERSION
    assert lib.wl_strerror(-3).decode().startswith("buffer alignment")


def test_struct_layout_matches_header(tmp_path):
    """compile a tiny C program against the header and compare sizeof / offsetof with ctypes"""
    probe = tmp_path / "probe.c"
    probe.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "wheeledlab_amd.h"\n'
        "int main(){printf(\"%zu %zu %zu %zu %zu %zu %zu %zu %zu %d %d\\n\", sizeof(WlDriftParams), sizeof(WlVehicleParams),"
        " sizeof(WlActionParams), sizeof(WlEnvBuffers), sizeof(WlStepOut), offsetof(WlDriftParams, vehicle),"
        " offsetof(WlDriftParams, weight), offsetof(WlDriftParams, log_episode_sums), offsetof(WlEnvBuffers, stride),"
        " (int)WL_S_COUNT, (int)WL_M_COUNT);return 0;}\n")
    exe = tmp_path / "probe"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(probe), "-o", str(exe)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    want = [C.sizeof(A.WlDriftParams), C.sizeof(A.WlVehicleParams), C.sizeof(A.WlActionParams), C.sizeof(A.WlEnvBuffers),
            C.sizeof(A.WlStepOut), A.WlDriftParams.vehicle.offset, A.WlDriftParams.weight.offset,
            A.WlDriftParams.log_episode_sums.offset, A.WlEnvBuffers.stride.offset, A.S_COUNT, A.M_COUNT]
    assert got == want


def _cmp(prod, orc, path=""):
    for k, v in prod.items():
        o = getattr(orc, k)
        if isinstance(v, dict):
            _cmp(v, o, path + k + ".")
        elif isinstance(v, list):
            assert [float(x) for x in o] == pytest.approx(v, rel=1e-6), path + k
        else:
            assert float(o) == pytest.approx(float(v), rel=1e-6), path + k


def test_product_defaults_equal_oracle_defaults():
    _cmp(PP.struct_to_dict(PP.drift_params()), OP.drift_params())


def test_invalid_arguments_are_rejected_without_a_gpu():
    _ensure_built()
    lib = A.load()
    p = PP.drift_params()
    b = A.WlEnvBuffers()  # all null
    out = A.WlStepOut()
    assert lib.wl_drift_step(C.byref(p), C.byref(b), None, None, C.byref(out), 0, 0, None) == -1
    assert lib.wl_action_map(C.byref(p.action), 0, None, None, None, None, None) == -1
    assert lib.wl_philox_uniform(0, 0, 0, 0, None, None) == -1
    # rollout-side entry points: null nets / wrong shapes are refused before any launch
    buf = (C.c_float * 8192)()
    base = C.addressof(buf)
    net = lambda i, o: A.WlMlp(base, base, base, base, base, base, i, o, 64, A.ACT_ELU)
    actor, critic, null = net(689, 2), net(689, 1), A.WlMlp()
    act = lib.wl_actor_critic_act
    assert act(C.byref(null), C.byref(critic), base, 4, base, 689, base, base, base, base, 0, 0, 0, 0, 3, None) == -1
    assert act(C.byref(actor), C.byref(net(700, 1)), base, 4, base, 700, base, base, base, base, 0, 0, 0, 0, 3, None) == -1   # widths differ
    assert act(C.byref(actor), C.byref(critic), base, 4, base, 688, base, base, base, base, 0, 0, 0, 0, 3, None) == -1          # stride < in_dim
    assert act(C.byref(actor), C.byref(critic), base, 4, base, 689, base + 4, base, base, base, 0, 0, 0, 0, 3, None) == -3      # actions not 8-byte aligned
    assert act(C.byref(actor), C.byref(critic), base, 4, base, 689, base, base, base, base, 0, 0, 0, 0, 4, None) == -1          # nets not in 1..3
    assert act(C.byref(actor), C.byref(critic), base, 4, base, 689, base, base, base, None, 0, 0, 0, 0, 2, None) == -1          # critic half without values
    hp, st = A.WlPpoParams(), A.WlPpoState()
    small_a, small_c = net(14, 2), net(14, 1)
    assert lib.wl_ppo_apply(C.byref(small_a), C.byref(small_c), base, 64, C.byref(hp), C.byref(st), 0, 1, None) == -1          # null state
    st = A.WlPpoState(base, base, base, base, base, base)
    assert lib.wl_ppo_apply(C.byref(small_a), C.byref(small_c), base, 64, C.byref(hp), C.byref(st), 2, 1, None) == -1          # parity
    assert lib.wl_ppo_apply(C.byref(small_a), C.byref(small_c), base, 0, C.byref(hp), C.byref(st), 0, 1, None) == -1           # empty batch
    assert lib.wl_ppo_apply(C.byref(actor), C.byref(critic), base, 64, C.byref(hp), C.byref(st), 0, 1, None) == -1             # not the 14-wide nets
    # depth ray-cast: pyramid sizing is host arithmetic; every malformed call is refused before a launch
    assert lib.wl_heightfield_pyramid_floats(800, 800) == 1024 * 1024 // 2 + 800 * 800 // 2 + 4      # bound pyramid (4-byte entries) + 16-bit height codes + header
    assert lib.wl_heightfield_pyramid_floats(3, 3) == 4 + 5 + 4 and lib.wl_heightfield_pyramid_floats(349, 613) == 1024 * 1024 // 2 + (349 * 613 + 1) // 2 + 4
    assert lib.wl_heightfield_pyramid_floats(1, 9) == 0 and lib.wl_heightfield_pyramid_floats(9, 16386) == 0
    vp = PP.visual_params()
    hf = A.WlHeightField(base, 16, 16, 0.0, 0.0, 0.5, 0.0, 2.0 ** -13)
    good_b = A.WlEnvBuffers(base, base, None, base, 64, 40, 0, 1, 0, 0)
    dep = lib.wl_visual_depth
    assert dep(C.byref(vp), C.byref(good_b), C.byref(hf), None, 10.0, base, None) == -1               # no pyramid
    assert dep(C.byref(vp), C.byref(good_b), C.byref(hf), base, 0.0, base, None) == -1                # range
    assert dep(C.byref(vp), C.byref(good_b), C.byref(hf), base, 10.0, None, None) == -1               # no output
    assert dep(C.byref(vp), C.byref(b), C.byref(hf), base, 10.0, base, None) == -1                    # null state
    assert dep(C.byref(vp), C.byref(good_b), C.byref(A.WlHeightField(base, 1, 16, 0.0, 0.0, 0.5, 0.0, 2.0 ** -13)), base, 10.0, base, None) == -1
    assert dep(C.byref(vp), C.byref(good_b), C.byref(A.WlHeightField(base, 16, 16, 0.0, 0.0, 0.0, 0.0, 2.0 ** -13)), base, 10.0, base, None) == -1   # cell 0
    assert dep(C.byref(vp), C.byref(good_b), C.byref(A.WlHeightField(base, 16, 16, 0.0, 0.0, 0.5, 0.0, 0.0)), base, 10.0, base, None) == -1   # z_scale 0
    assert lib.wl_heightfield_build_pyramid(C.byref(A.WlHeightField(None, 16, 16, 0.0, 0.0, 0.5, 0.0, 2.0 ** -13)), base, None) == -1
    assert lib.wl_heightfield_build_pyramid(C.byref(hf), None, None) == -1


def test_layout_of_task_structs_and_misuse_codes(tmp_path):
    """elevation / visual parameter structs match the header; stride / alignment violations are refused before launch"""
    probe = tmp_path / "probe2.c"
    probe.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "wheeledlab_amd.h"\n'
        "int main(){printf(\"%zu %zu %zu %zu %zu %zu %d %d\\n\", sizeof(WlElevParams), sizeof(WlVisualParams),"
        " sizeof(WlHeightField), sizeof(WlTravMap), offsetof(WlElevParams, weight), offsetof(WlVisualParams, cam_pos),"
        " (int)WL_ELEV_OBS_DIM, (int)WL_VIS_OBS_DIM);return 0;}\n")
    exe = tmp_path / "probe2"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(probe), "-o", str(exe)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    want = [C.sizeof(A.WlElevParams), C.sizeof(A.WlVisualParams), C.sizeof(A.WlHeightField), C.sizeof(A.WlTravMap),
            A.WlElevParams.weight.offset, A.WlVisualParams.cam_pos.offset, A.ELEV_OBS_DIM, A.VIS_OBS_DIM]
    assert got == want
    _ensure_built()
    lib = A.load()
    p = PP.drift_params()
    buf = (C.c_float * 4096)()
    base = C.addressof(buf)
    out = A.WlStepOut(base, base, base, base, None)
    ok_bufs = dict(state=base, episode_len=base, ref_poses=base, metrics=base)
    bad_stride = A.WlEnvBuffers(stride=100, n_envs=100, env_offset=0, metrics_slots=1, **ok_bufs)       # stride % 64 != 0
    assert lib.wl_drift_step(C.byref(p), C.byref(bad_stride), base, None, C.byref(out), 0, 0, None) == -3
    misaligned = A.WlEnvBuffers(stride=128, n_envs=100, env_offset=0, metrics_slots=1, **{**ok_bufs, "state": base + 4})
    assert lib.wl_drift_step(C.byref(p), C.byref(misaligned), base, None, C.byref(out), 0, 0, None) == -3
    short = A.WlEnvBuffers(stride=64, n_envs=100, env_offset=0, metrics_slots=1, **ok_bufs)              # stride < n_envs
    assert lib.wl_drift_step(C.byref(p), C.byref(short), base, None, C.byref(out), 0, 0, None) == -1
    zero_slots = A.WlEnvBuffers(stride=128, n_envs=100, env_offset=0, metrics_slots=0, **ok_bufs)
    assert lib.wl_drift_step(C.byref(p), C.byref(zero_slots), base, None, C.byref(out), 0, 0, None) == -1
    p.num_ref_points = 33
    good = A.WlEnvBuffers(stride=128, n_envs=100, env_offset=0, metrics_slots=1, **ok_bufs)
    assert lib.wl_drift_step(C.byref(p), C.byref(good), base, None, C.byref(out), 0, 0, None) == -1      # > 32 ref poses
    ep = PP.elev_params()
    assert lib.wl_elev_step(C.byref(ep), C.byref(good), None, base, C.byref(out), 0, 0, None) == -1      # no heightfield
    no_scale = A.WlHeightField(base, 8, 8, 0.0, 0.0, 1.0, 0.0, 0.0)
    assert lib.wl_elev_step(C.byref(ep), C.byref(good), C.byref(no_scale), base, C.byref(out), 0, 0, None) == -1      # z_scale 0
    # persistent elevation collector: observation rows k + 1 must be where the policy of step k + 1 reads them; quad form only
    hfb = A.WlHeightField(base, 8, 8, 0.0, 0.0, 1.0, 0.0, 2.0 ** -13)
    net = lambda o: A.WlMlp(base, base, base, base, base, base, A.ELEV_OBS_DIM, o, 64, A.ACT_ELU)
    na, nc = net(2), net(1)
    io = A.WlCollectIo(base, base, base, base, base)
    nxt = A.WlStepOut(base + 100 * A.ELEV_OBS_DIM * 4, base, base, base, None)
    cr = lib.wl_elev_collect_rollout
    assert cr(C.byref(ep), C.byref(good), C.byref(hfb), C.byref(na), C.byref(nc), base, C.byref(io), C.byref(out), 4, 0, 0, 0, None) == -1   # rows alias
    lanes1e = A.WlEnvBuffers(stride=128, n_envs=100, env_offset=0, metrics_slots=1, lanes=1, **ok_bufs)
    assert cr(C.byref(ep), C.byref(lanes1e), C.byref(hfb), C.byref(na), C.byref(nc), base, C.byref(io), C.byref(nxt), 4, 0, 0, 0, None) == -1
    assert cr(C.byref(ep), C.byref(good), C.byref(hfb), C.byref(na), C.byref(net(3)), base, C.byref(io), C.byref(nxt), 4, 0, 0, 0, None) == -1  # critic out_dim
    assert cr(C.byref(ep), C.byref(good), C.byref(hfb), C.byref(na), C.byref(nc), None, C.byref(io), C.byref(nxt), 4, 0, 0, 0, None) == -1       # no std
    # persistent visual rollout: refused without a map, in the lane form, and with per-step rows that alias
    vp = PP.visual_params()
    tm = A.WlTravMap(base, base, 500, 500, 10, 0.5, 0.5)
    vr = lib.wl_visual_rollout_persistent
    assert vr(C.byref(vp), C.byref(good), None, base, C.byref(out), 100 * A.VIS_OBS_DIM, 100, 2, 0, 0, None) == -1
    lanes1 = A.WlEnvBuffers(stride=128, n_envs=100, env_offset=0, metrics_slots=1, lanes=1, **ok_bufs)
    assert vr(C.byref(vp), C.byref(lanes1), C.byref(tm), base, C.byref(out), 100 * A.VIS_OBS_DIM, 100, 2, 0, 0, None) == -1
    assert vr(C.byref(vp), C.byref(good), C.byref(tm), base, C.byref(out), 0, 0, 2, 0, 0, None) == -1
    ring = A.WlEnvBuffers(stride=128, n_envs=100, env_offset=0, metrics_slots=2, **ok_bufs)
    assert vr(C.byref(vp), C.byref(ring), C.byref(tm), base, C.byref(out), 100 * A.VIS_OBS_DIM, 100, 4, 0, 0, None) == -1   # ring slot aliasing


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(A.HipExtensionMissing):
        A.load(str(tmp_path / "nope.so"))


def test_product_does_not_import_oracle():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "wheeledlab_amd")):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dirpath, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M):
                    bad.append(f)
    assert not bad


def test_layout_of_learner_structs_and_their_argument_checks(tmp_path):
    """round-2 structs (startup randomisation, policy-step scratch, wide PPO state, collection rows) match the header, the
    learner constants agree, and the wide entry points refuse misuse before any launch (no GPU needed)"""
    probe = tmp_path / "probe3.c"
    probe.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "wheeledlab_amd.h"\n'
        "int main(){printf(\"%zu %zu %zu %zu %zu %zu %zu %zu %zu %d %d %d %d\\n\", sizeof(WlStartupParams), sizeof(WlActScratch),"
        " sizeof(WlPpoWideState), sizeof(WlCollectIo), sizeof(WlPpoBatch), sizeof(WlPpoParams), sizeof(WlPpoState),"
        " offsetof(WlPpoWideState, in_dim), offsetof(WlActScratch, dp), (int)WL_PPO_NUM_PARAMS, (int)WL_PPO_PARTIAL_STRIDE,"
        " (int)WL_PPO_OPERAND_FLOATS, (int)WL_ABI_VERSION);return 0;}\n")
    exe = tmp_path / "probe3"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(probe), "-o", str(exe)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    want = [C.sizeof(A.WlStartupParams), C.sizeof(A.WlActScratch), C.sizeof(A.WlPpoWideState), C.sizeof(A.WlCollectIo),
            C.sizeof(A.WlPpoBatch), C.sizeof(A.WlPpoParams), C.sizeof(A.WlPpoState), A.WlPpoWideState.in_dim.offset,
            A.WlActScratch.dp.offset, A.PPO_NUM_PARAMS, A.PPO_PARTIAL_STRIDE, A.PPO_OPERAND_FLOATS, A.WL_ABI_VERSION]
    assert got == want
    _ensure_built()
    lib = A.load()
    # flat parameter count of the D-64-64-2 / D-64-64-1 pair (std included): the drift agents' constant at D = 14
    assert lib.wl_ppo_wide_num_params(14) == A.PPO_NUM_PARAMS
    assert lib.wl_ppo_wide_num_params(689) == 2 + 2 * (64 * 689 + 64 + 64 * 64 + 64) + 3 * 64 + 3
    buf = (C.c_float * 4096)()
    base = C.addressof(buf)
    mlp = lambda d, o: A.WlMlp(base, base, base, base, base, base, d, o, 64, 1)
    a, c = mlp(689, 2), mlp(689, 1)
    st = A.WlPpoWideState(*([base] * 15), 689, 704, 1024, 512, 8)
    bt = A.WlPpoBatch(*([base] * 9))
    hp = A.WlPpoParams(0.2, 1.0, 0.005, 0.01, 1.0, 0.9, 0.999, 1e-8, 1e-5, 1e-2, 1, 1)
    call = lambda s, start, size, actor=a: lib.wl_ppo_wide_gradients(C.byref(actor), C.byref(c), base, C.byref(bt), start, size,
                                                                     C.byref(hp), C.byref(s), 0, None)
    assert call(st, 32, 512) == -1                 # minibatch starts are multiples of 64
    assert call(st, 0, 500) == -1                  # ... and sizes
    assert call(st, 768, 512) == -1                # runs past the staged rows
    assert call(st, 0, 1024) == -1                 # larger than the minibatch capacity
    wrong_dp = A.WlPpoWideState(*([base] * 15), 689, 768, 1024, 512, 8)
    assert call(wrong_dp, 0, 512) == -1            # dp is D rounded up to 64
    assert call(st, 0, 512, actor=mlp(600, 2)) == -1   # the nets' width must be the state's
    assert lib.wl_ppo_wide_stage(base, base, 1000, C.byref(st), None) == -1          # rows are staged 64 at a time
    sc = A.WlActScratch(base, base, base, 704, 5, 4096, 0)                            # needs ceil(704 / 128) = 6 partial-sum rows
    assert lib.wl_actor_critic_act_planes(C.byref(a), C.byref(c), base, 128, base, 689, base, base, base, base, 0, 1, 2, 0, 3,
                                          C.byref(sc), None) == -1
    io = A.WlCollectIo(base, base, base, base, base)
    out = A.WlStepOut(base, base, base, base, None)
    ep = PP.elev_params()
    big = A.WlEnvBuffers(state=base, episode_len=base, ref_poses=base, metrics=base, stride=65536, n_envs=65536, env_offset=0,
                         metrics_slots=1)
    assert lib.wl_elev_collect_step(C.byref(ep), C.byref(big), None, C.byref(a), C.byref(c), base, C.byref(io), C.byref(out), 0, 1, 2,
                                    None) == -1       # no heightfield (and beyond the quad form)


def test_host_rules_of_the_wide_learner_and_the_policy_step_forms():
    """pure host logic: split-K factor of the dW1 contraction and the policy-step form by size (tables in DESIGN.md section 6)"""
    from wheeledlab_amd.policy import ActorCritic as KernelAC
    from wheeledlab_amd.rl.ppo import FusedWidePpoStep as F
    assert F.pick_splits(704, 131072) == 128 and F.pick_splits(3264, 32768) == 16      # elevation / visual agents
    assert F.pick_splits(704, 6400) == 100 and F.pick_splits(64, 512) == 8             # never more splits than K chunks
    assert F.shapes_ok(524288, 131072) and not F.shapes_ok(1000, 500)
    form = KernelAC.planes_form
    assert form(4096, 689) is None and form(16384, 689) == "one" and form(1024, 3208) == "two" and form(4096, 3208) == "one"
    assert form(16384, 3208) is None and form(100000, 40) is None
