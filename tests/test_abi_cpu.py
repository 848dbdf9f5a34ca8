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
