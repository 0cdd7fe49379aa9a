"""The C-ABI library loads on a CPU-only box and exports every symbol include/lld_amd.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

from lld_slam_amd import abi, orb_search, tracking

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "lld_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lld_[a-z0-9_]+)\s*\(", src)))


def test_header_and_python_symbol_lists_agree():
    assert _header_functions() == sorted(abi.PRODUCT_SYMBOLS)


def test_library_exports_every_declared_symbol():
    path = abi.product_library_path()
    assert os.path.exists(path), "build the HIP library first: python -c 'import __graft_entry__ as g; g.build()'"
    dll = ctypes.CDLL(path)
    missing = [s for s in _header_functions() if not hasattr(dll, s)]
    assert not missing, f"symbols declared in include/lld_amd.h but not exported: {missing}"


def test_struct_layouts_match_the_header(tmp_path):
    """Compile a tiny C program against include/lld_amd.h and compare sizeof() with the ctypes mirrors."""
    import subprocess
    names = [("lld_camera", abi.Camera), ("lld_ba_window", abi.BAWindow), ("lld_ba_params", abi.BAParams),
             ("lld_ba_stats", abi.BAStats), ("lld_ba_result", abi.BAResult), ("lld_pose_problem", abi.PoseProblem),
             ("lld_pose_params", abi.PoseParams), ("lld_pose_result", abi.PoseResult),
             ("lld_line_stereo_params", abi.LineStereoParams), ("lld_orb_search", orb_search.OrbSearch), ("lld_orb_search_result", orb_search.OrbSearchResult),
             ("lld_frame_view", orb_search.FrameView), ("lld_map_points", orb_search.MapPoints), ("lld_frustum_result", orb_search.FrustumResult),
             ("lld_last_frame_points", orb_search.LastFramePoints), ("lld_keypoints", orb_search.Keypoints),
             ("lld_stereo_pyramids", orb_search.StereoPyramids), ("lld_stereo_result", orb_search.StereoResult),
             ("lld_sim3_problem", abi.Sim3Problem), ("lld_sim3_params", abi.Sim3Params), ("lld_sim3_result", abi.Sim3Result),
             ("lld_pose_graph", abi.PoseGraph), ("lld_pose_graph_params", abi.PoseGraphParams), ("lld_pose_graph_result", abi.PoseGraphResult),
             ("lld_frame_lines", tracking.FrameLines), ("lld_map_lines", tracking.MapLines), ("lld_track_params", tracking.TrackParams),
             ("lld_track_result", tracking.TrackResult), ("lld_frame_held", tracking.FrameHeld)]
    src = tmp_path / "sz.c"
    body = "".join(f'printf("%zu\\n", sizeof({n}));' for n, _ in names)
    # field offsets of the widest struct too: equal sizes alone would not catch two swapped members
    probes = ["t_occupied", "q_valid", "q_epiline", "cand_range", "n_cand", "grid_min_x", "n_levels", "disp_min", "only_stereo",
              "candidates", "nnratio", "check_orientation"]
    body += "".join(f'printf("%zu\\n", offsetof(lld_orb_search, {f}));' for f in probes)
    tprobes = [("lld_track_params", tracking.TrackParams, f) for f in ("pose", "th_motion", "direction", "line_thr_reproj_base", "line_use_grid")] + \
              [("lld_track_result", tracking.TrackResult, f) for f in ("chi2", "n_search_first", "n_discarded", "kp_point_id", "ln_outlier")] + \
              [("lld_frame_lines", tracking.FrameLines, f) for f in ("right_octave", "line_matches", "dim", "sx")] + \
              [("lld_frame_held", tracking.FrameHeld, f) for f in ("kp_outlier", "seen_point_id", "ln_dir", "n_tracked", "tracked_line_id")]
    body += "".join(f'printf("%zu\\n", offsetof({n}, {f}));' for n, _, f in tprobes)
    src.write_text(f'#include <stdio.h>\n#include <stddef.h>\n#include "{ROOT}/include/lld_amd.h"\nint main(void){{{body}return 0;}}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", str(src), "-o", str(exe)])   # the header is plain C
    sizes = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert sizes == [ctypes.sizeof(c) for _, c in names] + [getattr(orb_search.OrbSearch, f).offset for f in probes] + [getattr(c, f).offset for _, c, f in tprobes]


def test_host_helpers_agree_with_oracle_without_a_gpu(oracle):
    """The float<->double conversions of the ABI are host code; they must match the oracle bit for bit."""
    import numpy as np
    from lld_slam_amd import host
    lib = abi.product()
    rng = np.random.default_rng(0)
    for _ in range(50):
        T = np.eye(4, dtype=np.float32)
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        T[:3, :3] = oracle.quat_to_R(q).astype(np.float32); T[:3, 3] = rng.normal(0, 10, 3).astype(np.float32)
        a = host.se3_from_tcw_f32(lib, T); b = host.se3_from_tcw_f32(oracle.lib(), T)
        np.testing.assert_allclose(a, b, rtol=0, atol=4e-16)
        np.testing.assert_array_equal(host.se3_to_tcw_f32(lib, b), host.se3_to_tcw_f32(oracle.lib(), b))
    np.testing.assert_array_equal(host.orb_inv_level_sigma2(lib), host.orb_inv_level_sigma2(oracle.lib()))


def test_context_creation_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from lld_slam_amd import Context
    with pytest.raises(RuntimeError, match="no HIP device"):
        Context(0)
