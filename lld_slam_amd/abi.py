"""ctypes mirror of include/lld_amd.h.

The struct layouts here are the single Python-side definition of the C ABI; the product library
(`lld_slam_amd/csrc/liblld_amd.so`, symbols ``lld_*``) and the test-only CPU oracle
(`oracle/liblld_oracle.so`, symbols ``lldo_*``) are both bound through :class:`Lib`.

Nothing in this module touches the oracle; see ``oracle/oracle_py.py`` for that loader (tests only).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

c_double_p = C.POINTER(C.c_double)
c_float_p = C.POINTER(C.c_float)
c_int32_p = C.POINTER(C.c_int32)
c_uint32_p = C.POINTER(C.c_uint32)
c_uint8_p = C.POINTER(C.c_uint8)

LLD_OK = 0
LLD_ERR_INVALID = -1
LLD_ERR_NO_DEVICE = -2
LLD_ERR_HIP = -3
LLD_ERR_ALLOC = -4
LLD_ERR_UNSUPPORTED = -5


class Camera(C.Structure):
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double), ("bf", C.c_double)]


class BAWindow(C.Structure):
    _fields_ = [
        ("cam", Camera),
        ("n_cams", C.c_int32), ("n_free_cams", C.c_int32),
        ("cam_qt", c_double_p),
        ("n_points", C.c_int32),
        ("pt_xyz", c_double_p),
        ("pt_obs_start", c_int32_p),
        ("n_pt_obs", C.c_int32),
        ("pt_obs_cam", c_int32_p),
        ("pt_obs_uvr", c_double_p),
        ("pt_obs_inv_sigma2", c_double_p),
        ("n_lines", C.c_int32),
        ("line_x0", c_double_p),
        ("line_dir", c_double_p),
        ("ln_obs_start", c_int32_p),
        ("n_ln_obs", C.c_int32),
        ("ln_obs_cam", c_int32_p),
        ("ln_obs_left", c_double_p),
        ("ln_obs_right", c_double_p),
        ("ln_obs_octave", c_int32_p),
    ]


class LineStereoParams(C.Structure):
    _fields_ = [("K", C.c_double * 9), ("b", C.c_double), ("tau", C.c_double), ("min_line_length", C.c_int32), ("is_stereo", C.c_int32)]


class LineTrackParams(C.Structure):
    _fields_ = [("K", C.c_double * 9), ("T_curr", C.c_double * 16), ("b", C.c_double), ("thr_reproj_base", C.c_double), ("md_thr", C.c_double),
                ("sx", C.c_double), ("sy", C.c_double), ("monocular", C.c_int32), ("use_grid", C.c_int32)]


class LineLastKfParams(C.Structure):
    _fields_ = [("K", C.c_double * 9), ("T_curr", C.c_double * 16), ("T_last", C.c_double * 16), ("b", C.c_double), ("thr_reproj_base", C.c_double),
                ("md_thr", C.c_double), ("sx", C.c_double), ("sy", C.c_double), ("use_grid", C.c_int32), ("pad", C.c_int32)]


class BAParams(C.Structure):
    _fields_ = [
        ("gamma", C.c_double),
        ("its_round1", C.c_int32), ("its_round2", C.c_int32), ("ln_filter", C.c_int32), ("max_trials", C.c_int32),
        ("pcg_rel_tol", C.c_double),
        ("pcg_max_iter", C.c_int32), ("reduced_solver", C.c_int32), ("protocol", C.c_int32), ("robust_points", C.c_int32),
        ("abort_after_trials", C.c_int32), ("deterministic", C.c_int32),
    ]


class BAStats(C.Structure):
    _fields_ = [
        ("chi2_round1", C.c_double), ("chi2_final", C.c_double),
        ("lm_iterations", C.c_int32 * 2), ("lm_trials", C.c_int32 * 2),
        ("pcg_iterations", C.c_int32), ("n_pt_obs_outlier", C.c_int32),
        ("n_ln_edge_outlier", C.c_int32), ("n_lines_removed", C.c_int32),
        ("aborted", C.c_int32), ("reserved", C.c_int32),
    ]


class BAResult(C.Structure):
    _fields_ = [
        ("cam_qt", c_double_p), ("pt_xyz", c_double_p), ("line_x0", c_double_p), ("line_dir", c_double_p),
        ("pt_obs_outlier", c_uint8_p), ("ln_edge_outlier", c_uint8_p), ("line_removed", c_uint8_p),
        ("stats", BAStats),
    ]


class PoseProblem(C.Structure):
    _fields_ = [
        ("cam", Camera),
        ("pose_qt", C.c_double * 7),
        ("n_points", C.c_int32),
        ("pt_xw", c_double_p), ("pt_uvr", c_double_p), ("pt_inv_sigma2", c_double_p),
        ("n_lines", C.c_int32),
        ("ln_x0", c_double_p), ("ln_dir", c_double_p), ("ln_left", c_double_p), ("ln_right", c_double_p),
        ("ln_octave", c_int32_p),
        ("ln_frame_index", c_int32_p),
    ]


class PoseParams(C.Structure):
    _fields_ = [("gamma", C.c_double), ("n_rounds", C.c_int32), ("its_per_round", C.c_int32),
                ("max_trials", C.c_int32), ("reserved", C.c_int32)]


class PoseResult(C.Structure):
    _fields_ = [
        ("pose_qt", C.c_double * 7),
        ("n_inliers", C.c_int32), ("lm_iterations", C.c_int32), ("lm_trials", C.c_int32), ("reserved", C.c_int32),
        ("chi2", C.c_double),
        ("pt_outlier", c_uint8_p), ("ln_outlier", c_uint8_p),
    ]


def _p(arr, ctype):
    return arr.ctypes.data_as(C.POINTER(ctype))


def as_f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


def as_i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


# Every symbol include/lld_amd.h declares (checked by tests/test_abi.py against the built library).
class Sim3Problem(C.Structure):
    _fields_ = [("fx1", C.c_double), ("fy1", C.c_double), ("cx1", C.c_double), ("cy1", C.c_double),
                ("fx2", C.c_double), ("fy2", C.c_double), ("cx2", C.c_double), ("cy2", C.c_double),
                ("s12_q", C.c_double * 4), ("s12_t", C.c_double * 3), ("s12_s", C.c_double), ("n", C.c_int32), ("reserved", C.c_int32),
                ("p1c", c_double_p), ("p2c", c_double_p), ("obs1", c_double_p), ("obs2", c_double_p), ("inv_sigma2_1", c_double_p),
                ("inv_sigma2_2", c_double_p)]


class Sim3Params(C.Structure):
    _fields_ = [("th2", C.c_double), ("fix_scale", C.c_int32), ("its_first", C.c_int32), ("its_more_bad", C.c_int32),
                ("its_more_clean", C.c_int32), ("min_inliers", C.c_int32), ("max_trials", C.c_int32)]


class Sim3Result(C.Structure):
    _fields_ = [("s12_q", C.c_double * 4), ("s12_t", C.c_double * 3), ("s12_s", C.c_double), ("dropped", c_uint8_p), ("n_inliers", C.c_int32),
                ("n_bad_first", C.c_int32), ("lm_iterations", C.c_int32 * 2), ("lm_trials", C.c_int32 * 2), ("chi2", C.c_double)]


class PoseGraph(C.Structure):
    _fields_ = [("n_vertices", C.c_int32), ("n_edges", C.c_int32), ("sim3", c_double_p), ("fixed", c_uint8_p), ("edge_i", c_int32_p),
                ("edge_j", c_int32_p), ("edge_sji", c_double_p)]


class PoseGraphParams(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("fix_scale", C.c_int32), ("lambda_init", C.c_double), ("max_trials", C.c_int32),
                ("pcg_max_iter", C.c_int32), ("pcg_rel_tol", C.c_double), ("solver", C.c_int32), ("reserved", C.c_int32)]


class PoseGraphResult(C.Structure):
    _fields_ = [("sim3", c_double_p), ("chi2", C.c_double), ("lm_iterations", C.c_int32), ("lm_trials", C.c_int32), ("pcg_iterations", C.c_int32),
                ("solver_used", C.c_int32)]


PRODUCT_SYMBOLS = [
    "lld_status_string", "lld_ctx_create", "lld_ctx_destroy", "lld_ctx_stream", "lld_ctx_synchronize", "lld_ctx_release_cache",
    "lld_se3_from_tcw_f32", "lld_se3_to_tcw_f32", "lld_orb_inv_level_sigma2",
    "lld_ba_params_default", "lld_local_ba",
    "lld_local_ba_stopflag", "lld_ba_batch_create", "lld_ba_batch_solve", "lld_ba_batch_download", "lld_ba_batch_download_range", "lld_ba_batch_stats",
    "lld_ba_batch_result_records", "lld_ba_batch_set_phase_timing", "lld_ba_batch_phase_ms", "lld_ba_batch_kernel_stats", "lld_ba_batch_set_groups",
    "lld_ba_batch_destroy", "lld_ba_chol_plan",
    "lld_device_count", "lld_ba_multi_shard", "lld_ba_multi_create", "lld_ba_multi_solve", "lld_ba_multi_result_records", "lld_ba_multi_verify_gathered",
    "lld_ba_multi_download", "lld_ba_multi_times_ms", "lld_ba_multi_destroy",
    "lld_pose_params_default", "lld_pose_opt",
    "lld_pose_batch_create", "lld_pose_batch_solve", "lld_pose_batch_download", "lld_pose_batch_destroy",
    "lld_match_hamming256", "lld_match_hamming256_csr", "lld_match_hamming256_batch_dev",
    "lld_match_l2f32", "lld_match_l2f32_batch_dev", "lld_line_match_greedy", "lld_line_match_stereo",
    "lld_line_track_match", "lld_line_hough_cells", "lld_line_match_last_frame",
    "lld_orb_search_run", "lld_orb_search_batch", "lld_orb_search_local_points", "lld_orb_search_last_frame", "lld_orb_fuse_search", "lld_orb_search_projected", "lld_orb_search_by_sim3",
    "lld_compute_stereo_matches",
    "lld_frame_create", "lld_frame_search_last_frame", "lld_frame_search_local_points", "lld_frame_destroy",
    "lld_frame_set_lines", "lld_track_params_default", "lld_frame_track_motion_model", "lld_frame_track_local_map", "lld_frame_track_download", "lld_frame_track_set_state",
    "lld_sim3_params_default", "lld_optimize_sim3", "lld_optimize_sim3_batch",
    "lld_pose_graph_params_default", "lld_optimize_essential_graph",
]


# see bench.py: more hardware queues than the default 4, so that the library's group streams do not share a queue with the caller's
# other streams (effective only if HIP has not been initialised yet)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")


def product_library_path() -> str:
    if os.environ.get("LLD_AMD_LIB"):          # kernel experiments: an alternative build of the same ABI
        return os.environ["LLD_AMD_LIB"]
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "liblld_amd.so")


class Lib:
    """Thin typed view over one shared library exporting the ABI with a given symbol prefix."""

    def __init__(self, path: str, prefix: str):
        if not os.path.exists(path):
            raise FileNotFoundError(
                f"{path} is missing: build it first (python -c 'import __graft_entry__ as g; g.build()'). "
                "There is no fallback path.")
        self.path = path
        self.prefix = prefix
        self.dll = C.CDLL(path)
        self._bind()

    def fn(self, name):
        return getattr(self.dll, self.prefix + name)

    def has(self, name) -> bool:
        try:
            self.fn(name)
            return True
        except AttributeError:
            return False

    def _bind(self):
        vp = C.c_void_p
        f = self.fn
        f("se3_from_tcw_f32").argtypes = [c_float_p, c_double_p]; f("se3_from_tcw_f32").restype = None
        f("se3_to_tcw_f32").argtypes = [c_double_p, c_float_p]; f("se3_to_tcw_f32").restype = None
        f("orb_inv_level_sigma2").argtypes = [C.c_float, C.c_int, c_float_p]; f("orb_inv_level_sigma2").restype = None
        f("ba_params_default").argtypes = [C.POINTER(BAParams)]; f("ba_params_default").restype = None
        f("pose_params_default").argtypes = [C.POINTER(PoseParams)]; f("pose_params_default").restype = None
        f("local_ba").argtypes = [vp, C.POINTER(BAWindow), C.POINTER(BAParams), C.POINTER(C.c_int), C.POINTER(BAResult)]
        f("local_ba").restype = C.c_int
        f("pose_opt").argtypes = [vp, C.POINTER(PoseProblem), C.POINTER(PoseParams), C.POINTER(PoseResult)]
        f("pose_opt").restype = C.c_int
        f("match_hamming256").argtypes = [vp, c_uint32_p, C.c_int, c_uint32_p, C.c_int, c_uint8_p,
                                          c_int32_p, c_int32_p, c_int32_p, c_int32_p]
        f("match_hamming256").restype = C.c_int
        f("match_hamming256_csr").argtypes = [vp, c_uint32_p, C.c_int, c_uint32_p, C.c_int, c_int32_p, c_int32_p,
                                              c_int32_p, c_int32_p, c_int32_p, c_int32_p]
        f("match_hamming256_csr").restype = C.c_int
        f("match_l2f32").argtypes = [vp, c_float_p, C.c_int, c_float_p, C.c_int, C.c_int, c_uint8_p,
                                     c_int32_p, c_double_p, c_int32_p, c_double_p]
        f("match_l2f32").restype = C.c_int
        f("line_match_greedy").argtypes = [vp, c_float_p, C.c_int, c_float_p, C.c_int, C.c_int, c_uint8_p, C.c_double,
                                           c_int32_p, c_double_p]
        f("line_match_greedy").restype = C.c_int
        f("line_match_stereo").argtypes = [vp, C.POINTER(LineStereoParams), c_float_p, c_int32_p, c_float_p, C.c_int, c_float_p, c_int32_p,
                                           c_float_p, C.c_int, C.c_int, c_int32_p, c_double_p, c_uint8_p]
        f("line_match_stereo").restype = C.c_int
        f("line_track_match").argtypes = [vp, C.POINTER(LineTrackParams), C.c_int, c_double_p, c_double_p, c_double_p, c_double_p, c_uint8_p, c_float_p,
                                          C.c_int, c_float_p, c_int32_p, C.c_int, c_float_p, c_int32_p, c_uint8_p, c_float_p, C.c_int,
                                          c_int32_p, c_double_p, c_uint8_p]
        f("line_track_match").restype = C.c_int
        f("line_hough_cells").argtypes = [c_float_p, C.c_int, C.c_double, C.c_double, c_int32_p]
        f("line_hough_cells").restype = C.c_int
        f("line_match_last_frame").argtypes = [vp, C.POINTER(LineLastKfParams), C.c_int, c_float_p, C.c_int, c_float_p, c_int32_p, c_uint8_p, c_float_p,
                                               C.c_int, c_float_p, c_int32_p, C.c_int, c_float_p, c_int32_p, c_uint8_p, c_float_p, C.c_int,
                                               c_int32_p, c_uint8_p, c_double_p, c_double_p]
        f("line_match_last_frame").restype = C.c_int
        if self.prefix == "lld_":
            f("status_string").argtypes = [C.c_int]; f("status_string").restype = C.c_char_p
            f("ctx_create").argtypes = [C.c_int, C.POINTER(vp)]; f("ctx_create").restype = C.c_int
            f("ctx_destroy").argtypes = [vp]; f("ctx_destroy").restype = None
            f("ctx_stream").argtypes = [vp]; f("ctx_stream").restype = vp
            f("ctx_synchronize").argtypes = [vp]; f("ctx_synchronize").restype = C.c_int
            f("ctx_release_cache").argtypes = [vp]; f("ctx_release_cache").restype = C.c_int
            f("ba_batch_create").argtypes = [vp, C.c_int, C.POINTER(BAWindow), C.POINTER(BAParams), C.POINTER(vp)]
            f("ba_batch_create").restype = C.c_int
            f("ba_batch_solve").argtypes = [vp, C.POINTER(C.c_int)]; f("ba_batch_solve").restype = C.c_int
            f("ba_batch_download").argtypes = [vp, C.c_int, C.POINTER(BAResult)]; f("ba_batch_download").restype = C.c_int
            f("ba_batch_download_range").argtypes = [vp, C.c_int, C.c_int, C.POINTER(BAResult)]; f("ba_batch_download_range").restype = C.c_int
            f("ba_batch_stats").argtypes = [vp, C.POINTER(BAStats)]; f("ba_batch_stats").restype = C.c_int
            f("ba_batch_result_records").argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_uint64)]
            f("ba_batch_result_records").restype = C.c_int
            f("ba_batch_phase_ms").argtypes = [vp, c_double_p]; f("ba_batch_phase_ms").restype = C.c_int
            f("ba_batch_set_phase_timing").argtypes = [vp, C.c_int]; f("ba_batch_set_phase_timing").restype = C.c_int
            f("ba_batch_kernel_stats").argtypes = [vp, C.c_int, C.POINTER(C.c_int64), c_double_p]
            f("ba_batch_kernel_stats").restype = C.c_int
            f("ba_batch_set_groups").argtypes = [vp, C.c_int]; f("ba_batch_set_groups").restype = C.c_int
            f("ba_batch_destroy").argtypes = [vp]; f("ba_batch_destroy").restype = None
            f("pose_batch_create").argtypes = [vp, C.c_int, C.POINTER(PoseProblem), C.POINTER(PoseParams), C.POINTER(vp)]
            f("pose_batch_create").restype = C.c_int
            f("pose_batch_solve").argtypes = [vp]; f("pose_batch_solve").restype = C.c_int
            f("pose_batch_download").argtypes = [vp, C.c_int, C.POINTER(PoseResult)]; f("pose_batch_download").restype = C.c_int
            f("pose_batch_destroy").argtypes = [vp]; f("pose_batch_destroy").restype = None
            f("match_hamming256_batch_dev").argtypes = [vp, C.c_int, vp, C.c_int, vp, C.c_int, vp, vp, vp, vp]
            f("match_hamming256_batch_dev").restype = C.c_int
            f("match_l2f32_batch_dev").argtypes = [vp, C.c_int, vp, C.c_int, vp, C.c_int, C.c_int, vp, vp, vp, vp]
            f("match_l2f32_batch_dev").restype = C.c_int


_PRODUCT = None


def product() -> Lib:
    """The HIP library.  Raises if it has not been built — the product path never falls back to CPU."""
    global _PRODUCT
    if _PRODUCT is None:
        # PyTorch wheels bundle their own libamdhip64.so (soname libamdhip64.so.7).  Two HIP runtimes in one process
        # cannot both own the GPU, so when torch is importable it is loaded first: liblld_amd.so's DT_NEEDED
        # libamdhip64.so.7 then resolves to the runtime torch already mapped.  C/C++ hosts simply link /opt/rocm's.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        _PRODUCT = Lib(product_library_path(), "lld_")
    return _PRODUCT
