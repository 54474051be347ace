"""ctypes loader for libbh.so (the C-ABI of include/bh.h).

There is no fallback of any kind: if the HIP library is missing or fails to load,
importing this module raises, and every compute entry point needs a HIP device.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# BH_LIB_PATH: A/B measurement of an alternative build (tools/) without overwriting the product library
LIB_PATH = os.environ.get("BH_LIB_PATH") or os.path.join(_HERE, "libbh.so")


class BhParams(C.Structure):
    """struct bh_params (include/bh.h) — defaults are the reference's #defines
    (nbody_v5_bench.cu:14-18)."""
    _fields_ = [
        ("G", C.c_float), ("theta", C.c_float), ("dt", C.c_float), ("eps2", C.c_float),
        ("max_speed", C.c_float),
        ("leaf_cap", C.c_int32), ("max_depth", C.c_int32), ("key_bits", C.c_int32),
        ("strict_fp", C.c_int32), ("force_variant", C.c_int32), ("xcd_mode", C.c_int32),
        ("sort_variant", C.c_int32), ("literal_force", C.c_int32), ("force_block", C.c_int32),
        ("step_graph", C.c_int32), ("force_group", C.c_int32), ("key_curve", C.c_int32),
        ("force_coop", C.c_int32),
    ]


class BhNode(C.Structure):
    """struct bh_node: one 32-byte octree record."""
    _fields_ = [
        ("x", C.c_float), ("y", C.c_float), ("z", C.c_float), ("m", C.c_float),
        ("s", C.c_float), ("first", C.c_int32), ("count", C.c_int32), ("kind", C.c_int32),
    ]


class BhStats(C.Structure):
    _fields_ = [
        ("n", C.c_int32), ("n_internal", C.c_int32), ("n_entries", C.c_int32),
        ("max_level", C.c_int32), ("status_flags", C.c_int32), ("steps", C.c_int32),
        ("ms_bbox", C.c_float), ("ms_morton", C.c_float), ("ms_sort", C.c_float),
        ("ms_build", C.c_float), ("ms_com", C.c_float), ("ms_force", C.c_float),
        ("ms_integrate", C.c_float), ("ms_step", C.c_float),
        ("count_V", C.c_uint64), ("count_O", C.c_uint64), ("count_P", C.c_uint64),
        ("force_redo_waves", C.c_int32), ("sort_slow_buckets", C.c_int32), ("reserved", C.c_int32 * 6),
    ]


class BhWalkStats(C.Structure):
    """struct bh_walk_stats: event counters of one launch of the default force walk (measurement)."""
    _fields_ = [("waves", C.c_uint64), ("pairs", C.c_uint64), ("blocks", C.c_uint64), ("masked_pairs", C.c_uint64),
                ("clock_ghz", C.c_double), ("wave_cycles_max", C.c_double), ("wave_cycles_mean", C.c_double),
                ("lane_spills", C.c_uint64), ("no_taker_pairs", C.c_uint64), ("fetch_wait_cycles", C.c_uint64),
                ("reserved", C.c_uint64 * 1)]


class BhDdSizes(C.Structure):
    """struct bh_dd_sizes: buffer sizes of the domain-decomposed stepping."""
    _fields_ = [(k, C.c_int64) for k in
                ("x1_bytes", "x2_bytes", "x3_bytes", "pool_records", "seg_base", "let_min", "let_cap", "top_base")]


_P = C.c_void_p
_F = C.POINTER(C.c_float)
BH_DD_PIECE_CAP = 512
BH_ERR_COMM, BH_ERR_DOMAIN_LEFT = -9, -10

# ---- the multi-GPU step behind the ABI (bh_comm / bh_rank / bh_group)
COMM_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
COMM_RELEASE_FN = C.CFUNCTYPE(None, C.c_void_p)
COMM_V_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64),
                        C.POINTER(C.c_int64), C.c_void_p)   # all_to_all_v (ABI 6; may be NULL)


class BhComm(C.Structure):
    """struct bh_comm: how a rank's buffers travel (all_gather / all_to_all on a HIP stream)."""
    _fields_ = [("world", C.c_int32), ("rank", C.c_int32), ("user", C.c_void_p),
                ("all_gather", COMM_FN), ("all_to_all", COMM_FN), ("release", COMM_RELEASE_FN),
                ("all_to_all_v", COMM_V_FN)]


class BhRankOpts(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("n_cap", "mig_cap", "let_cap", "let_mode", "split", "log", "serial",
                                         "split_pct")] + \
               [("reserved", C.c_int32 * 8)]


class BhRankPlan(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("n_cap", "mig_cap", "let_cap", "stride0")] + \
               [("sz", BhDdSizes), ("bytes", C.c_int64 * 8)]


class BhRankBuffers(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("x1s", "x1r", "x2s", "x2r", "x3s", "x3r", "x4s", "pool")]


class BhRankInfo(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("n_loc", "stride", "mig_stride", "mig_last", "mig_rounds", "let_retries",
                                         "left_rank", "left_status")] + \
               [("steps", C.c_int64), ("let_counts", C.c_int32 * 64), ("split_now", C.c_int32), ("x4_us", C.c_int32),
                ("x4_recv_kb", C.c_int32), ("reserved", C.c_int32 * 5)]


_PI = C.POINTER(C.c_int)
SCRIPT_FNS = [
    ("cube_pack", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)),
    ("phase_migrate", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int)),
    ("migrate_pack", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int)),
    ("phase_tree", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, _PI, _PI, _PI)),
    ("phase_let", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int)),
    ("phase_force", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int32), _PI)),
    ("phase_end", C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)),
]


class BhRankScript(C.Structure):
    """struct bh_rank_script: a scripted engine behind the step protocol (CPU protocol tests)."""
    _fields_ = [("user", C.c_void_p)] + SCRIPT_FNS

# every symbol include/bh.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("bh_abi_version", C.c_int, []),
    ("bh_default_params", C.c_int, [C.POINTER(BhParams)]),
    ("bh_create", C.c_int, [C.POINTER(_P), C.c_int, C.POINTER(BhParams), C.c_int]),
    ("bh_create_on_stream", C.c_int, [C.POINTER(_P), C.c_int, C.POINTER(BhParams), C.c_int, _P]),
    ("bh_destroy", None, [_P]),
    ("bh_strerror", C.c_char_p, [C.c_int]),
    ("bh_last_hip_error", C.c_int, [_P]),
    ("bh_upload", C.c_int, [_P] + [_F] * 7),
    ("bh_step", C.c_int, [_P]),
    ("bh_bbox", C.c_int, [_P]),
    ("bh_morton", C.c_int, [_P]),
    ("bh_sort", C.c_int, [_P]),
    ("bh_build", C.c_int, [_P]),
    ("bh_com", C.c_int, [_P]),
    ("bh_force", C.c_int, [_P]),
    ("bh_integrate", C.c_int, [_P]),
    ("bh_force_range", C.c_int, [_P, C.c_int, C.c_int]),
    ("bh_force_count", C.c_int, [_P]),
    ("bh_force_walk_stats", C.c_int, [_P, C.POINTER(BhWalkStats)]),
    ("bh_force_launch_trace", C.c_int, [_P, C.POINTER(C.c_uint32), C.c_int, C.POINTER(C.c_int)]),
    ("bh_download", C.c_int, [_P] + [_F] * 6),
    ("bh_download_acc", C.c_int, [_P] + [_F] * 3),
    ("bh_download_bounds", C.c_int, [_P, _F]),
    ("bh_download_keys", C.c_int, [_P, C.POINTER(C.c_uint64)]),
    ("bh_download_order", C.c_int, [_P, C.POINTER(C.c_int32)]),
    ("bh_download_sorted_bodies", C.c_int, [_P, _F]),
    ("bh_download_tree", C.c_int, [_P, C.POINTER(BhNode), C.c_int, C.POINTER(C.c_int)]),
    ("bh_download_counters", C.c_int, [_P] + [C.POINTER(C.c_uint32)] * 3),
    ("bh_download_mass", C.c_int, [_P, _F]),
    ("bh_export_visual", C.c_int, [_P, _F, _F]),
    ("bh_write_text", C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_float, C.c_float] + [_F] * 6),
    ("bh_read_text", C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)] + [_F] * 6),
    ("bh_write_snapshot", C.c_int, [C.c_char_p, C.c_int, C.c_int, C.POINTER(BhParams)] + [_F] * 7),
    ("bh_read_snapshot", C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                   C.POINTER(BhParams)] + [_F] * 7),
    ("bh_get_stats", C.c_int, [_P, C.POINTER(BhStats)]),
    ("bh_set_timing", C.c_int, [_P, C.c_int]),
    ("bh_sync", C.c_int, [_P]),
    ("bh_device_acc", C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_int64)]),
    ("bh_bind_acc", C.c_int, [_P, _P]),
    ("bh_timing_history", C.c_int, [_P, _F, _F, C.c_int, C.POINTER(C.c_int)]),
    ("bh_n", C.c_int, [_P]),
    ("bh_dd_query", C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(BhDdSizes)]),
    ("bh_dd_init", C.c_int, [_P, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, _P, C.c_int64]),
    ("bh_dd_upload", C.c_int, [_P, C.c_int] + [_F] * 7 + [C.POINTER(C.c_int32)]),
    ("bh_dd_cube_pack", C.c_int, [_P, _P]),
    ("bh_dd_cube_apply", C.c_int, [_P, _P]),
    ("bh_dd_migrate_pack", C.c_int, [_P, _P, C.c_int]),
    ("bh_dd_migrate_apply", C.c_int, [_P, _P, C.c_int] + [C.POINTER(C.c_int)] * 3),
    ("bh_dd_tree", C.c_int, [_P, _P]),
    ("bh_dd_let_pack", C.c_int, [_P, _P, _P, C.c_int]),
    ("bh_dd_force_local", C.c_int, [_P, _P]),
    ("bh_dd_top", C.c_int, [_P, _P, C.c_int]),
    ("bh_dd_force", C.c_int, [_P]),
    ("bh_dd_let_check", C.c_int, [_P, C.c_int, C.POINTER(C.c_int32)]),
    ("bh_dd_set_let_mode", C.c_int, [_P, C.c_int]),
    ("bh_dd_set_serial", C.c_int, [_P, C.c_int]),
    ("bh_dd_set_split_percent", C.c_int, [_P, C.c_int]),
    ("bh_dd_phase_migrate", C.c_int, [_P, _P, _P, C.c_int]),
    ("bh_dd_phase_tree", C.c_int, [_P, _P, C.c_int, _P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("bh_dd_phase_let", C.c_int, [_P, _P, _P, C.c_int, C.c_int]),
    ("bh_dd_phase_force", C.c_int, [_P, _P, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int)]),
    ("bh_dd_phase_end", C.c_int, [_P, _P]),
    ("bh_dd_download", C.c_int, [_P, _F, _F, _F]),
    ("bh_dd_get_info", C.c_int, [_P, C.POINTER(C.c_int32)]),
    ("bh_dd_walk_stats", C.c_int, [_P, C.c_int, C.POINTER(BhWalkStats)]),
    ("bh_dd_pass_times", C.c_int, [_P, _F]),
    ("bh_dd_needs_matrix", C.c_int, [_P, C.POINTER(C.c_int32)]),
    ("bh_comm_rccl_from", C.c_int, [C.POINTER(BhComm), _P, C.c_int, C.c_int]),
    ("bh_comm_rccl_unique_id", C.c_int, [_P]),
    ("bh_comm_rccl_init_rank", C.c_int, [C.POINTER(BhComm), _P, C.c_int, C.c_int, C.c_int]),
    ("bh_comm_check", C.c_int, [C.POINTER(BhComm)]),
    ("bh_hub_create", C.c_int, [C.POINTER(_P), C.c_int]),
    ("bh_comm_hub", C.c_int, [C.POINTER(BhComm), _P, C.c_int]),
    ("bh_hub_abort", None, [_P]),
    ("bh_hub_destroy", None, [_P]),
    ("bh_rank_default_opts", C.c_int, [C.POINTER(BhRankOpts)]),
    ("bh_rank_query", C.c_int, [C.c_int64, C.c_int, C.POINTER(BhRankOpts), C.POINTER(BhRankPlan)]),
    ("bh_rank_create", C.c_int, [C.POINTER(_P), C.POINTER(BhComm), C.c_int64, C.POINTER(BhParams),
                                 C.POINTER(BhRankOpts), C.c_int, _P, C.POINTER(BhRankBuffers)]),
    ("bh_rank_create_scripted", C.c_int, [C.POINTER(_P), C.POINTER(BhComm), C.POINTER(BhRankScript),
                                          C.POINTER(BhRankPlan), C.POINTER(BhRankOpts)]),
    ("bh_rank_upload", C.c_int, [_P, C.c_int] + [_F] * 7 + [C.POINTER(C.c_int32)]),
    ("bh_rank_step", C.c_int, [_P, C.c_int]),
    ("bh_rank_get_info", C.c_int, [_P, C.POINTER(BhRankInfo)]),
    ("bh_rank_ctx", _P, [_P]),
    ("bh_rank_buffers_of", C.c_int, [_P, C.POINTER(BhRankBuffers), C.POINTER(BhRankPlan)]),
    ("bh_rank_set_profile", C.c_int, [_P, C.c_int]),
    ("bh_rank_phase_ms", C.c_int, [_P, C.POINTER(C.c_double), _PI]),
    ("bh_rank_read_log", C.c_int, [_P, C.POINTER(C.c_int32), C.c_int, _PI]),
    ("bh_rank_replay_force_phase", C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    ("bh_rank_destroy", None, [_P]),
    ("bh_create_group", C.c_int, [C.POINTER(_P), C.c_int, _PI, C.c_int64, C.POINTER(BhParams),
                                  C.POINTER(BhRankOpts), C.c_int]),
    ("bh_group_upload", C.c_int, [_P] + [_F] * 7),
    ("bh_step_group", C.c_int, [_P, C.c_int]),
    ("bh_group_sync", C.c_int, [_P]),
    ("bh_group_download", C.c_int, [_P] + [_F] * 6),
    ("bh_group_download_acc", C.c_int, [_P] + [_F] * 3),
    ("bh_group_size", C.c_int, [_P]),
    ("bh_group_rank", _P, [_P, C.c_int]),
    ("bh_destroy_group", None, [_P]),
    ("bh_ic_plummer", C.c_int, [C.c_int, C.c_uint64, C.c_float, C.c_float] + [_F] * 7),
    ("bh_ic_disc", C.c_int, [C.c_int, C.c_uint64, C.c_float] + [_F] * 7),
    ("bh_ic_disc_msvc", C.c_int, [C.c_int, C.c_uint32, C.c_float] + [_F] * 7),
]


def load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C nbody-barnes-hut-cuda_amd). "
            "There is no CPU fallback.")
    # PyTorch bundles its own libamdhip64; if libbh.so drags /opt/rocm's copy into the process first,
    # torch later finds "No HIP GPUs".  Loading torch first makes both share one HIP runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    return lib


lib = load()
