"""ctypes binding of the C ABI in include/bprx.h (libbprx.so, built in-tree by build.py).

There is NO fallback: if the HIP library is missing or fails to load, importing the engine raises.
PyTorch is used by callers only for device memory, streams and torch.distributed; nothing here takes a
torch type -- pointers are plain integers (tensor.data_ptr()).
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BPRX_LIB") or os.path.join(HERE, "libbprx.so")    # BPRX_LIB: A/B builds (scripts/)
ABI_VERSION = 6
FLAG_EXPORT_USER_GRAD = 1
FLAG_EXPORT_ITEM_GRAD = 2
FLAG_DENSE_ALLREDUCE = 4
FLAG_ADAM_SWEEP, FLAG_ADAM_LAZY = 8, 16

MODEL = {"bprmf": 0, "vbpr": 1}
OPTIMIZER = {"sgd": 0, "adam_tf23": 1}
FEAT_DTYPE = {"fp32": 0, "bf16": 1, "fp8": 2}
E_RANGE = -4
PHASES = ["cast_Et", "proj_fwd", "triplet_grad", "proj_bwd", "reduce_parts", "apply", "dense_update", "loss_reduce", "item_seg", "seg_alloc", "row_count", "adam_catchup"]


class BprxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libbprx error %d: %s" % (code, msg))
        self.code = code


class Config(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("model", C.c_int32), ("num_users", C.c_int32), ("num_items", C.c_int32),
                ("embed_k", C.c_int32), ("embed_d", C.c_int32), ("feat_dim", C.c_int32), ("feat_dtype", C.c_int32),
                ("optimizer", C.c_int32), ("device", C.c_int32), ("max_batch", C.c_int64),
                ("lr", C.c_float), ("reg", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float),
                ("epsilon", C.c_float), ("flags", C.c_int32), ("feat_scale", C.c_float)]


TABLE_FIELDS = ["Gu", "Gi", "Bi", "Tu", "F", "E", "Bp", "m_Gu", "v_Gu", "m_Gi", "v_Gi", "m_Bi", "v_Bi",
                "m_Tu", "v_Tu", "m_E", "v_E", "m_Bp", "v_Bp"]


class Tables(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in TABLE_FIELDS]


_lib = None


def lib():
    """Load libbprx.so; raise loudly when it is absent (the product path has no CPU fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own HIP runtime (torch/lib/libamdhip64.so); it must be the first one loaded so that
    # libbprx.so's DT_NEEDED entry binds to the SAME runtime instance.  Two HIP runtimes in one process
    # leave the second one without devices ("no ROCm-capable device is detected").
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise ImportError("libbprx.so not found at %s: build it with `python -m fashionvisualexpl_recommend_amd.build` "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, u32, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_float
    sig = {
        "bprx_abi_version": (C.c_int, []),
        "bprx_create": (C.c_int, [C.POINTER(Config), C.POINTER(vp)]),
        "bprx_destroy": (C.c_int, [vp]),
        "bprx_last_error": (C.c_char_p, [vp]),
        "bprx_bind_tables": (C.c_int, [vp, C.POINTER(Tables)]),
        "bprx_set_hyper": (C.c_int, [vp, f32, f32]),
        "bprx_tables_dirty": (C.c_int, [vp, vp]),
        "bprx_set_adam_step": (C.c_int, [vp, i64, vp]),
        "bprx_get_adam_step": (i64, [vp]),
        "bprx_adam_is_lazy": (C.c_int, [vp]),
        "bprx_sync_adam": (C.c_int, [vp, vp]),
        "bprx_score_pairs": (C.c_int, [vp, vp, vp, i64, vp, vp]),
        "bprx_step": (C.c_int, [vp, vp, vp, vp, i64, vp, vp]),
        "bprx_step_begin": (C.c_int, [vp, vp, vp, vp, i64, vp]),
        "bprx_step_begin_sparse": (C.c_int, [vp, vp, vp, vp, i64, vp]),
        "bprx_step_begin_dense": (C.c_int, [vp, vp]),
        "bprx_dense_grad": (C.c_int, [vp, C.POINTER(vp), C.POINTER(i64)]),
        "bprx_step_end": (C.c_int, [vp, vp, vp]),
        "bprx_step_project": (C.c_int, [vp, vp]),
        "bprx_user_grad": (C.c_int, [vp, C.POINTER(vp), C.POINTER(vp)]),
        "bprx_clear_user_grad": (C.c_int, [vp, i64, i32, vp]),
        "bprx_item_grad": (C.c_int, [vp, C.POINTER(vp), C.POINTER(vp)]),
        "bprx_clear_item_grad": (C.c_int, [vp, i64, i32, vp]),
        "bprx_scatter_add": (C.c_int, [vp, i32, i32, vp, vp, i64, f32, vp]),
        "bprx_route_reset": (C.c_int, [vp, i64, vp, i32, vp]),
        "bprx_route_plan": (C.c_int, [vp, i64, vp, i64, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp]),
        "bprx_route_gather": (C.c_int, [vp, i32, vp, i32, i32, vp, i64, vp, vp, vp]),
        "bprx_route_unpack": (C.c_int, [vp, vp, i64, vp, i32, vp, i32, vp, vp, i32, vp]),
        "bprx_route_pack": (C.c_int, [vp, i32, vp, i32, vp, i64, vp, vp, vp, i32, f32, vp, vp, i64, vp, i32, vp]),
        "bprx_route_scatter_add": (C.c_int, [vp, i32, vp, i32, i32, vp, vp, i64, f32, vp, vp]),
        "bprx_score_block": (C.c_int, [vp, i32, i32, vp, vp]),
        "bprx_eval_users": (C.c_int, [vp, i32, i32, vp, vp, vp, vp, vp, i32, vp, vp]),
        "bprx_topk": (C.c_int, [vp, i32, i32, vp, vp, vp, i32, vp, vp, vp, vp]),
        "bprx_eval_pos": (C.c_int, [vp, i32, i32, vp, i32, i32, vp, vp, vp, vp]),
        "bprx_eval_counts": (C.c_int, [vp, i32, i32, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp]),
        "bprx_eval_finish": (C.c_int, [vp, i32, i32, i32, vp, vp, vp, i32, vp, vp]),
        "bprx_sync_check": (C.c_int, [vp, vp]),
        "bprx_probe_stream_read": (i64, [vp, i64, vp, vp]),
        "bprx_probe_stream_read_nt": (i64, [vp, i64, vp, vp]),
        "bprx_probe_row_gather": (i64, [vp, i64, i32, vp, i64, i32, vp, vp]),
        "bprx_profile_enable": (C.c_int, [vp, C.c_int]),
        "bprx_profile_read": (C.c_int, [vp, vp, vp]),
        "bprx_sample_philox": (C.c_int, [vp, vp, vp, i64, i32, C.c_uint64, C.c_uint64, i64, vp, vp, vp, vp]),
        "bprx_user_msg_floats": (C.c_int64, [vp, i64]),
        "bprx_pack_user_msg": (C.c_int, [vp, vp, i64, i64, vp, vp]),
        "bprx_apply_user_msgs": (C.c_int, [vp, vp, i32, i64, C.c_float, vp]),
        "bprx_sum_dense_parts": (C.c_int, [vp, vp, i32, vp]),
        "bprx_epoch_prepare": (C.c_int, [C.c_uint64, u32, i32, vp, vp, vp, vp]),
        "bprx_epoch_slots": (C.c_int, [vp, i32, vp, i64, vp]),
        "bprx_sample_epoch": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, C.c_uint64, u32, i64, i64, vp, vp, vp, vp]),
        "bprx_index_pass_kind": (C.c_int, [vp]),
        "bprx_adam_rows": (C.c_int, [vp, vp, vp, vp, i64, C.c_float, C.c_float, C.c_float, C.c_float, vp]),
        "bprx_step_lr": (C.c_int, [vp, vp]),
        "bprx_sample_philox_h": (C.c_int, [vp, vp, vp, vp, i64, i32, C.c_uint64, C.c_uint64, i64, vp, vp, vp, i64, i64, vp]),
        "bprx_sample_epoch_h": (C.c_int, [vp, vp, vp, vp, vp, vp, i32, i32, C.c_uint64, u32, i64, i64, vp, vp, vp, i64, i64, vp]),
        "bprx_sampler_create": (C.c_int, [vp, vp, i32, i32, C.POINTER(vp)]),
        "bprx_sampler_destroy": (C.c_int, [vp]),
        "bprx_sampler_count": (i64, [vp, i32, i32]),
        "bprx_sampler_ref_stream": (i64, [vp, i32, i32, u32, u32, vp, vp, vp, i64]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)       # AttributeError here == header/library mismatch: fail loudly
        fn.restype, fn.argtypes = res, args
    if L.bprx_abi_version() != ABI_VERSION:
        raise ImportError("libbprx.so ABI %d != binding ABI %d" % (L.bprx_abi_version(), ABI_VERSION))
    _lib = L
    return L


EXPORTS = ["bprx_abi_version", "bprx_create", "bprx_destroy", "bprx_last_error", "bprx_bind_tables", "bprx_set_hyper", "bprx_tables_dirty",
           "bprx_set_adam_step", "bprx_get_adam_step", "bprx_adam_is_lazy", "bprx_sync_adam", "bprx_score_pairs", "bprx_step", "bprx_step_begin",
           "bprx_step_begin_sparse", "bprx_step_begin_dense", "bprx_sum_dense_parts",
           "bprx_dense_grad", "bprx_step_end", "bprx_step_project", "bprx_user_grad", "bprx_clear_user_grad", "bprx_item_grad", "bprx_clear_item_grad",
           "bprx_scatter_add", "bprx_route_reset", "bprx_route_plan", "bprx_route_gather", "bprx_route_unpack", "bprx_route_pack",
           "bprx_route_scatter_add", "bprx_score_block", "bprx_eval_users", "bprx_eval_pos", "bprx_eval_counts", "bprx_eval_finish", "bprx_topk", "bprx_sync_check", "bprx_probe_stream_read", "bprx_probe_stream_read_nt", "bprx_probe_row_gather", "bprx_profile_enable",
           "bprx_profile_read", "bprx_sample_philox", "bprx_epoch_prepare", "bprx_epoch_slots", "bprx_sample_epoch", "bprx_sample_philox_h", "bprx_sample_epoch_h", "bprx_index_pass_kind", "bprx_adam_rows", "bprx_step_lr", "bprx_user_msg_floats", "bprx_pack_user_msg",
           "bprx_apply_user_msgs", "bprx_sampler_create",
           "bprx_sampler_destroy", "bprx_sampler_count", "bprx_sampler_ref_stream"]


def check(handle, rc):
    if rc < 0:
        msg = lib().bprx_last_error(handle)
        raise BprxError(rc, msg.decode() if msg else "?")
    return rc
