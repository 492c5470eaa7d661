"""ctypes binding of libarchi_hip.so (include/archi_knn.h).

There is no CPU fallback: if the HIP library is missing or no gfx950 device is
visible, every entry point raises. The oracle under oracle/ is test
infrastructure and is never imported from here.
"""
from __future__ import annotations

import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libarchi_hip.so")

DTYPES = {"f32": 0, "bf16": 1, "f16": 2}
# store-level metric names (postgres_vectorstore.py:74-78) -> AK_METRIC_*
METRICS = {"cosine": 0, "l2": 1, "inner_product": 2}
SEARCH_MODES = {"auto": 0, "exact": 1, "fast_only": 2}
POOLING = {"mean": 0, "cls": 1}


class HipBackendError(RuntimeError):
    """Raised when the HIP backend is unavailable or a call fails."""


class StaleFilterError(HipBackendError):
    """A search was handed a row_filter built for another layout of the index (AK_ERR_STALE_FILTER): a writer added rows
    or reclaimed tombstones between the mask's construction and the search. Rebuild the mask and retry."""


ERR_STALE_FILTER = -11
ERR_COMM_BROKEN = -13
ABI_VERSION = 4          # AK_ABI_VERSION of include/archi_knn.h this binding was written against

# switches that exist only in libarchi_hip_dbg.so (`make -C archi_amd/csrc dbg`): instrumented kernels, stage-skipping
# ablations (WRONG RESULTS) and the superseded kernel generations kept as A/B references. One of them in the environment --
# or ARCHI_HIP_DBG=1 -- makes load() pick that library.
DBG_SWITCHES = ("AK_SCAN_DBG", "AK_SCAN_ABLATE", "AK_FFN_DBG", "AK_FFN_ABLATE", "AK_TAIL_ABLATE", "AK_QKV_DBG", "AK_GEMM_ABLATE",
                "AK_ENC_NOFFN", "AK_FFN_W8", "AK_FFN_PAIR", "AK_FFN_ATT", "AK_ATTN_DBG")


class AkBertConfig(ctypes.Structure):
    _fields_ = [
        ("vocab_size", ctypes.c_int),
        ("hidden", ctypes.c_int),
        ("layers", ctypes.c_int),
        ("heads", ctypes.c_int),
        ("intermediate", ctypes.c_int),
        ("max_position", ctypes.c_int),
        ("type_vocab", ctypes.c_int),
        ("ln_eps", ctypes.c_float),
        ("residual_bf16", ctypes.c_int),
        ("precision", ctypes.c_int),
    ]


_lock = threading.Lock()
_lib = None
_inited_device = None

# every symbol include/archi_knn.h declares: (name, restype, argtypes)
_P, _I, _I64, _U64, _U32 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_uint64, ctypes.c_uint32
SYMBOLS = [
    ("ak_last_error", ctypes.c_char_p, []),
    ("ak_version", ctypes.c_char_p, []),
    ("ak_abi_version", _I, []),
    ("ak_debug_set", _I, [ctypes.c_char_p, ctypes.c_char_p]),
    ("ak_init", _I, [_I]),
    ("ak_device_info", _I, [ctypes.c_char_p, _I, ctypes.POINTER(_I), ctypes.POINTER(_I64)]),
    ("ak_sync", _I, [_P]),
    ("ak_index_create", _I, [_I64, _I, _I, _I, ctypes.POINTER(_P)]),
    ("ak_index_destroy", _I, [_P]),
    ("ak_index_add", _I, [_P, _P, _I, _I64, _P, _I]),
    ("ak_index_generate", _I, [_P, _U64, _U32, _U64, _I64, _I, _I64]),
    ("ak_index_remove", _I, [_P, _P, _I64, ctypes.POINTER(_I64)]),
    ("ak_index_count", _I, [_P, ctypes.POINTER(_I64)]),
    ("ak_index_fetch", _I, [_P, _P, _I64, _P]),
    ("ak_index_lookup", _I, [_P, _P, _I64, _P]),
    ("ak_index_distances", _I, [_P, _P, _P, _I64, _P, _P]),
    ("ak_index_search", _I, [_P, _P, _I, _I, _I, _P, _I64, _U64, _P, _P, _P, _P]),
    ("ak_index_search_dev", _I, [_P, _P, _I, _I, _I, _P, _I64, _U64, _P, _P, _P, _P]),
    ("ak_index_slots", _I, [_P, ctypes.POINTER(_I64), ctypes.POINTER(_I64), ctypes.POINTER(_U64)]),
    ("ak_index_compact", _I, [_P, ctypes.POINTER(_I64)]),
    ("ak_index_scan_plan", _I, [_P, _I, _I, _P]),
    ("ak_index_debug_read", _I, [_P, _P, _I]),
    ("ak_index_profile", _I, [_P, _I]),
    ("ak_index_profile_read", _I, [_P, _P, _I, ctypes.POINTER(_I)]),
    ("ak_merge_topk_dev", _I, [_I, _I, _I, _P, _P, _P, _P, _P]),
    ("ak_merge_shards_dev", _I, [_I, _I, _I, _P, _I64, _P, _P, _P, _P]),
    ("ak_comm_unique_id", _I, [_P]),
    ("ak_comm_create", _I, [_P, _I, _I, ctypes.POINTER(_P)]),
    ("ak_comm_destroy", _I, [_P]),
    ("ak_index_search_sharded_dev", _I, [_P, _P, _P, _I, _I, _P, _I64, _U64, _P, _P, ctypes.POINTER(_I64), _P]),
    ("ak_shard_payload_begin_dev", _I, [_P, _I, _I, _P]),
    ("ak_shard_fail_payload_dev", _I, [_P, _I, _I, _I, _P]),
    ("ak_shard_status_dev", _I, [_I, _P, _I64, _P, _P]),
    ("ak_shard_gather_rows_dev", _I, [_P, _P, _I, _I, _P, _P]),
    ("ak_shard_scatter_topk_dev", _I, [_P, _I, _I, _P, _P, _P, _P, _P]),
    ("ak_l2_normalize_dev", _I, [_P, _I64, _I, _P]),
    ("ak_encoder_create", _I, [ctypes.POINTER(AkBertConfig), _P, _I, ctypes.POINTER(_P)]),
    ("ak_encoder_destroy", _I, [_P]),
    ("ak_encoder_forward", _I, [_P, _P, _P, _I, _I, _I, _I, _P, _P]),
    ("ak_encoder_forward_lens", _I, [_P, _P, _I, _P, _I, _I, _I, _I, _I, _P, _P]),
    ("ak_encoder_gelu_table", _I, [_P]),
    ("ak_wordpiece_create", _I, [ctypes.c_char_p, _I, ctypes.POINTER(_P)]),
    ("ak_wordpiece_destroy", _I, [_P]),
    ("ak_wordpiece_encode", _I, [_P, _P, _P, _I64, _I, _I, _P, _P]),
]


def load() -> ctypes.CDLL:
    """dlopen the library and bind every declared symbol (no GPU needed)."""
    global _lib
    with _lock:
        if _lib is None:
            # torch ships its own libamdhip64.so.7 + libhsa-runtime64; two HIP runtimes in one
            # process cannot both open the GPU. Loading torch's first makes the dynamic loader
            # resolve our NEEDED libamdhip64.so.7 to the copy torch already mapped.
            import torch  # noqa: F401
            # measurement switches select the library that carries the instrumented kernel instantiations (`make dbg`)
            path = LIB_PATH
            if os.environ.get("ARCHI_HIP_DBG") or any(os.environ.get(v) for v in DBG_SWITCHES) or \
                    os.environ.get("AK_SCAN_CFG", "")[:1] in ("X", "O") or os.environ.get("AK_ATTN_STREAM", "") in ("3", "4"):
                dbg = os.path.join(_HERE, "lib", "libarchi_hip_dbg.so")
                if not os.path.exists(dbg):
                    raise HipBackendError(f"{dbg} not found: the instrumented kernels are built by `make -C archi_amd/csrc dbg`")
                path = dbg
            if not os.path.exists(LIB_PATH):
                raise HipBackendError(
                    f"{LIB_PATH} not found: build it with `make -C archi_amd/csrc` "
                    "(or __graft_entry__.build()); archi_amd has no CPU fallback")
            lib = ctypes.CDLL(path)
            for name, res, args in SYMBOLS:
                fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
                fn.restype = res
                fn.argtypes = args
            got = lib.ak_abi_version()
            if got != ABI_VERSION:
                raise HipBackendError(f"{path}: ABI version {got}, this binding expects {ABI_VERSION} "
                                      "(rebuild with `make -C archi_amd/csrc`; signatures moved between the two)")
            _lib = lib
    return _lib


def is_dbg_library() -> bool:
    """True when load() picked libarchi_hip_dbg.so (the A/B reference kernels and the stage-skipping switches live there)."""
    lib = load()
    return bool(getattr(lib, "_name", "").endswith("libarchi_hip_dbg.so"))


def debug_set(name: str, value: str | None) -> None:
    """Set one of the library's measurement switches for this process (ak_debug_set): the library reads its environment once,
    so a test or probe that wants another scan tile mid-process says so here. None restores the default."""
    check(load().ak_debug_set(name.encode(), None if value is None else str(value).encode()), f"ak_debug_set({name})")


def last_error() -> str:
    return (load().ak_last_error() or b"").decode("utf-8", "replace")


def check(rc: int, what: str) -> None:
    if rc == ERR_STALE_FILTER:
        raise StaleFilterError(f"{what}: {last_error()}")
    if rc != 0:
        raise HipBackendError(f"{what} failed (rc={rc}): {last_error()}")


def init(device: int | None = None) -> ctypes.CDLL:
    """Bind this process to one GPU (one process per GPU). LOCAL_RANK selects it by default."""
    global _inited_device
    lib = load()
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0")) if _inited_device is None else _inited_device
    if _inited_device != device:
        check(lib.ak_init(device), "ak_init")
        _inited_device = device
    return lib


def bound_device() -> int:
    """The GPU this process is bound to (ak_init's device; one process per GPU). torch tensors and streams handed to
    the C ABI must live on THIS device -- torch.cuda.current_device() is a per-thread default that a torchrun rank > 0
    process may never have set."""
    if _inited_device is None:
        raise HipBackendError("libarchi_hip is not initialised: call archi_amd._lib.init() first")
    return _inited_device
