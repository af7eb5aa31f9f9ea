"""ctypes binding of ``libstv_hip.so`` (C ABI declared in ``include/stv.h``).

The library is the product: there is no eager/PyTorch or CPU fallback.  If it
is missing or a call fails, a ``RuntimeError`` is raised.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("STV_LIB_PATH") or os.path.join(_HERE, "libstv_hip.so")   # override: diagnostic builds

STV_F32, STV_BF16 = 0, 1
RELU_IN, RELU_OUT, MASK, ACCUM, W_BLOCKED, POOL_IDX, POOL_ROUTE, POOL_ONLY = 1, 2, 4, 8, 16, 32, 64, 128
LANE_SIDE, LANE_JOIN = 1 << 29, 1 << 30          # scheduling hints of the command-buffer executor

(OP_CONV_FIRST_FWD, OP_CONV_FIRST_DGRAD, OP_CONV, OP_POOL_FWD, OP_POOL_BWD, OP_RELU_FWD,
 OP_RELU_BWD, OP_GRAM_PARTIAL, OP_GRAM_FINISH, OP_CONTENT_LOSS, OP_CONTENT_GRAD,
 OP_LOSS_COMBINE, OP_MEMSET, OP_GRAM_MULTI, OP_LBFGS_STEP) = range(1, 16)

CONTENT_LOSS_PARTS = 256

_ERRORS = {1: "STV_ERR_ARG (unsupported shape / null pointer / dtype)",
           2: "STV_ERR_LAUNCH (HIP launch failed)", 3: "STV_ERR_ALLOC", 4: "STV_ERR_GRAPH"}


class StvOp(ctypes.Structure):
    """Mirror of ``stv_op_t`` (include/stv.h)."""

    _fields_ = [
        ("op", c_int32), ("dtype", c_int32), ("flags", c_int32), ("taps", c_int32),
        ("H", c_int32), ("W", c_int32), ("cin", c_int32), ("cout", c_int32),
        ("n", c_int64),
        ("f0", c_float), ("f1", c_float), ("f2", c_float), ("f3", c_float),
        ("p0", c_void_p), ("p1", c_void_p), ("p2", c_void_p), ("p3", c_void_p),
        ("q0", c_void_p), ("q1", c_void_p), ("q2", c_void_p), ("q3", c_void_p),
    ]


class StvGramTap(ctypes.Structure):
    """Mirror of ``stv_gram_tap_t`` (include/stv.h)."""

    _fields_ = [
        ("F", c_void_p), ("partials", c_void_p), ("target", c_void_p), ("gram_out", c_void_p),
        ("loss_part", c_void_p), ("sgrad", c_void_p), ("coef_dev", c_void_p),
        ("n_pixels", c_int32), ("channels", c_int32),
        ("clamp_max", c_float), ("norm", c_float), ("coef", c_float),
    ]


# name -> (restype, argtypes); every symbol include/stv.h declares
SIGNATURES = {
    "stv_version": (c_int, []),
    "stv_gram_partials_bytes": (c_size_t, [c_int, c_int]),
    "stv_gram_ksplit": (c_int, [c_int, c_int]),
    "stv_gram_loss_parts": (c_int, [c_int]),
    "stv_conv_first_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "stv_conv_first_dgrad": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "stv_conv_first_packed_bytes": (c_size_t, [c_int, c_int]),
    "stv_conv_first_pack": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "stv_conv_first_fwd_packed": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "stv_conv_first_gram_supported": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "stv_conv_first_fwd_gram": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "stv_conv_first_dgrad_packed": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "stv_conv_igemm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "stv_conv_igemm_pool": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                    c_void_p]),
    "stv_conv_igemm_route": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "stv_conv_igemm_dual": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                    c_int, c_int, c_int, c_void_p]),
    "stv_conv_tune": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "stv_conv_tune_export": (c_int, [ctypes.POINTER(c_int), c_int]),
    "stv_conv_tune_import": (c_int, [ctypes.POINTER(c_int), c_int]),
    "stv_conv_next_weights": (None, [c_void_p, c_size_t]),
    "stv_conv_workspace": (None, [c_void_p, c_size_t]),
    "stv_conv_workspace_bytes": (c_size_t, []),
    "stv_conv_num_configs": (c_int, []),
    "stv_conv_uses_ws": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "stv_conv_config": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "stv_maxpool_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "stv_maxpool_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "stv_relu_fwd": (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "stv_relu_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int, c_void_p]),
    "stv_gram_partial": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "stv_gram_finish": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_float, c_float, c_void_p, c_int, c_void_p]),
    "stv_gram_multi": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "stv_content_loss": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "stv_content_loss_grad": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_float, c_int, c_void_p]),
    "stv_content_grad": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_float, c_void_p, c_int, c_int, c_void_p]),
    "stv_image_to_u8": (c_int, [c_void_p, c_void_p, c_int, c_int, ctypes.POINTER(c_float), ctypes.POINTER(c_float), c_int, c_void_p]),
    "stv_loss_combine": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_float, c_float, c_void_p, c_void_p, c_void_p]),
    "stv_loss_combine_log": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_float, c_float, c_void_p, c_void_p, c_void_p,
                                     c_int, c_void_p, c_void_p, c_void_p]),
    "stv_host_mailbox_alloc": (c_int, [c_size_t, ctypes.POINTER(c_void_p)]),
    "stv_host_mailbox_free": (None, [c_void_p]),
    "stv_lbfgs_state_bytes": (c_size_t, [c_int]),
    "stv_lbfgs_workspace_bytes": (c_size_t, [c_size_t, c_int]),
    "stv_lbfgs_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int, c_float, c_float, c_float, c_void_p]),
    "stv_lbfgsc_state_bytes": (c_size_t, [c_int]),
    "stv_lbfgsc_workspace_bytes": (c_size_t, [c_size_t, c_int]),
    "stv_lbfgsc_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int, c_float, c_float, c_float, c_void_p]),
    "stv_lbfgsc_dots": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int, c_void_p]),
    "stv_lbfgsc_apply": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_float, c_float, c_float, c_void_p]),
    "stv_lbfgsc_dots_offset": (c_size_t, [c_size_t, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "stv_adam_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_float, c_float, c_float, c_float, c_float, c_float, c_float, c_void_p]),
    "stv_program_create": (c_int, [ctypes.POINTER(StvOp), c_int, ctypes.POINTER(c_void_p)]),
    "stv_program_run": (c_int, [c_void_p, c_int, c_void_p]),
    "stv_program_profile": (c_int, [c_void_p, c_void_p, ctypes.POINTER(c_float), c_int]),
    "stv_program_profile_reps": (c_int, [c_void_p, c_void_p, c_int, ctypes.POINTER(c_float), c_int]),
    "stv_program_op_count": (c_int, [c_void_p]),
    "stv_program_destroy": (None, [c_void_p]),
}

_lib = None


def load() -> ctypes.CDLL:
    """Load the HIP library or fail loudly (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    # Load order matters: PyTorch ships its own libamdhip64; if this library pulled in the system copy
    # first, the process would hold two HIP runtimes and every pointer / stream handed over from torch
    # would be foreign to the kernels here (seen as STV_ERR_LAUNCH on the first call).
    import torch  # noqa: F401, PLC0415
    if not os.path.exists(LIB_PATH):
        msg = (f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
               "(hipcc --offload-arch=gfx950). There is no CPU/eager fallback for this path.")
        raise RuntimeError(msg)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    _import_tile_table(lib)
    return lib


TILE_TABLE_PATH = os.environ.get("STV_TILE_TABLE") or os.path.join(_HERE, "conv_tiles_gfx950.json")
tile_table_info: dict = {"source": None, "entries": 0}


def _import_tile_table(lib: ctypes.CDLL) -> None:
    """Hand the persisted tile choices (tools/tune_tiles.py, measured on an MI355X) to the library: which tile a
    conv shape runs on is then the same in every process (results bit-reproducible run to run, kernel names in a
    profile reproducible).  STV_CONV_TUNE=0 makes the library ignore the table (analytic choice); =1 re-measures
    shapes the table does not hold.  A missing file is fine: analytic choices."""
    import json  # noqa: PLC0415
    import warnings  # noqa: PLC0415
    if not os.path.exists(TILE_TABLE_PATH):
        return

    def give_up(why: str) -> None:
        # a table the library cannot take (a tile index that no longer exists, a malformed entry, another device's
        # measurements) must not make the package unusable: forget it and run on the analytic choices
        lib.stv_conv_tune_import(None, 0)
        tile_table_info.update(source=None, entries=0, ignored=f"{os.path.basename(TILE_TABLE_PATH)}: {why}")
        warnings.warn(f"conv tile table {TILE_TABLE_PATH} ignored ({why}): analytic tile choices", RuntimeWarning, stacklevel=3)
    try:
        with open(TILE_TABLE_PATH) as fh:
            doc = json.load(fh)
        rows = [int(v) for e in doc.get("entries", []) for v in (e["H"], e["W"], e["cin"], e["cout"], e["taps"], e["elem_bytes"], e["cfg"])]
    except (OSError, ValueError, KeyError, TypeError, AttributeError) as exc:
        give_up(f"unreadable: {exc!r}")
        return
    if not rows:
        return
    arch = _device_arch()
    measured_on = str(doc.get("arch") or "")
    if arch is not None and measured_on and arch != measured_on:
        give_up(f"measured on {measured_on}, this device is {arch}")
        return
    arr = (c_int * len(rows))(*rows)
    rc = lib.stv_conv_tune_import(arr, len(rows) // 7)
    if rc != 0:
        give_up(f"stv_conv_tune_import: {_ERRORS.get(rc, rc)}")
        return
    tile_table_info.update(source=os.path.basename(TILE_TABLE_PATH), entries=len(rows) // 7, measured_on=doc.get("device"),
                           tool=doc.get("tool"))


def _device_arch() -> str | None:
    """gcnArchName of the current GPU ("gfx950"), or None when there is none (host-only use of the library)."""
    import torch  # noqa: PLC0415
    try:
        if torch.cuda.device_count() == 0 or not torch.cuda.is_available():
            return None
        return str(torch.cuda.get_device_properties(torch.cuda.current_device()).gcnArchName).split(":")[0]
    except (RuntimeError, AttributeError, AssertionError):
        return None


def export_tile_table() -> list[dict]:
    """The library's current tile table (imported + measured in this process) as a list of dicts."""
    lib = load()
    n = int(lib.stv_conv_tune_export(None, 0))
    arr = (c_int * (7 * max(n, 1)))()
    n = min(n, int(lib.stv_conv_tune_export(arr, n)))
    keys = ("H", "W", "cin", "cout", "taps", "elem_bytes", "cfg")
    return [dict(zip(keys, (int(arr[7 * i + k]) for k in range(7)), strict=True)) for i in range(n)]


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = f"{what} failed: {_ERRORS.get(rc, rc)}"
        raise RuntimeError(msg)
