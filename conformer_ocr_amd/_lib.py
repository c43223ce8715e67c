"""ctypes binding of libcocr_hip.so (include/cocr.h).  There is no CPU fallback: if the HIP
library cannot be built or loaded every compute call raises."""
from __future__ import annotations

import ctypes as C
import os
import threading
import warnings
from typing import Optional

from . import build as _build

F32, BF16, U8, I64 = 0, 1, 2, 3
OK, EINVAL, ESTATE, EHIP, EUNSUPPORTED = 0, -1, -2, -3, -4


class HParamsC(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        'num_classes', 'height', 'encoder_dim', 'num_encoder_layers', 'num_attention_heads',
        'feed_forward_expansion_factor', 'conv_expansion_factor', 'conv_kernel_size', 'half_step_residual',
        'subsampling_conv_channels', 'subsampling_factor')]


# every symbol include/cocr.h declares: name -> (restype, argtypes)
_P = C.c_void_p
_I = C.c_int
_I32P = C.POINTER(C.c_int32)
SYMBOLS = {
    'cocr_last_error': (C.c_char_p, []),
    'cocr_version': (C.c_char_p, []),
    'cocr_create': (_I, [C.POINTER(HParamsC), _I, C.POINTER(_P)]),
    'cocr_destroy': (None, [_P]),
    'cocr_set_tensor': (_I, [_P, C.c_char_p, _P, _I, _I, C.POINTER(C.c_int64)]),
    'cocr_missing_tensors': (_I, [_P, C.c_char_p, C.c_size_t]),
    'cocr_finalize': (_I, [_P, _I]),
    'cocr_finalize_empty': (_I, [_P, _I]),
    'cocr_weight_blob': (_I, [_P, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    'cocr_blob_export': (_I, [_P, _P, C.c_size_t, _P]),
    'cocr_blob_import': (_I, [_P, _P, C.c_size_t, _P]),
    'cocr_out_len': (C.c_int32, [C.c_int32, C.c_int32]),
    'cocr_share_weights': (_I, [_P, _P]),
    'cocr_collate_lines': (_I, [C.POINTER(_P), _I32P, _I, _I, _I, _P, _I, _I]),
    'cocr_reserve': (_I, [_P, _I, _I]),
    'cocr_forward': (_I, [_P, _P, _I, _I, _I, _I, _I32P, _P, _I32P, _P]),
    'cocr_ctc_greedy': (_I, [_P, _P, _I, _I, _I, _I32P, _P, _P, _P, _P, _P, _I, _P]),
    'cocr_forget_argmax': (_I, [_P]),
    'cocr_ctc_beam': (_I, [_P, _P, _I, _I, _I, _I32P, _P, _P, _P, _P, _P, _I, _I, _P]),
    'cocr_ctc_loss': (_I, [_P, _P, _I, _I, _I, _I32P, _I32P, _I32P, _P, _P, _P]),
    'cocr_decoder_backward': (_I, [_P, _P, _I, _I, _P, _P, _P, _P]),
    'cocr_decoder_adamw': (_I, [_P, _P, _P, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _P]),
    'cocr_train_begin': (_I, [_P]),
    'cocr_train_step': (_I, [_P, _P, _I, _I, _I, _I, _I32P, _I32P, _I32P, C.POINTER(C.c_float), C.c_uint64, C.POINTER(C.c_float), _P]),
    'cocr_train_get': (_I, [_P, C.c_char_p, _I, _P, C.c_int64, _P]),
    'cocr_train_adamw': (_I, [_P, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _P]),
    'cocr_train_end': (_I, [_P]),
    'cocr_train_set_matmul': (_I, [_P, _I]),
    'cocr_train_grad_buffer': (_I, [_P, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    'cocr_train_param_buffer': (_I, [_P, C.POINTER(_P), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    'cocr_train_layout': (_I, [_P, C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    'cocr_get_tensor': (_I, [_P, C.c_char_p, _P, C.c_int64, _P]),
    'cocr_preproc_width': (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    'cocr_preproc_lines': (_I, [_P, _P, C.POINTER(C.c_int64), _I32P, _I32P, _I32P, _I, _I, _I, _I, _P, _I32P, _P]),
    'cocr_set_graph': (_I, [_P, _I]),
    'cocr_set_chain_rows': (_I, [_P, _I]),
    'cocr_set_debug': (_I, [_P, _I]),
    'cocr_debug_tap': (_I, [_P, C.c_char_p, _P, C.c_int64, C.POINTER(C.c_int64)]),
    'cocr_profile': (_I, [_P, _I]),
    'cocr_profile_read': (_I, [_P, C.c_char_p, C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_int64), _I]),
}

_lock = threading.Lock()
_lib: Optional[C.CDLL] = None


def lib_path() -> str:
    return os.environ.get('COCR_LIB_PATH') or _build.LIB


def load(build_if_missing: bool = True) -> C.CDLL:
    """Loads (building first if needed) the in-tree library.  torch is imported first so that its
    HIP runtime (same SONAME libamdhip64.so.7) is the one both sides share: device pointers and
    streams cross the boundary."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        import torch  # noqa: F401  (loads libamdhip64 of the torch wheel)
        path = lib_path()
        dev_lib = path != _build.LIB                  # COCR_LIB_PATH: an experiment build, loaded as it is
        if build_if_missing and not dev_lib and not _build.up_to_date():
            try:
                _build.build(verbose=False)
            except Exception as e:
                if not os.path.exists(path):
                    raise RuntimeError(f'libcocr_hip.so is missing and cannot be built: {e}') from e
                # a library built from OTHER sources than the tree's: numbers and test results would describe kernels that are
                # not the committed ones.  Refused unless explicitly allowed.
                msg = (f'libcocr_hip.so was built from sources {_build.built_hash() or "?"} but the tree is {_build.source_hash()} '
                       f'and the rebuild failed ({e})')
                if os.environ.get('COCR_ALLOW_STALE_LIB') != '1':
                    raise RuntimeError(msg + '; set COCR_ALLOW_STALE_LIB=1 to load the stale library anyway') from e
                warnings.warn(msg + '; loading the STALE library (COCR_ALLOW_STALE_LIB=1)')
        if not os.path.exists(path):
            raise RuntimeError(f'{path} not found: run `python -m conformer_ocr_amd.build` (no CPU fallback exists)')
        lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)      # AttributeError if the library does not export it
            fn.restype = res
            fn.argtypes = args
        ver = (lib.cocr_version() or b'').decode()
        if not dev_lib and _build.built_hash() and f'src={_build.built_hash()}' not in ver and os.environ.get('COCR_ALLOW_STALE_LIB') != '1':
            raise RuntimeError(f'{path} reports "{ver}" but its build record says src={_build.built_hash()}: rebuild with '
                               '`python -m conformer_ocr_amd.build --force`')
        _lib = lib
        return lib


def check(rc: int) -> int:
    """Maps a negative return code to the Python exception the reference would raise."""
    if rc >= 0:
        return rc
    msg = (load().cocr_last_error() or b'').decode('utf-8', 'replace')
    if rc == EINVAL:
        raise ValueError(msg)
    if rc == EUNSUPPORTED:
        raise NotImplementedError(msg)
    raise RuntimeError(msg)
