"""ctypes binding of `libhilcodec_amd.so` (the C ABI declared in `include/hilcodec_amd.h`).

The HIP library IS the product: there is no PyTorch/CPU fallback.  If the shared object is missing
or a symbol cannot be resolved this module raises at import time, and every op raises
`RuntimeError` when handed a non-GPU tensor."""
from __future__ import annotations

import ctypes as C
import os

# PyTorch-ROCm bundles its own HIP runtime (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7).  It
# must be in the process BEFORE our library is dlopen'ed so that the kernels, the streams and the
# device memory all belong to ONE runtime; loaded the other way round the library binds to
# /opt/rocm's copy and every launch fails with "no ROCm-capable device is detected".
import torch  # noqa: F401,E402

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HILC_LIB") or os.path.join(_HERE, "lib", "libhilcodec_amd.so")   # HILC_LIB: A/B builds (tools/)

_f, _i, _p, _d = C.c_float, C.c_int, C.c_void_p, C.c_double

# name -> argtypes, in the order of include/hilcodec_amd.h
SIGNATURES = {
    "hilc_pw_conv": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _i, _f, _p],
    "hilc_dws_conv": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _i, _f, _i, _p],
    "hilc_up_conv": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _i, _p],
    "hilc_resblock": [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _f, _p],
    "hilc_resblock_supported": [_i, _i],
    "hilc_resblock_stream_supported": [_i, _i],
    "hilc_dws_conv_wave_row": [_i, _i, _i, _i],
    "hilc_resblock_pack_weights": [_p, _p, _i, _p],
    "hilc_dws_conv_stream": [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _i, _f, _i, _p],
    "hilc_up_conv_expand_taps": [_p, _p, _i, _i, _p],
    "hilc_up_conv_expanded": [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _i, _p],
    "hilc_up_conv_stream": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _i, _p],
    "hilc_resblock_balanced": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _f, _p],
    "hilc_resblock_chain_supported": [_i, _i, _i, _i],
    "hilc_resblock_chain_row_classes": [_i],
    "hilc_resblock_chain_row_classes_offline": [_i],
    "hilc_resblock_pack_weights_rc": [_p, _p, _i, _i, _p],
    "hilc_resblock_chain": [_p, _p, _p, _i, _i, _i, _i, _i, _p],
    "hilc_decoder_stage_supported": [_i, _i, _i, _i, _i],
    "hilc_decoder_stage": [_p, _p, _i, _p, _i, _i, _i, _i, _p],
    "hilc_decoder_stage_post_supported": [_i, _i, _i, _i, _i],
    "hilc_decoder_stage_post": [_p, _p, _i, _p, _i, _i, _i, _i, _p],
    "hilc_encoder_stage0_supported": [_i, _i, _i, _i, _i, _i, _i],
    "hilc_encoder_stage0": [_p, _p, _i, _p, _i, _i, _i, _p],
    "hilc_encoder_stage_supported": [_i, _i, _i, _i, _i],
    "hilc_encoder_stage": [_p, _p, _i, _p, _i, _i, _i, _i, _p],
    "hilc_resblock_stream": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _f, _p],
    "hilc_dw_conv": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _i, _f, _i, _p],
    "hilc_dw_convtr": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _i, _p],
    "hilc_conv_pre": [_p, _p, _i, _p, _p, _p, _i, _i, _i, _i, _f, _p],
    "hilc_conv_post": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _i, _f, _i, _p],
    "hilc_stft_logmag": [_p, _p, _i, _p, _p, _i, _i, _i, _i, _f, _f, _i, _p],
    "hilc_tail": [_p, _p, _p, C.c_long, _i, _i, _i, _p],
    "hilc_spec_block_supported": [_i, _i, _i, _i],
    "hilc_spec_block_packed_floats": [_i, _i],
    "hilc_spec_block_pack": [_p, _p, _i, _i, _i, _p],
    "hilc_spec_block_conv_pre": [_p, _p, _i, _p, _p, _p, _p, _p, _p, _f, _p, _i, _i, _i, _i, _i, _f, _f, _i, _f, _p],
    "hilc_spec_block": [_p, _p, _i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _f, _i, _f, _p],
    "hilc_l2norm": [_p, _p, _i, _i, _i, _f, _f, _i, _p],
    "hilc_rvq_encode": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "hilc_mse_finalize": [_p, _p, _i, _d, _p],
    "hilc_rvq_decode": [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "hilc_rvq_encode_mixed": [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "hilc_rvq_ema_stats": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "hilc_rvq_ema_update": [_p, _p, _p, _p, _d, _i, _i, _i, _p],
    "hilc_rvq_decode_mixed": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p],
}

ABI_VERSION = 15


def source_hash() -> str:
    """sha256 (first 16 hex digits) over the kernel sources and the C header the library is built from: the build id that
    profiles/latest_mfma_family.json is stamped with (tools/summarize_profile.py) and bench.py compares against."""
    import glob
    import hashlib
    h = hashlib.sha256()
    root = os.path.dirname(_HERE)
    files = sorted(glob.glob(os.path.join(_HERE, "csrc", "*.h*"))) + [os.path.join(root, "include", "hilcodec_amd.h")]
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


class ResblockParams(C.Structure):
    """`hilc_resblock_params` of include/hilcodec_amd.h: one residual block of a chain launch"""
    _fields_ = [("w1t", _p), ("dw1_w", _p), ("dw1_b", _p), ("w2t", _p), ("dw2_w", _p), ("dw2_b", _p),
                ("hist1", _p), ("hist2", _p), ("hist1_out", _p), ("hist2_out", _p), ("pre_scale", _f), ("out_scale", _f)]


class UpParams(C.Structure):
    """`hilc_up_params` of include/hilcodec_amd.h: the up-sampling layer of a decoder stage launch"""
    _fields_ = [("x", _p), ("tr_w", _p), ("w_lo", _p), ("w_hi", _p), ("bias", _p), ("hist", _p), ("hist_out", _p),
                ("in_scale", _f), ("stride", _i)]


class Spec0Params(C.Structure):
    """`hilc_spec0_params` of include/hilcodec_amd.h: first conv + stage-0 SpecBlock as the opening phase of hilc_encoder_stage0"""
    _fields_ = [("wav", _p), ("dft_packed", _p), ("nyq_sin", _p), ("pw_packed", _p), ("bias", _p), ("pre_w", _p), ("pre_b", _p),
                ("hist", _p), ("hist_len", _i),
                ("pre_in_scale", _f), ("mean", _f), ("std", _f), ("out_scale", _f), ("normalize", _i), ("n_fft", _i), ("hop", _i),
                ("pre_ksize", _i)]


class PostParams(C.Structure):
    """`hilc_post_params` of include/hilcodec_amd.h: the decoder's closing conv behind its last stage (hilc_decoder_stage_post)"""
    _fields_ = [("w", _p), ("bias", _p), ("wav", _p), ("hist", _p), ("hist_out", _p), ("in_scale", _f), ("out_scale", _f), ("do_tanh", _i),
                ("ksize", _i)]


class DownParams(C.Structure):
    """`hilc_down_params` of include/hilcodec_amd.h: the down-sampling layer of an encoder stage launch"""
    _fields_ = [("w_lo", _p), ("w_hi", _p), ("dw_w", _p), ("dw_b", _p), ("hist", _p), ("hist_out", _p), ("res", _p), ("y", _p),
                ("in_scale", _f), ("stride", _i)]


class HilcodecLibraryError(RuntimeError):
    pass


def _load() -> C.CDLL:
    if not os.path.isfile(LIB_PATH):
        raise HilcodecLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.argtypes = argtypes
        fn.restype = _i
    lib.hilc_abi_version.restype = _i
    lib.hilc_error_string.restype = C.c_char_p
    lib.hilc_error_string.argtypes = [_i]
    lib.hilc_last_hip_error.restype = C.c_char_p
    if lib.hilc_abi_version() != ABI_VERSION:
        raise HilcodecLibraryError(f"ABI mismatch: library {lib.hilc_abi_version()} != binding {ABI_VERSION}")
    return lib


lib = _load()


def check(code: int, what: str) -> None:
    if code != 0:
        msg = lib.hilc_error_string(code).decode()
        if code == -5:
            raise AssertionError(msg)      # reference: `assert 1 <= n <= len(self.layers)`
        if code == -3:
            msg += " — " + lib.hilc_last_hip_error().decode()
        raise RuntimeError(f"{what}: {msg} (code {code})")
