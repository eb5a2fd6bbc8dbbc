"""ctypes binding of libdotaclient_hip.so (the C ABI declared in include/dotaclient_hip.h).

There is deliberately no fallback: if the library is missing the import of any compute entry point
raises, so a GPU test can never pass on a silent CPU/PyTorch path.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('DC_LIB') or os.path.join(HERE, 'libdotaclient_hip.so')   # DC_LIB: an A/B build (build.py)

c_f32p = ctypes.c_void_p
c_ptr = ctypes.c_void_p
c_int = ctypes.c_int
c_i64 = ctypes.c_int64
c_dbl = ctypes.c_double
c_flt = ctypes.c_float

# name -> (restype, argtypes); must list every symbol declared in include/dotaclient_hip.h
SIGNATURES = {
    'dc_abi_version': (c_int, []),
    'dc_last_error': (ctypes.c_char_p, []),
    'dc_gae_scan': (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_dbl, c_dbl, c_ptr, c_ptr, c_ptr]),
    'dc_discount': (c_int, [c_ptr, c_int, c_dbl, c_ptr, c_ptr]),
    'dc_advantage_returns': (c_int, [c_ptr, c_ptr, c_int, c_dbl, c_dbl, c_ptr, c_ptr, c_ptr]),
    'dc_gemm_f32': (c_int, [c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                            c_ptr, c_int, c_ptr, c_int, c_int, c_int, c_ptr, c_i64, c_ptr]),
    'dc_gemm_x3': (c_int, [c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                           c_ptr, c_int, c_ptr, c_int, c_int, c_int, c_ptr, c_i64, c_ptr]),
    'dc_dp_average_grads': (c_int, [c_ptr, c_ptr, c_ptr, c_int, c_int, c_ptr, c_ptr, c_flt, c_ptr]),
    'dc_pack_rows': (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int]),
    'dc_profile_enable': (c_int, [c_int]),
    'dc_profile_report': (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int]),
    'dc_workspace_layout': (c_i64, [c_ptr, c_ptr]),
    'dc_policy_forward': (c_int, [c_ptr] * 13),
    'dc_chunk_initial_state': (c_int, [c_ptr, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_ptr]),
    'dc_policy_single': (c_int, [c_ptr] * 11),
    'dc_select_logp': (c_int, [c_ptr] * 8),
    'dc_ppo_loss_fwd_bwd': (c_int, [c_ptr] * 9 + [c_flt, c_flt, c_flt, c_ptr]),
    'dc_policy_backward': (c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    'dc_gradnorm_clip_adam': (c_int, [c_ptr, c_ptr, c_ptr, c_int, c_int] + [c_ptr] * 11 +
                              [c_flt, c_flt, c_dbl, c_dbl, c_dbl, c_flt, c_ptr]),
}

DC_SINGLE_SCRATCH_FLOATS = 20480      # include/dotaclient_hip.h
ABI_VERSION = 4      # include/dotaclient_hip.h DC_ABI_VERSION: a library built from other sources would mis-call silently

_lib = None


class DotaHipError(RuntimeError):
    pass


def load():
    """Loads the shared library once; raises DotaHipError if it has not been built, is stale, or speaks another ABI version."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch first: both link libamdhip64.  If this library is loaded before torch, the process ends up with two HIP
    # runtimes (the system one behind this library, torch's bundled one behind torch) and every launch from here
    # fails with hipErrorNoDevice - seen with __graft_entry__.build() followed by smoke() in one interpreter.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise DotaHipError('libdotaclient_hip.so not found at %s - run `python -m dotaclient_amd.build` '
                           '(or __graft_entry__.build()); there is no CPU fallback' % LIB_PATH)
    if not os.environ.get('DC_LIB'):
        # stale? the build left the digest of its sources next to the library (build.py); a library older than the sources beside it
        # would run yesterday's kernels under today's tests
        from . import build as _build
        try:
            with open(LIB_PATH + '.sha1') as f:
                stamp = f.read().strip()
        except OSError:
            stamp = None
        if stamp != _build.sources_digest():
            raise DotaHipError('%s is STALE (built from other sources than the ones in dotaclient_amd/csrc: %s) - run `python -m '
                               'dotaclient_amd.build` (or __graft_entry__.build())' % (LIB_PATH, 'no build stamp' if stamp is None else 'digest mismatch'))
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)        # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    got = lib.dc_abi_version()
    if got != ABI_VERSION:
        raise DotaHipError('%s reports ABI version %d, this binding speaks %d - rebuild it (python -m dotaclient_amd.build)'
                           % (LIB_PATH, got, ABI_VERSION))
    _lib = lib
    return lib


def check(code, what=''):
    if code != 0:
        msg = load().dc_last_error().decode('utf-8', 'replace')
        raise DotaHipError('%s failed: %s' % (what or 'dotaclient_hip call', msg))


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def stream_ptr(stream=None):
    import torch
    s = stream if stream is not None else torch.cuda.current_stream()
    return ctypes.c_void_p(s.cuda_stream)
