"""Builds the oracle's C restatement (gcc) into oracle/_build/.  Test infrastructure only."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, '_build')
LIB = os.path.join(OUT, 'libgae_ref.so')


def build():
    os.makedirs(OUT, exist_ok=True)
    src = os.path.join(HERE, 'gae_ref.c')
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.check_call(['gcc', '-O2', '-ffp-contract=off', '-shared', '-fPIC', src, '-o', LIB])
    return LIB


def gae_ref(rewards, values, off, lens, gamma=0.98, lam=0.97):
    """numpy front-end of gae_ref.c."""
    import ctypes
    import numpy as np
    lib = ctypes.CDLL(build())
    rewards = np.ascontiguousarray(rewards, np.float32)
    values = np.ascontiguousarray(values, np.float32)
    off = np.ascontiguousarray(off, np.int64)
    lens = np.ascontiguousarray(lens, np.int32)
    adv = np.empty(values.shape[0], np.float32)
    ret = np.empty(values.shape[0], np.float32)
    P = ctypes.c_void_p
    lib.gae_ref.argtypes = [P, P, P, P, ctypes.c_int, ctypes.c_double, ctypes.c_double, P, P]
    lib.gae_ref.restype = None
    lib.gae_ref(rewards.ctypes.data, values.ctypes.data, off.ctypes.data, lens.ctypes.data, len(lens),
                gamma, lam, adv.ctypes.data, ret.ctypes.data)
    return adv, ret


if __name__ == '__main__':
    print(build())
