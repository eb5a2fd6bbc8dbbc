"""Builds libdotaclient_hip.so (hipcc, --offload-arch=gfx950) in-tree next to the sources.

The .so is git-ignored but travels to the GPU box with the gpurun snapshot.  There is no JIT and no
fallback: `dotaclient_amd._lib` refuses to load a missing library, one of another ABI version, or a STALE one - the
build leaves the digest of every source it compiled next to the library (`<lib>.sha1`), and the loader compares it with
the sources it finds (A/B builds loaded through DC_LIB are exempt: they are built with other flags on purpose).
"""
import concurrent.futures
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
# A/B builds: DC_BUILD_VARIANT=name (+ DC_BUILD_FLAGS='-DX ...') produces libdotaclient_hip_name.so next to the regular
# library (own object directory); run with DC_LIB=<that file> to load it (dotaclient_amd/_lib.py).
VARIANT = os.environ.get('DC_BUILD_VARIANT', '')
OBJ = os.path.join(CSRC, '_obj' + ('_' + VARIANT if VARIANT else ''))
LIB = os.path.join(HERE, 'libdotaclient_hip%s.so' % ('_' + VARIANT if VARIANT else ''))
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function'] + os.environ.get('DC_BUILD_FLAGS', '').split()


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith('.hip'))


def _digest(paths):
    h = hashlib.sha1()
    for p in paths:
        with open(p, 'rb') as f:
            h.update(f.read())
    h.update(' '.join(FLAGS).encode())
    return h.hexdigest()


def sources_digest():
    """Digest of everything the default library is built from (sources, headers, the C ABI header, the default flags)."""
    paths = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(('.hip', '.h'))]
    paths.append(os.path.join(HERE, '..', 'include', 'dotaclient_hip.h'))
    h = hashlib.sha1()
    for p in paths:
        with open(p, 'rb') as f:
            h.update(f.read())
    return h.hexdigest()


def _compile_one(src):
    deps = [os.path.join(CSRC, src)] + [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith('.h')]
    deps.append(os.path.join(HERE, '..', 'include', 'dotaclient_hip.h'))
    tag = _digest(deps)
    obj = os.path.join(OBJ, src[:-4] + '.o')
    stamp = obj + '.sha1'
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == tag:
        return obj, False
    cmd = [HIPCC] + FLAGS + ['-c', os.path.join(CSRC, src), '-o', obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('hipcc failed for %s:\n%s\n%s' % (src, r.stdout, r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    with open(stamp, 'w') as f:
        f.write(tag)
    return obj, True


def build_library(verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    srcs = _sources()
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        results = list(ex.map(_compile_one, srcs))
    objs = [o for o, _ in results]
    rebuilt = any(c for _, c in results)
    if rebuilt or not os.path.exists(LIB):
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n%s\n%s' % (r.stdout, r.stderr))
    with open(LIB + '.sha1', 'w') as f:       # what the loader checks the sources against (dotaclient_amd/_lib.py)
        f.write(sources_digest() + ('' if not (VARIANT or os.environ.get('DC_BUILD_FLAGS')) else ' variant'))
    if verbose:
        print('libdotaclient_hip.so: %s (%d sources, %s)' % (LIB, len(srcs), 'rebuilt' if rebuilt else 'up to date'))
    return LIB


if __name__ == '__main__':
    build_library()
