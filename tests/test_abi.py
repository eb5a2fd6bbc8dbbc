"""CPU: the C-ABI library loads and exports every symbol include/dotaclient_hip.h declares."""
import os
import re

import pytest

from dotaclient_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(REPO, 'include', 'dotaclient_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(dc_[a-z0-9_]+)\s*\(', txt)) - {'dc_stream_t'})


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    syms = _declared_symbols()
    assert len(syms) >= 3
    for s in syms:
        assert hasattr(lib, s), s
        assert s in _lib.SIGNATURES, 'ctypes signature missing for ' + s
    assert sorted(_lib.SIGNATURES) == syms
    hdr = open(os.path.join(REPO, "include", "dotaclient_hip.h")).read()
    assert lib.dc_abi_version() == int(re.search(r"#define DC_ABI_VERSION (\d+)", hdr).group(1)) == _lib.ABI_VERSION


def test_every_declaration_cites_the_reference():
    txt = open(os.path.join(REPO, 'include', 'dotaclient_hip.h')).read()
    assert txt.count('optimizer.py:') + txt.count('policy.py:') >= 3


def test_loader_refuses_a_stale_library(tmp_path, monkeypatch):
    # build.py leaves the digest of the sources it compiled next to the library; the loader compares it with the sources it finds and
    # refuses a library built from other ones (build.py's docstring promised this, the loader only checked the ABI number)
    import shutil
    from dotaclient_amd import _lib, build
    assert open(_lib.LIB_PATH + '.sha1').read().strip() == build.sources_digest(), 'rebuild: python -m dotaclient_amd.build'
    monkeypatch.delenv('DC_LIB', raising=False)
    fake = tmp_path / 'libdotaclient_hip.so'
    shutil.copy(_lib.LIB_PATH, fake)
    (tmp_path / 'libdotaclient_hip.so.sha1').write_text('0' * 40)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(fake))
    monkeypatch.setattr(_lib, '_lib', None)
    with pytest.raises(_lib.DotaHipError, match='STALE'):
        _lib.load()
    (tmp_path / 'libdotaclient_hip.so.sha1').unlink()
    with pytest.raises(_lib.DotaHipError, match='no build stamp'):
        _lib.load()


def test_gemm_x3s_k_loop_carries_no_compiler_vmcnt_wait():
    # csrc/gemm_x3s.hip counts its LDS-DMA by hand (inline asm, vmcnt(6) across the barriers).  Twice while it was written hipcc put a
    # vmcnt(0) of its own into the K loop - in front of a ds_read behind a builtin LDS-DMA, and in the loop header for a load whose use a
    # divergent branch skipped - and drained the pipeline every stage: results right, kernel 1.5x slower.  tools/x3s_isa_check.py compiles
    # the file and inspects the basic blocks that issue MFMAs / fragment reads / DMA pieces.
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(REPO, 'tools', 'x3s_isa_check.py')], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count('compiler vmcnt waits: none') == 6, r.stdout[-3000:]       # PARTIAL x {plain, relu, mask}
