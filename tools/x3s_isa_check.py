"""Developer check for csrc/gemm_x3s.hip: compiles it, cuts the K loop of each instantiation out of the ISA and fails if hipcc put a vmcnt wait
of its own into it (the loop's DMA is counted by hand in asm statements; a compiler-inserted vmcnt(0) there drains the pipeline every stage -
it has happened twice while the kernel was written: a ds_read behind a builtin LDS-DMA, a load whose use a divergent branch skipped)."""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(REPO, 'dotaclient_amd', 'csrc', 'gemm_x3s.hip')
with tempfile.TemporaryDirectory() as d:
    r = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-c', src, '-o', os.path.join(d, 'x.o'), '-save-temps',
                        '-Rpass-analysis=kernel-resource-usage'] + sys.argv[1:], cwd=d, capture_output=True, text=True)
    for l in r.stderr.splitlines():
        if re.search(r'Function Name|VGPRs:|Spill|Occupancy', l):
            print(l.split('remark: ')[-1].split(' [-R')[0].split(':0:')[-1].strip())
    asm = open(os.path.join(d, 'gemm_x3s-hip-amdgcn-amd-amdhsa-gfx950.s')).read()
bad = 0
for name in re.findall(r'^(_ZN2dc\S*gemm_x3s_kernel\S*):', asm, re.M):
    body = asm[asm.index(name + ':'):]
    body = body[:body.index('.Lfunc_end')].splitlines()
    # every basic block (it starts at a label or a "; %bb." line) that issues MFMAs, fragment reads or DMA pieces - the hot path of the K
    # loop; the epilogue's blocks (global loads / stores only) may wait as hipcc sees fit
    blocks, cur = [], []
    for l in body:
        if re.match(r'^\.LBB\d+_\d+:', l) or l.startswith('; %bb.'):
            blocks.append(cur); cur = []
        cur.append(l)
    blocks.append(cur)
    n_mfma = n_dma = n_hot = 0
    own = []
    for b in blocks:
        hot = any(('v_mfma' in l or 'ds_read_b128' in l or 'global_load_lds' in l) for l in b)
        if not hot or not any('Loop' in l for l in b[:4]):
            continue
        n_hot += 1
        n_mfma += sum('v_mfma' in l for l in b); n_dma += sum('global_load_lds' in l for l in b)
        in_asm = False
        for l in b:
            if '#ASMSTART' in l: in_asm = True
            elif '#ASMEND' in l: in_asm = False
            elif 's_waitcnt' in l and 'vmcnt' in l and not in_asm: own.append(l.strip())
    print('%s: %d hot blocks inside loops: %d MFMAs, %d DMA pieces, compiler vmcnt waits: %s' % (name[-30:], n_hot, n_mfma, n_dma, own or 'none'))
    if own or n_mfma < 24:
        bad += 1
sys.exit(1 if bad else 0)
