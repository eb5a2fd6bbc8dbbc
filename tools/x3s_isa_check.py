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
    body = body[:body.index('s_endpgm')].splitlines()
    # the K loop = the depth-2 loop: every basic block whose header comment names it (blocks start at a label or a "; %bb." line)
    m = re.search(r'^\.(LBB\d+_\d+):[^\n]*\n(?:[^\n]*\n)?[^\n]*This Inner Loop Header: Depth=2', '\n'.join(body), re.M)
    if not m:
        print(name[-30:], ': no depth-2 loop found'); bad += 1; continue
    hdr = m.group(1)[1:]
    seg, inside = [], False
    for i, l in enumerate(body):
        if re.match(r'^\.LBB\d+_\d+:', l) or l.startswith('; %bb.'):
            ctx = ' '.join(body[i:i + 3])
            inside = ('Header=' + hdr + ' ') in ctx + ' ' or ('.' + hdr + ':') in l
        if inside:
            seg.append(l)
    n_mfma = sum('v_mfma' in l for l in seg)
    n_dma = sum('global_load_lds' in l for l in seg)
    own, in_asm = [], False
    for l in seg:
        if '#ASMSTART' in l: in_asm = True
        elif '#ASMEND' in l: in_asm = False
        elif 's_waitcnt' in l and 'vmcnt' in l and not in_asm: own.append(l.strip())
    print('%s: K loop %s: %d lines, %d MFMAs, %d DMA pieces, compiler vmcnt waits: %s' % (name[-30:], hdr, len(seg), n_mfma, n_dma, own or 'none'))
    if own or n_mfma < 16:
        bad += 1
sys.exit(1 if bad else 0)
