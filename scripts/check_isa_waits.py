"""Dev lint: compile the conv translation units to gfx950 assembly and flag any s_waitcnt vmcnt(...) the compiler
placed directly in front of a K-loop fragment-read cluster (>= 8 ds_read_b128) of an igemm_stagger_kernel or
igemm_wide_kernel.

Why: the staggered kernel keeps two K-tiles of LDS-DMA in flight with counted waits of its own.  Whether the compiler
adds a draining `s_waitcnt vmcnt(0)` before the fragment reads depends on its register scoreboard at the loop header
(a global load consumed only under a condition leaves "maybe pending" registers; re-using one of them inside the loop
forces the wait) -- it appeared and disappeared with unrelated edits during round 1.  Run after touching
igemm_stagger.hip.h / igemm_wide.hip.h / igemm.hip.h epilogues:   python scripts/check_isa_waits.py
Also checks the wgrad_patch.hip.h kernels (no scratch access between their MFMAs, accumulation in place).
"""
import os, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'recurrent_gaze_prediction_amd', 'csrc')


def kernels(asm):
    lines = asm.split('\n')
    for st, l in enumerate(lines):
        if (l.startswith('_ZN3rgp20igemm_stagger_kernel') or l.startswith('_ZN3rgp17igemm_wide_kernel')) and '@' in l:
            end = next(i for i in range(st, len(lines)) if 's_endpgm' in lines[i])
            yield l.split(':')[0], lines[st:end]


def stray_waits(body):
    i, bad = 0, []
    while i < len(body):
        if 'ds_read_b128' in body[i]:
            j = i
            while j < len(body) and ('ds_read_b128' in body[j] or 'v_add_u32' in body[j]):
                j += 1
            n = sum('ds_read_b128' in l for l in body[i:j])
            pre = [l.strip() for l in body[max(0, i - 14):i] if 'vmcnt' in l]
            if n >= 8 and pre:
                bad.append((i, pre))
            i = j
        else:
            i += 1
    return bad


def wgrad_patch_kernels(asm):
    lines = asm.split('\n')
    for st, l in enumerate(lines):
        if l.startswith('_ZN3rgpL23wgrad_patch_bf16_kernel') and '@' in l:
            end = next(i for i in range(st, len(lines)) if 's_endpgm' in lines[i])
            yield l.split(':')[0], lines[st:end]


def wgrad_patch_faults(body):
    """wgrad_patch.hip.h: between the first and the last MFMA there must be no scratch access (a spill inside the group
    loop waits on vmcnt and with it on the slab DMA in flight: +10 % when it happened), every MFMA must accumulate in
    place, and no compiler-inserted vmcnt wait may sit between two MFMAs of a step (the kernel's own are vmcnt(0) in
    front of a barrier)."""
    mf = [i for i, l in enumerate(body) if 'v_mfma' in l]
    faults = []
    if not mf:
        return ['no MFMA found']
    loop = body[mf[0]:mf[-1] + 1]
    if any('scratch_' in l for l in loop):
        faults.append('scratch access inside the MFMA region')
    for l in loop:
        if 'v_mfma' in l:
            ops = [o.strip() for o in l.split('v_mfma_f32_16x16x32_bf16')[1].split(',')]
            if ops[0] != ops[3]:
                faults.append('MFMA not in place: ' + l.strip())
                break
    return faults


def patch_kernels(asm):
    lines = asm.split('\n')
    for st, l in enumerate(lines):
        if l.startswith('_ZN3rgpL') and 'conv_patch' in l and '@' in l:
            end = next(i for i in range(st, len(lines)) if 's_endpgm' in lines[i])
            yield l.split(':')[0], lines[st:end]


def patch_faults(name, body):
    """conv_patch*.hip.h: no scratch access between the first and the last MFMA (a reload there waits on vmcnt and with it
    on the LDS-DMA ring), and -- inference kernels (no arg-max codes) -- none behind the last MFMA either: the round-3
    kernels lost 2-4 % to epilogue-only values the compiler had hoisted out of the tile loop and spilled across the K loop
    (each reload drained the look-ahead DMA once per tile)."""
    mf = [i for i, l in enumerate(body) if 'v_mfma' in l]
    sc = [i for i, l in enumerate(body) if 'scratch_load' in l]
    faults = []
    if any(mf[0] <= i <= mf[-1] for i in sc):
        faults.append('scratch reload inside the MFMA region')
    training = 'Lb1ELb1E' in name                      # <..., POOL = true, ARGMAX = true, ...>
    if not training and any(i > mf[-1] for i in sc):
        faults.append('scratch reload in the epilogue')
    return faults


def main():
    rc = 0
    with tempfile.NamedTemporaryFile(suffix='.s') as tf:
        subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-I' + os.path.join(ROOT, 'include'),
                        '-I' + CSRC, '-S', '--cuda-device-only', '-o', tf.name, os.path.join(CSRC, 'rgp_conv_patch.hip')], check=True,
                       stderr=subprocess.DEVNULL)
        asm = open(tf.name).read()
    n = nbad = 0
    for name, body in patch_kernels(asm):
        n += 1
        faults = patch_faults(name, body)
        if faults:
            nbad += 1
            print('CONV_PATCH', name[:100], faults)
    print('rgp_conv_patch.hip: %d patch kernels, %d with scratch reloads in the K loop / an inference epilogue' % (n, nbad))
    rc |= nbad > 0
    for tu in ('rgp_c3d.hip', 'rgp_c3d_bwd.hip'):
        with tempfile.NamedTemporaryFile(suffix='.s') as tf:
            subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-I' + os.path.join(ROOT, 'include'),
                            '-I' + CSRC, '-S', '--cuda-device-only', '-o', tf.name, os.path.join(CSRC, tu)], check=True,
                           stderr=subprocess.DEVNULL)
            asm = open(tf.name).read()
        n = nbad = 0
        for name, body in kernels(asm):
            n += 1
            bad = stray_waits(body)
            if bad:
                nbad += 1
                print('STRAY WAIT', tu, name[:90], bad[:2])
        print('%s: %d staggered / wide kernels, %d with a compiler wait in front of the fragment reads' % (tu, n, nbad))
        rc |= nbad > 0
        nw = nwbad = 0
        for name, body in wgrad_patch_kernels(asm):
            nw += 1
            faults = wgrad_patch_faults(body)
            if faults:
                nwbad += 1
                print('WGRAD_PATCH', name[:90], faults)
        if nw:
            print('%s: %d wgrad_patch kernels, %d with spills in the loop or renamed accumulators' % (tu, nw, nwbad))
        rc |= nwbad > 0
    sys.exit(rc)


if __name__ == '__main__':
    main()
