"""Generates the hand-scheduled unit statements of the 8-member cluster tile (the CL8_*_TEXT macros of distr_mlp.hpp, between
`#define CL8_A_TEXT` and `#define CL8_LOADS_TEXT`): a unit = 8 k-groups = 32 dependent v_mfma_f32_16x16x4_f32 of a wave's one 16-row block.

    python gen_cl8_units.py > /tmp/cl8_text.inc        # paste over the macro block (the header holds the generated text itself)
    python gen_cl8_units.py --check distr_mlp.hpp      # exit 1 if the header's block differs from the generator's output

Schedule (see the comment above the macros in distr_mlp.hpp): B fragments in four fixed 4-register buffers v[208:223], read two k-groups
ahead with ds_read2st64 from the fixed LDS address v207; lgkmcnt(4) in front of every group; the eight requests of the weight chunk three
units on behind the first two MFMAs of groups 0..3; AS variants: the staging of the next input unit behind MFMAs 17..22.
"""
import sys
# named operands:
#   b0, b1 (s64: request bases), ab (n: accumulator block), rb (n: ring slot of this unit), nw (n: vmcnt for this unit's chunk), nrb (n: ring slot requested),
#   uo (n: 32 * unit: ds offsets), staging: nws (n: vmcnt for the next unit's granules), land (n), tag (s32), m0, m1 (s64 exec masks of the two
#   entries), so (s32: LDS offset of the wave's first row block of the unit), flag (=s64), ex (=&s64)
def mf(k, b): return '"v_mfma_f32_16x16x4_f32 a[%%[ab]:%%[ab]+3], a[%%[rb]+%d], v%d, a[%%[ab]:%%[ab]+3]\\n\\t"' % (k, b)
def rd(dst, off): return '"ds_read2st64_b32 v[%d:%d], v207 offset0:%%[uo]+%d offset1:%%[uo]+%d\\n\\t"' % (dst, dst + 1, off, off + 1)
def wait(n): return '"s_waitcnt lgkmcnt(%d)\\n\\t"' % n
BUF = [208, 212, 216, 220]
def mask(E): return '"s_bitcmp1_b32 %%[own], %d\\n\\ts_cselect_b64 exec, 0, -1\\n\\t"' % E     # exec = 0 when entry E holds the member's own rows
FULL = '"s_mov_b64 exec, -1\\n\\t"'
def sads(E):
    o = 8 * E
    return [mask(E),
            '"v_sad_u8 v[%%[land]+%d], v[%%[land]+%d], %%[tag], 0\\n\\t"' % (o + 1, o + 1),
            '"v_sad_u8 v[%%[land]+%d], v[%%[land]+%d], %%[tag], v[%%[land]+%d]\\n\\t"' % (o + 3, o + 3, o + 1),
            '"v_sad_u8 v[%%[land]+%d], v[%%[land]+%d], %%[tag], v[%%[land]+%d]\\n\\t"' % (o + 5, o + 5, o + 3),
            '"v_sad_u8 v[%%[land]+%d], v[%%[land]+%d], %%[tag], v[%%[land]+%d]\\n\\t"' % (o + 7, o + 7, o + 5), FULL]
def stores():
    return ['"v_add_u32 v[%[land]+1], %[so0], v206\\n\\t"', mask(0),
            '"ds_write2_b32 v[%[land]+1], v[%[land]+0], v[%[land]+2] offset1:16\\n\\t"',
            '"ds_write2_b32 v[%[land]+1], v[%[land]+4], v[%[land]+6] offset0:32 offset1:48\\n\\t"', mask(1),
            '"ds_write_b32 v[%[land]+1], v[%[land]+8] offset:4096\\n\\t"', '"ds_write_b32 v[%[land]+1], v[%[land]+10] offset:4160\\n\\t"',
            '"ds_write_b32 v[%[land]+1], v[%[land]+12] offset:4224\\n\\t"', '"ds_write_b32 v[%[land]+1], v[%[land]+14] offset:4288\\n\\t"', FULL]
def stmt_a(loads, stage):
    L = []
    if loads: L.append('"s_nop 4\\n\\t"')
    L.append('"s_waitcnt vmcnt(%[nw])\\n\\t"')
    li = 0
    for g in range(6):
        nb = BUF[(g + 2) % 4]
        L += [rd(nb, 4 * (g + 2)), rd(nb + 2, 4 * (g + 2) + 2), wait(4)]
        b = BUF[g % 4]
        for s in range(4):
            L.append(mf(4 * g + s, b + s))
            if loads and g < 4 and s < 2:
                L.append('CL8_L%d' % li); li += 1
            k = 4 * g + s
            if stage and k == 17: L += ['"s_waitcnt vmcnt(%[nws])\\n\\t"'] + sads(0)
            if stage and k == 18: L += sads(1)
            if stage and k == 19: L += stores()
            if stage and k == 20: L += [mask(0), '"v_cmp_ne_u32 vcc, 0, v[%[land]+7]\\n\\t"', FULL]
            if stage and k == 21: L += ['"s_mov_b64 %[flag], vcc\\n\\t"', mask(1), '"v_cmp_ne_u32 vcc, 0, v[%[land]+15]\\n\\t"', FULL]
            if stage and k == 22: L += ['"s_or_b64 %[flag], %[flag], vcc\\n\\t"']
    return L
def emit(name, L):
    return '#define %s \\\n' % name + ' \\\n'.join('  ' + x for x in L) + '\n'
LS = '(CL8_L0, CL8_L1, CL8_L2, CL8_L3, CL8_L4, CL8_L5, CL8_L6, CL8_L7)'
out = []
out.append(emit('CL8_A_TEXT' + LS, stmt_a(True, False)))
out.append(emit('CL8_AS_TEXT' + LS, stmt_a(True, True)))
out.append(emit('CL8_A0_TEXT', stmt_a(False, False)))
out.append(emit('CL8_A0S_TEXT', stmt_a(False, True)))
L = []
for g in (6, 7):
    nb = BUF[(g + 2) % 4]
    L += [rd(nb, 4 * (g + 2)), rd(nb + 2, 4 * (g + 2) + 2), wait(4)]
    for s in range(4): L.append(mf(4 * g + s, BUF[g % 4] + s))
out.append(emit('CL8_B_NEXT_TEXT', L))
L = [wait(2)]
for s in range(4): L.append(mf(24 + s, BUF[2] + s))
L.append(wait(0))
for s in range(4): L.append(mf(28 + s, BUF[3] + s))
out.append(emit('CL8_B_LAST_TEXT', L))
text = '\n'.join(out)
if len(sys.argv) > 2 and sys.argv[1] == '--check':
    h = open(sys.argv[2]).read()
    a, b = h.index('#define CL8_A_TEXT('), h.index('#define CL8_LOADS_TEXT(')
    same = h[a:b] == text
    print('distr_mlp.hpp: CL8_*_TEXT block %s the generator output' % ('equals' if same else 'DIFFERS from'))
    sys.exit(0 if same else 1)
sys.stdout.write(text)
