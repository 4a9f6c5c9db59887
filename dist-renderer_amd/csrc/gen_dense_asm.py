#!/usr/bin/env python
"""Generates distr_dense_asm.hpp: the hand-scheduled k-loop of one dense decoder layer (Y^T = W X^T on
v_mfma_f32_32x32x2_f32) as ONE inline-asm statement per layer shape.

Why not leave it to the compiler (profiles/ubench/mfma_fillers.log, MI355X): beside a back-to-back f32 MFMA stream
  * every VALU instruction costs 8 cycles of MFMA time plus 8 per MFMA gap it sits in (v_mov, v_add, v_cndmask ... all the
    same: the f32 MFMA shares the f32 datapath), v_accvgpr_* cost a whole MFMA latency,
  * every global_load_* costs 8 cycles (64-bit VGPR address),
  * buffer_load with an SGPR offset, ds_read / ds_write, SALU and s_nop are free.
The compiler's loop for distr_mlp.hpp::dense() carries ~11 VALU address / select instructions and 8 global loads per 64 MFMAs:
138.9 k cycles per 512x512 layer against the ideal 131.1 k. The loop below has ONE VALU instruction per 64 MFMAs (the LDS
address step); weights stream through buffer_load_dwordx4 with an SGPR offset, activations through ds_read_b32 with immediate
offsets, the loop counter lives in SGPRs, and the accumulators start from the bias through the MFMA's srcC operand (no
v_accvgpr_write pass).

Data layout = distr_mlp.hpp::dense(): A fragments float4 index ((g*4 + wave)*NOB + ob)*64 + lane, activations X[feature][ray]
(TILE = 64 rays), lane (j, h) reads X[8g + 2s + h][32 rb + j]. Same k order, same MFMA -> bit-identical results.

Register plan (fixed physical registers, listed as clobbers; the accumulators and the cross-layer prefetch tuples are operands):
  v[176:191] A0   v[192:207] A1   (weight fragments of the current / next 8-feature group: 4 row blocks x float4)
  v[208:215] B0   v[216:223] B1   (activation fragments: 4 k-steps x 2 ray blocks)
  v224            running LDS address
  a[128:143], a[144:159], a[160:175]   bias tuples (srcC of the first MFMA of an accumulator)
  s90 loop counter, s91 running weight offset
Usage: python gen_dense_asm.py > distr_dense_asm.hpp   (the header is committed; regenerate after editing this file).
"""
import sys

A = (176, 192)
B = (208, 216)
VX = 224
BIAS = (128, 144, 160)
ACC_VGPR = False     # measured: VGPR accumulators (no v_accvgpr_read in the write-back) change nothing -- the write-back is bound by LDS write bandwidth + its two barriers (2.1 k cycles per layer either way) and only raise register pressure
S_CNT, S_OFF = 's90', 's91'
TILE = 64


def vr(lo, n=1):
    return 'v%d' % lo if n == 1 else 'v[%d:%d]' % (lo, lo + n - 1)


def ar(lo, n):
    return ('v[%d:%d]' if ACC_VGPR else 'a[%d:%d]') % (lo, lo + n - 1)


def gen(K, NOB, NOUT, INIT):
    """INIT: 'bias' (accumulators start from LDS bias tuples) or 'zero'."""
    NG = K // 8
    assert NG % 2 == 0 and NG >= 4
    NIT = (NG - 2) // 2
    GS = 4 * NOB * 1024            # bytes between consecutive 8-feature groups of the fragment stream
    L = []
    emit = L.append

    def acc(ob, rb):
        return '%%[c%d%d]' % (ob, rb)

    def mfma(buf, s, ob, rb, srcc=None):
        a = vr(A[buf] + 4 * ob + s)
        b = vr(B[buf] + 2 * s + rb)
        emit('v_mfma_f32_32x32x2_f32 %s, %s, %s, %s' % (acc(ob, rb), a, b, srcc if srcc is not None else acc(ob, rb)))

    def load_a(buf, ob):
        emit('buffer_load_dwordx4 %s, %%[voff], %%[rs], %s offen offset:%d' % (vr(A[buf] + 4 * ob, 4), S_OFF, ob * 1024))

    def load_b(buf, s, rb, base, goff):
        emit('ds_read_b32 %s, %s offset:%d' % (vr(B[buf] + 2 * s + rb), base, goff * 8 * TILE * 4 + s * 2 * TILE * 4 + rb * 128))

    def group(cur, nxt, base, goff, loads=True, tail_prefetch=False):
        """32/16 MFMAs on buffers `cur`; the loads of the next group (into `nxt`) are interleaved one per MFMA gap."""
        emit('s_waitcnt vmcnt(0) lgkmcnt(0)')
        pend = []
        if loads:
            pend += [('a', ob) for ob in range(NOB)]
            pend += [('b', s, rb) for s in range(4) for rb in range(2)]
            pend += [('soff',)]
        if tail_prefetch:
            pend += [('t', ob) for ob in range(NOUT)]
        for s in range(4):
            for ob in range(NOB):
                for rb in range(2):
                    mfma(cur, s, ob, rb)
                    if pend:
                        p = pend.pop(0)
                        if p[0] == 'a':
                            load_a(nxt, p[1])
                        elif p[0] == 'b':
                            load_b(nxt, p[1], p[2], base, goff)
                        elif p[0] == 'soff':
                            emit('s_add_u32 %s, %s, %d' % (S_OFF, S_OFF, GS))
                        else:
                            emit('buffer_load_dwordx4 %%[t%d], %%[voff], %%[rsn], %%[soffn] offen offset:%d' % (p[1], p[1] * 1024))
        assert not pend

    # ---------------------------------------------------------------- prologue + peeled group 0
    # LDS operations complete in order; `lgkm` mirrors the outstanding queue so that every wait names exactly what it needs
    # (at most 15 may be outstanding: the counter has 4 bits)
    lgkm = []

    def lds(text, tag):
        emit(text)
        lgkm.append(tag)
        assert len(lgkm) <= 15, 'too many LDS operations outstanding'

    def wait_for(*tags):           # in-order completion: waiting for the youngest of `tags` covers all of them
        idxs = [i for i, t in enumerate(lgkm) if t in tags]
        if not idxs:
            return
        idx = max(idxs)
        emit('s_waitcnt lgkmcnt(%d)' % (len(lgkm) - 1 - idx))
        del lgkm[:idx + 1]

    def roundtrip(ob):             # this layer's first group arrived in the operand tuples: move it to A0 through LDS
        lds('ds_write_b128 %%[scr], %%[t%d]' % ob, 'rt%d' % ob)
        lds('ds_read_b128 %s, %%[scr]' % vr(A[0] + 4 * ob, 4), 'rt%d' % ob)

    bias = INIT == 'bias'
    slot_of = {0: 0, 1: 1, 2: 2, 3: 0}      # bias tuple of row block ob (tuple 0 is reloaded for ob 3 after ob 1's MFMAs were issued)

    def bias_tuple(ob):
        if bias:
            for q in range(4):
                lds('ds_read_b128 %s, %%[bias] offset:%d' % (ar(BIAS[slot_of[ob]] + 4 * q, 4), ob * 128 + q * 32), 'bias%d' % ob)

    def b0(s):
        for rb in range(2):
            lds('ds_read_b32 %s, %%[xaddr] offset:%d' % (vr(B[0] + 2 * s + rb), s * 2 * TILE * 4 + rb * 128), 'b0s%d' % s)

    def first_mfmas(ob):           # s = 0: the accumulators of row block ob start from the bias tuple (or 0)
        wait_for('rt%d' % ob, 'b0s0', 'bias%d' % ob)
        for rb in range(2):
            mfma(0, 0, ob, rb, ar(BIAS[slot_of[ob]], 16) if bias else '0')

    emit('s_nop 4')
    emit('s_add_u32 %s, %%[soff], %d' % (S_OFF, GS))
    for ob in range(NOB):          # group 1's weights: on their way during the whole of group 0
        load_a(1, ob)
    emit('s_add_u32 %s, %s, %d' % (S_OFF, S_OFF, GS))
    emit('s_mov_b32 %s, %d' % (S_CNT, NIT))
    roundtrip(0); b0(0); bias_tuple(0)
    roundtrip(1); bias_tuple(1)
    emit('v_add_u32 %s, %d, %%[xaddr]' % (vr(VX), 8 * TILE * 4))
    first_mfmas(0)
    if NOB == 4:
        roundtrip(2); bias_tuple(2)
        first_mfmas(1)
        roundtrip(3); bias_tuple(3)
        first_mfmas(2)
        b0(1); b0(2); b0(3)
        first_mfmas(3)
    else:
        b0(1); b0(2); b0(3)
        first_mfmas(1)
    pend = [(ps, prb) for ps in range(4) for prb in range(2)]
    for s in range(1, 4):
        wait_for('b0s%d' % s)
        for ob in range(NOB):
            for rb in range(2):
                mfma(0, s, ob, rb)
                if pend and s >= 2:
                    ps, prb = pend.pop(0)
                    lds('ds_read_b32 %s, %%[xaddr] offset:%d' % (vr(B[1] + 2 * ps + prb), 8 * TILE * 4 + ps * 2 * TILE * 4 + prb * 128), 'b1')
    assert not pend
    # ---------------------------------------------------------------- main loop: groups 1 .. NG-2, two per iteration
    emit('.Ldense_loop_%=:')
    group(1, 0, vr(VX), 1)
    group(0, 1, vr(VX), 2)
    emit('v_add_u32 %s, %d, %s' % (vr(VX), 2 * 8 * TILE * 4, vr(VX)))
    emit('s_sub_u32 %s, %s, 1' % (S_CNT, S_CNT))
    emit('s_cmp_lg_u32 %s, 0' % S_CNT)
    emit('s_cbranch_scc1 .Ldense_loop_%=')
    # ---------------------------------------------------------------- peeled last group (+ the next layer's first fragments)
    group(1, 0, vr(VX), 0, loads=False, tail_prefetch=NOUT > 0)
    emit('s_waitcnt vmcnt(0)')
    emit('s_nop 15')
    emit('s_nop 3')
    body = '\n'.join('      "%s\\n"' % x for x in L)

    name = 'dense_asm_k%d_n%d_o%d_%s' % (K, NOB, NOUT, INIT)
    outs = ', '.join('[c%d%d] "=&%s"(acc[%d][%d])' % (ob, rb, 'v' if ACC_VGPR else 'a', ob, rb) for ob in range(NOB) for rb in range(2))
    outs += ', ' + ', '.join('[t%d] "+v"(t[%d])' % (i, i) for i in range(4))
    ins = '[xaddr] "v"(xaddr), [voff] "v"(voff), [rs] "s"(rs), [rsn] "s"(rsn), [soff] "s"(soff), [soffn] "s"(soffn), [bias] "v"(biasaddr), [scr] "v"(scratch)'
    clob = ['"v%d"' % i for i in range(A[0], VX + 1)] + ['"%s%d"' % ('v' if ACC_VGPR else 'a', i) for i in range(BIAS[0], BIAS[2] + 16)] + ['"%s"' % S_CNT, '"%s"' % S_OFF, '"scc"', '"memory"']
    return name, '''// K = %d input features, %d row blocks of 32 per wave, fetches %d fragments of the next layer's first group, accumulators start from %s
__device__ __forceinline__ void %s(f32x16 (&acc)[%d][2], f32x4 (&t)[4], uint32_t xaddr, uint32_t voff, rsrc_t rs, rsrc_t rsn,
    uint32_t soff, uint32_t soffn, uint32_t biasaddr, uint32_t scratch) {
  asm volatile(
%s
      : %s
      : %s
      : %s);
}
''' % (K, NOB, NOUT, 'the LDS bias tuples' if INIT == 'bias' else 'zero', name, NOB, body, outs, ins, ', '.join(clob))


# =====================================================================================================================
# 16-ray tile: v_mfma_f32_16x16x4_f32 (32 cycles), the latency-bound tail of the march. A VALU instruction costs the same 8 + 8
# cycles beside it, i.e. twice as much relative to the MFMA, so the hand-scheduled loop pays off more here. Layout =
# distr_mlp.hpp::dense16(): A fragments float4 index ((g*4 + wave)*NB + ob)*64 + lane (g = group of 16 features), activations
# X[feature][16 rays], lane (j, kq) reads X[16g + 4s + kq][j] = byte offset lane*4 + (16g + 4s)*64.
# Fixed registers: v[100:131] A0, v[132:163] A1 (NB x float4), v[164:167] B0, v[168:171] B1, v172 running LDS address.
# The accumulators are in/out operands initialised by the caller (bias), the next layer's first group travels in t[0..NB).
A16 = (100, 132)
B16 = (164, 168)
VX16 = 172


def gen16(K, NB, NOUT):
    NG = K // 16
    assert NG % 2 == 0 and NG >= 4
    NIT = (NG - 2) // 2
    GS = 4 * NB * 1024
    L = []
    emit = L.append

    def mfma(buf, s, ob):
        emit('v_mfma_f32_16x16x4_f32 %%[c%d], %s, %s, %%[c%d]' % (ob, vr(A16[buf] + 4 * ob + s), vr(B16[buf] + s), ob))

    S_OFF2, S_OFFN2 = 's92', 's93'        # the 12-bit immediate reaches 4 fragments; the upper four go through a second SGPR offset

    def load_a(buf, ob):
        emit('buffer_load_dwordx4 %s, %%[voff], %%[rs], %s offen offset:%d' % (vr(A16[buf] + 4 * ob, 4), S_OFF if ob < 4 else S_OFF2, (ob % 4) * 1024))

    def bump():
        emit('s_add_u32 %s, %s, %d' % (S_OFF, S_OFF, GS))
        emit('s_add_u32 %s, %s, 4096' % (S_OFF2, S_OFF))

    def load_b(buf, s, base, goff):
        emit('ds_read_b32 %s, %s offset:%d' % (vr(B16[buf] + s), base, goff * 1024 + s * 256))

    def group(cur, nxt, base, goff, loads=True, tail_prefetch=False):
        emit('s_waitcnt vmcnt(0) lgkmcnt(0)')
        pend = []
        if loads:
            pend += [('a', ob) for ob in range(NB)] + [('b', s) for s in range(4)] + [('soff',)]
        if tail_prefetch:
            pend += [('t', ob) for ob in range(NOUT)]
        for s in range(4):
            for ob in range(NB):
                mfma(cur, s, ob)
                if pend:
                    p = pend.pop(0)
                    if p[0] == 'a':
                        load_a(nxt, p[1])
                    elif p[0] == 'b':
                        load_b(nxt, p[1], base, goff)
                    elif p[0] == 'soff':
                        bump()
                    else:
                        emit('buffer_load_dwordx4 %%[t%d], %%[voff], %%[rsn], %s offen offset:%d' % (p[1], '%[soffn]' if p[1] < 4 else S_OFFN2, (p[1] % 4) * 1024))
        assert not pend

    # prologue: group 1's weights first (longest latency), then this layer's first group from the operand tuples through LDS
    emit('s_nop 4')
    emit('s_add_u32 %s, %%[soff], %d' % (S_OFF, GS))
    emit('s_add_u32 %s, %s, 4096' % (S_OFF2, S_OFF))
    emit('s_add_u32 %s, %%[soffn], 4096' % S_OFFN2)
    for ob in range(NB):
        load_a(1, ob)
    bump()
    emit('s_mov_b32 %s, %d' % (S_CNT, NIT))
    for s in range(4):
        load_b(0, s, '%[xaddr]', 0)
    for ob in range(NB):
        emit('ds_write_b128 %%[scr], %%[t%d]' % ob)
        emit('ds_read_b128 %s, %%[scr]' % vr(A16[0] + 4 * ob, 4))
        if ob % 4 == 3 and ob + 1 < NB:
            emit('s_waitcnt lgkmcnt(0)')                   # keep the LGKM queue below its 15 entries
    emit('v_add_u32 %s, 1024, %%[xaddr]' % vr(VX16))
    emit('s_waitcnt lgkmcnt(0)')
    pend = list(range(4))
    for s in range(4):                                     # peeled group 0 (B1 of group 1 is read on the way)
        for ob in range(NB):
            mfma(0, s, ob)
            if pend and s >= 1:
                load_b(1, pend.pop(0), '%[xaddr]', 1)
    assert not pend
    emit('.Ldense16_loop_%=:')
    group(1, 0, vr(VX16), 1)
    group(0, 1, vr(VX16), 2)
    emit('v_add_u32 %s, 2048, %s' % (vr(VX16), vr(VX16)))
    emit('s_sub_u32 %s, %s, 1' % (S_CNT, S_CNT))
    emit('s_cmp_lg_u32 %s, 0' % S_CNT)
    emit('s_cbranch_scc1 .Ldense16_loop_%=')
    group(1, 0, vr(VX16), 0, loads=False, tail_prefetch=NOUT > 0)
    emit('s_waitcnt vmcnt(0)')
    emit('s_nop 7')
    emit('s_nop 3')
    body = '\n'.join('      "%s\\n"' % x for x in L)
    name = 'dense16_asm_k%d_n%d_o%d' % (K, NB, NOUT)
    outs = ', '.join('[c%d] "+v"(acc[%d])' % (ob, ob) for ob in range(NB))
    outs += ', ' + ', '.join('[t%d] "+v"(t[%d])' % (i, i) for i in range(8))
    ins = '[xaddr] "v"(xaddr), [voff] "v"(voff), [rs] "s"(rs), [rsn] "s"(rsn), [soff] "s"(soff), [soffn] "s"(soffn), [scr] "v"(scratch)'
    clob = ['"v%d"' % i for i in range(A16[0], VX16 + 1)] + ['"%s"' % S_CNT, '"%s"' % S_OFF, '"s92"', '"s93"', '"scc"', '"memory"']
    return name, '''// 16-ray tile: K = %d input features, %d row blocks of 16 per wave, fetches %d fragments of the next layer's first group
__device__ __forceinline__ void %s(f32x4 (&acc)[%d], f32x4 (&t)[8], uint32_t xaddr, uint32_t voff, rsrc_t rs, rsrc_t rsn,
    uint32_t soff, uint32_t soffn, uint32_t scratch) {
  asm volatile(
%s
      : %s
      : %s
      : %s);
}
''' % (K, NB, NOUT, name, NB, body, outs, ins, ', '.join(clob))


def main():
    out = ['// GENERATED by gen_dense_asm.py -- do not edit; see that file for the design notes.',
           '#pragma once', '#include <hip/hip_runtime.h>', '#include <stdint.h>', '', 'namespace distr {', '',
           'typedef float f32x16 __attribute__((ext_vector_type(16)));', 'typedef float f32x4 __attribute__((ext_vector_type(4)));',
           'typedef __amdgpu_buffer_rsrc_t rsrc_t;', '']
    for init in ('bias', 'zero'):
        for (K, NOB, NOUT) in ((512, 4, 4), (512, 4, 2), (512, 2, 4), (256, 4, 4), (512, 4, 0)):
            out.append(gen(K, NOB, NOUT, init)[1])
    for (K, NB, NOUT) in ((512, 8, 8), (512, 8, 4), (512, 4, 8), (256, 8, 8), (512, 8, 0)):
        out.append(gen16(K, NB, NOUT)[1])
    out.append('}  // namespace distr')
    sys.stdout.write('\n'.join(out) + '\n')


if __name__ == '__main__':
    main()
