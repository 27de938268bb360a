#!/usr/bin/env python3
"""Generates tools/attn_round_asm.inc: the key-tile rounds of the persistent spatial-attention forward (tools/attn_fwd_p4.hip) with EVERY
instruction of the hot path as its own `asm volatile` statement.

Why: hipcc orders `asm volatile` statements as written and nothing else -- the sched_barrier / sched_group_barrier forms of the round left the
MFMA and softmax streams overlapping by only ~380 of 1 048 cycles (profiles/r05_attn_fwd_p4.txt).  Here the source order IS the issue order:
one MFMA, then ~9 vector instructions of ANOTHER query tile's softmax, and so on; register allocation stays with the compiler (operands are
ordinary C++ values; elements of the S accumulator tuple are read as sub-registers, never written).

What the compiler no longer does for these statements (cdna_hip_programming.md 5.7) and how the schedule covers it:
  * MFMA result -> VALU read: >= 11 wait states for the 8-pass 32x32x16: the first reader of S_x comes a whole MFMA gap (>= 9 instructions) + one
    MFMA + an s_nop 1 behind the last S MFMA of x;
  * VALU write -> MFMA operand read: the packed P registers are finished >= 2 instructions before the first PV MFMA (s_nop 1 in front of it);
  * v_exp_f32 result -> dependent VALU: never the next instruction (the adds trail the exponentials by >= 2 statements);
  * LDS reads are compiler-visible loads: hipcc still places their lgkmcnt waits.

Tile step of query tile X against the current key tile (lane = query, 16 keys per lane; !CT form of the C++ round: S = K Q'^T with C = 0):
  S0..S3   four S MFMAs                                   M0..M7  maximum tree (7 x v_max3 / v_max) on raw scores
  W        half-wave exchange + max                       R       mxr = mx - m;  DEC: C++ branch  `rescale(X)` when any(mxr > 8) or first tile
  D0..D15  d_r = s_r - m                                  E0..E15 p_r = exp2(d_r)
  A0..A15  pa / pb partial sums, l update                 C0..C7  pk_i = cvt_pk(p_2i, p_2i+1)
  P0..P3   four PV MFMAs
Rounds: NQ = 2 tiles (A, B) per wave -- slot 0: S_A beside the second half of B's previous step, slot 1: PV_B(previous) beside A's first half,
slot 2: S_B beside A's second half (+ V fragment reads), slot 3: PV_A beside B's first half (+ barrier / ring advance / K fragment reads).
NQ = 1: the chain in sequence (its SIMD partner wave fills the gaps)."""
import sys

OUT = []


def emit(s):
    OUT.append(s)


def first_half(X, pad):
    """ops of tile X from the maximum tree to the first exponentials; returns list of statement strings (DEC is a C++ marker)"""
    T = X
    X = 'AB'[T] + '.'
    s = (lambda r: f'{X}sm[{r}]') if pad else (lambda r: f't[{T}].s[{r}]')
    ops = []
    if pad:                                              # padding keys of the sequence's last tile: + (-1e30) from a lane-constant table
        for r in range(16):
            ops.append(f'PA_ADD({X}sm[{r}], t[{T}].s[{r}], padadd[{r}]);')
    for i in range(5):
        ops.append(f'PA_MAX3({X}t{i}, {s(3 * i)}, {s(3 * i + 1)}, {s(3 * i + 2)});')
    ops.append(f'PA_MAX3({X}u0, {X}t0, {X}t1, {X}t2);')
    ops.append(f'PA_MAX3({X}u1, {X}t3, {X}t4, {s(15)});')
    ops.append(f'PA_MAX({X}mx, {X}u0, {X}u1);')
    ops.append(f'PA_HALFMAX({X}mx);')
    ops.append(f'PA_SUB({X}mxr, {X}mx, t[{T}].m);')
    ops.append(f'DEC({T}, {X[0]});')
    for r in range(16):
        ops.append(f'PA_SUB({X}d[{r}], {s(r)}, t[{T}].m);')
    for r in range(6):
        ops.append(f'PA_EXP({X}p[{r}], {X}d[{r}]);')
    return ops


def second_half(X):
    T = X
    X = 'AB'[T] + '.'
    ops = []
    # exponentials 6..15 interleaved with the sums (pa = p0 + p2 + ..., pb = p1 + p3 + ...) and the packs: two exps, then adds / packs of values that are
    # at least two statements old (no consumer directly behind its v_exp_f32)
    ops.append(f'PA_EXP({X}p[6], {X}d[6]);'); ops.append(f'PA_EXP({X}p[7], {X}d[7]);')
    ops.append(f'PA_ADD({X}pa, {X}p[0], {X}p[2]);'); ops.append(f'PA_ADD({X}pb, {X}p[1], {X}p[3]);')
    ops.append(f'PA_CVT(t[{T}].pk[0], {X}p[0], {X}p[1]);'); ops.append(f'PA_CVT(t[{T}].pk[1], {X}p[2], {X}p[3]);')
    ops.append(f'PA_EXP({X}p[8], {X}d[8]);'); ops.append(f'PA_EXP({X}p[9], {X}d[9]);')
    ops.append(f'PA_ADD({X}pa, {X}pa, {X}p[4]);'); ops.append(f'PA_ADD({X}pb, {X}pb, {X}p[5]);')
    ops.append(f'PA_CVT(t[{T}].pk[2], {X}p[4], {X}p[5]);')
    ops.append(f'PA_EXP({X}p[10], {X}d[10]);'); ops.append(f'PA_EXP({X}p[11], {X}d[11]);')
    ops.append(f'PA_ADD({X}pa, {X}pa, {X}p[6]);'); ops.append(f'PA_ADD({X}pb, {X}pb, {X}p[7]);')
    ops.append(f'PA_CVT(t[{T}].pk[3], {X}p[6], {X}p[7]);')
    ops.append(f'PA_EXP({X}p[12], {X}d[12]);'); ops.append(f'PA_EXP({X}p[13], {X}d[13]);')
    ops.append(f'PA_ADD({X}pa, {X}pa, {X}p[8]);'); ops.append(f'PA_ADD({X}pb, {X}pb, {X}p[9]);')
    ops.append(f'PA_CVT(t[{T}].pk[4], {X}p[8], {X}p[9]);')
    ops.append(f'PA_EXP({X}p[14], {X}d[14]);'); ops.append(f'PA_EXP({X}p[15], {X}d[15]);')
    ops.append(f'PA_ADD({X}pa, {X}pa, {X}p[10]);'); ops.append(f'PA_ADD({X}pb, {X}pb, {X}p[11]);')
    ops.append(f'PA_CVT(t[{T}].pk[5], {X}p[10], {X}p[11]);')
    ops.append(f'PA_ADD({X}pa, {X}pa, {X}p[12]);'); ops.append(f'PA_ADD({X}pb, {X}pb, {X}p[13]);')
    ops.append(f'PA_CVT(t[{T}].pk[6], {X}p[12], {X}p[13]);')
    ops.append(f'PA_ADD({X}pa, {X}pa, {X}p[14]);'); ops.append(f'PA_ADD({X}pb, {X}pb, {X}p[15]);')
    ops.append(f'PA_CVT(t[{T}].pk[7], {X}p[14], {X}p[15]);')
    ops.append(f'PA_ADD({X}pa, {X}pa, {X}pb);')
    ops.append(f'PA_ADD(t[{T}].l, t[{T}].l, {X}pa);')
    return ops


def s_mfmas(X, first):
    ops = []
    for k in range(4):
        if k == 0:
            ops.append(f'PA_MFMA_Z(t[{X}].s, kf[0], t[{X}].qf[0]);')
        else:
            ops.append(f'PA_MFMA(t[{X}].s, kf[{k}], t[{X}].qf[{k}]);')
    return ops


def pv_mfmas(T):
    X = 'AB'[T] + '.'
    return [f'PA_PV_FIRST(t[{T}].o0, vf[0][0], {X}f0);', f'PA_MFMA(t[{T}].o1, vf[0][1], {X}f0);',
            f'PA_MFMA(t[{T}].o0, vf[1][0], {X}f1);', f'PA_MFMA(t[{T}].o1, vf[1][1], {X}f1);']


def chunks(ops, n):
    """split ops into n nearly equal consecutive chunks"""
    k, r = divmod(len(ops), n)
    out, i = [], 0
    for j in range(n):
        m = k + (1 if j < r else 0)
        out.append(ops[i:i + m]); i += m
    return out


def slot(mfmas, fillers, extra=None, lead=None):
    """four gaps: MFMA j, then filler chunk j (and extra[j] statements)"""
    fc = chunks(fillers, 4) if fillers else [[], [], [], []]
    for j in range(4):
        if mfmas:
            emit('    ' + mfmas[j])
        if lead and j == 0:
            for l in lead:
                emit('    ' + l)
        for f in fc[j]:
            emit('    ' + f)
        if extra and extra[j]:
            for e in extra[j]:
                emit('    ' + e)


def round_nq2(mode, pad):
    """mode: 'FIRST' | 'STEADY' | 'DRAIN'"""
    cur, prev = mode != 'DRAIN', mode != 'FIRST'
    name = f'P4A_ROUND_NQ2_{mode}' + ('_PAD' if pad else '')
    emit(f'#define {name}() do {{ \\')
    body = []
    global OUT
    save, OUT = OUT, body
    # slot 0: S_A | B second half (previous key tile)
    slot(s_mfmas(0, mode == 'FIRST') if cur else None, second_half(1) if prev else None)
    # slot 1: PV_B(previous) | A first half
    if prev:
        emit('    PA_PFRAG(1, B);')
    slot(pv_mfmas(1) if prev else None, first_half(0, pad) if cur else None, lead=(['PA_NOP1();'] if prev else ['PA_NOP11();']) if cur else None)
    if cur:
        # slot 2: S_B | A second half (+ V fragment reads of this key tile: PV_B(previous) has issued)
        vreads = [[f'PA_VREAD({j & 1}, {j >> 1});'] for j in range(4)]
        slot(s_mfmas(1, mode == 'FIRST'), second_half(0), extra=vreads)
        # slot 3: PV_A | B first half (+ barrier / ring advance behind the first PV MFMA, K fragments of the next key tile behind the second)
        emit('    PA_PFRAG(0, A);')
        slot(pv_mfmas(0), first_half(1, pad), extra=[['PA_ADVANCE();'], ['PA_KREAD();'], [], []], lead=['PA_NOP1();'])
    OUT = save
    for l in body:
        emit(l + ' \\')
    emit('} while (0)')
    emit('')


def round_nq1(mode, pad):
    name = f'P4A_ROUND_NQ1_{mode}' + ('_PAD' if pad else '')
    emit(f'#define {name}() do {{ \\')
    body = []
    global OUT
    save, OUT = OUT, body
    sm = s_mfmas(0, mode == 'FIRST')
    for k in range(4):
        emit('    ' + sm[k])
        emit(f'    PA_VREAD({k & 1}, {k >> 1});')
    emit('    PA_NOP11();')
    for o in first_half(0, pad):
        emit('    ' + o)
    for o in second_half(0):
        emit('    ' + o)
    emit('    PA_PFRAG(0, A);')
    for o in pv_mfmas(0):
        emit('    ' + o)
    emit('    PA_ADVANCE();')
    emit('    PA_KREAD();')
    OUT = save
    for l in body:
        emit(l + ' \\')
    emit('} while (0)')
    emit('')


def main():
    emit('// GENERATED by tools/gen_attn_round.py -- do not edit.  Key-tile rounds of tools/attn_fwd_p4.hip with every hot instruction as its own')
    emit('// asm volatile statement (source order = issue order); see the generator for the schedule and the hazard argument.')
    emit('')
    for mode in ('FIRST', 'STEADY', 'DRAIN'):
        round_nq2(mode, False)
    round_nq2('STEADY', True)
    for mode in ('FIRST', 'STEADY'):
        round_nq1(mode, False)
    round_nq1('STEADY', True)
    open(sys.argv[1] if len(sys.argv) > 1 else 'tools/attn_round_asm.inc', 'w').write('\n'.join(OUT) + '\n')


if __name__ == '__main__':
    main()
