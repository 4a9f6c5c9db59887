"""The cluster tile (csrc/distr_mlp.hpp) keeps its weight ring, accumulators and granule landing zone in FIXED registers that only its
asm statements name (a[96:255] at most, v[224:255]; with 8 members also v[202:223]: B fragments and per-lane constants); every statement lists them as clobbered so that the compiler keeps nothing there
across them. This checks the generated code: inside the span of the cluster code (first .. last asm statement that names a fixed
register) no COMPILER-generated instruction may name one of them.

    hipcc ... --cuda-device-only -S x.hip -o x.s ;  python profiles/tools/check_fixed_regs.py x.s <kernel name substring> [a_lo=96] [v_lo=224]
"""
import re
import sys


def main(argv=None):
    """Returns the number of findings (0 = clean); importable: main(['x.s', 'k_step'])."""
    argv = sys.argv[1:] if argv is None else argv
    path, key = argv[0], argv[1]
    a_lo = int(argv[2]) if len(argv) > 2 else 96
    v_lo = int(argv[3]) if len(argv) > 3 else 224
    lines = open(path).read().split('\n')
    start = end = None
    for i, l in enumerate(lines):
        if start is None and re.match(r'^_Z\S+:', l) and key in l:
            start = i
        elif start is not None and l.startswith('.Lfunc_end'):
            end = i
            break
    # statements that name fixed registers (hex register indices only come from the "n" operands of those asm statements)
    asm_lines = [i for i in range(start, end) if re.search(r'\b[av]\[0x', lines[i])]
    lo, hi = min(asm_lines), max(asm_lines)
    # cluster size of the code around a line: the tag ("; distr-cl N", first line of every CL_ASM statement) of the NEAREST fixed-register
    # statement on either side (round 6: the accumulator base of the previous MFMA statement mis-classified the prologue of a 2-member
    # instance laid out behind an 8-member one); without tags (older .s files) the accumulator base of the previous MFMA statement
    import bisect
    tags = [(i, int(re.search(r'distr-cl (\d)', lines[i]).group(1))) for i in range(start, end) if '; distr-cl ' in lines[i]]
    tag_idx = [t[0] for t in tags]
    mf = [(i, int(re.search(r'v_mfma\S+ a\[0x([0-9a-f]+)', lines[i]).group(1), 16)) for i in asm_lines if 'v_mfma' in lines[i]]
    mf_idx = [m[0] for m in mf]
    def a_limit(i):
        if tags:
            k = bisect.bisect_left(tag_idx, i)
            cand = [tags[j] for j in (k - 1, k) if 0 <= j < len(tags)]
            cl = min(cand, key=lambda t: abs(t[0] - i))[1]
            return {8: 120, 4: 112, 2: 96}[cl]
        k = bisect.bisect_right(mf_idx, i) - 1
        base = mf[max(k, 0)][1]
        return {0xf8: 120, 0xfc: 120, 0xf0: 112, 0xe0: 96}.get(base & ~7 if base >= 0xf8 else base & ~15, a_lo)
    def v_limit(i):      # 8 members: v[202:255] (B fragments v[208:223], per-lane constants v[202:207]); else the landing zone only
        return 202 if a_limit(i) == 120 else v_lo
    WIN = 300          # a compiler instruction is "inside cluster code" when fixed-register statements lie within WIN lines on both sides
    in_asm = False
    hits = []
    for i in range(lo, hi + 1):
        l = lines[i].strip()
        if l.startswith(';;#ASMSTART'):
            in_asm = True
            continue
        if l.startswith(';;#ASMEND'):
            in_asm = False
            continue
        if in_asm or not l or l[0] in ';.':
            continue
        k = bisect.bisect_left(asm_lines, i)
        before = asm_lines[k - 1] if k > 0 else -10 ** 9
        after = asm_lines[k] if k < len(asm_lines) else 10 ** 9
        if i - before > WIN or after - i > WIN:
            continue
        al = a_limit(i)
        for m in re.finditer(r'\b([av])(\d+)\b|\b([av])\[(\d+):(\d+)\]', l.split(';')[0]):
            f = m.group(1) or m.group(3)
            top = int(m.group(2)) if m.group(2) else int(m.group(5))
            if (f == 'a' and top >= al) or (f == 'v' and top >= v_limit(i)):
                hits.append((i, l))
                break
    print('%s: cluster code spans lines %d..%d; compiler instructions naming a%d+ / v%d+ inside it: %d' % (key, lo, hi, a_lo, v_lo, len(hits)))
    for h in hits[:20]:
        print('   ', h[0], h[1][:110])
    return len(hits)


if __name__ == '__main__':
    sys.exit(1 if main() else 0)
