"""Static check of the hand-counted vmcnt waits (csrc/distr_mlp.hpp, cluster tile): walks the disassembly of one kernel linearly,
models the in-order vector-memory queue (loads with their destination registers, stores as entries without one) and reports every
instruction that READS or OVERWRITES a register an outstanding load has not delivered yet -- e.g. a register-allocator copy
(v_accvgpr_write / v_mov / scratch spill) of a value whose inline-asm load is still in flight, or a wait count that is too large.
Control flow is ignored (the cluster code is unrolled, straight-line between its barriers); the queue is cleared at s_endpgm / s_setpc.
On compiler assembly (-S, with ;;#ASMSTART markers) only loads issued by inline asm are tracked (the compiler waits for its own).

    llvm-objdump -d --no-show-raw-insn <gfx950 code object> > all.s          (or: hipcc ... --cuda-device-only -S x.hip -o all.s)
    python profiles/tools/vm_hazard_scan.py all.s <mangled kernel name substring> [max reports]
"""
import re
import sys


def regs(tok):
    tok = tok.strip()
    m = re.fullmatch(r'([va])(\d+)', tok)
    if m:
        return {(m.group(1), int(m.group(2)))}
    m = re.fullmatch(r'([va])\[(\d+):(\d+)\]', tok)
    if m:
        return {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    return set()


def main(argv=None):
    """Returns the number of hazards found (0 = clean); importable: main(['x.s', 'k_step'])."""
    argv = sys.argv[1:] if argv is None else argv
    path, key = argv[0], argv[1]
    limit = int(argv[2]) if len(argv) > 2 else 40
    lines = open(path).read().split('\n')
    start = None
    end = len(lines)
    for i, l in enumerate(lines):
        if re.match(r'^[0-9a-f]+ <.*>:$', l) or re.match(r'^_Z\S+:', l) or (start is not None and l.startswith('.Lfunc_end')):
            if start is not None:
                end = i
                break
            if key in l:
                start = i
    has_markers = any(x.strip().startswith(';;#ASMSTART') for x in lines[start:end])     # compiler assembly: track only the hand-issued loads
    queue = []          # entries: (set of dest regs, text, line, issued by inline asm)
    nrep = 0
    nload = 0
    in_asm = False
    for i in range(start + 1, end):
        l = lines[i].strip()
        if l.startswith(';;#ASMSTART'):
            in_asm = True
        elif l.startswith(';;#ASMEND'):
            in_asm = False
        if not l or l.startswith('//') or l.startswith(';') or l.startswith('.'):
            continue
        l = l.split('//')[0].split(';')[0].strip()
        if not l or l.endswith(':'):
            continue
        parts = l.split(None, 1)
        op = parts[0]
        ops = [o.strip() for o in parts[1].split(',')] if len(parts) > 1 else []
        if op in ('s_endpgm',) or op.startswith('s_setpc'):
            queue = []
            continue
        if op == 's_waitcnt':
            m = re.search(r'vmcnt\((\d+)\)', l)
            if m:
                n = int(m.group(1))
                while len(queue) > n:
                    queue.pop(0)
            elif re.fullmatch(r's_waitcnt\s+0(x0)?', l):
                queue = []
            continue
        is_load = re.match(r'(global|buffer|flat|scratch)_load', op) is not None
        is_store = re.match(r'(global|buffer|flat|scratch)_(store|atomic)', op) is not None
        pending = set().union(*[q[0] for q in queue if (q[3] or not has_markers)]) if queue else set()
        if is_load:
            dst = regs(ops[0]) if ops else set()
            src = set().union(*[regs(o.split(' ')[0]) for o in ops[1:]]) if len(ops) > 1 else set()
            if 'lds' in l.split() or '_load_lds_' in op:          # LDS-DMA: no register destination, every operand is a source
                src = src | dst
                dst = set()
            hz = (src | dst) & pending
            if hz and nrep < limit:
                print('line %d: %-70s touches %s in flight from: %s' % (i, l, sorted(hz)[:4], [q[1] for q in queue if q[0] & hz][:2]))
                nrep += 1
            queue.append((dst, '%d:%s' % (i, l[:60]), i, in_asm))
            nload += 1
            continue
        allr = set().union(*[regs(o.split(' ')[0]) for o in ops]) if ops else set()
        if is_store:
            hz = allr & pending
            if hz and nrep < limit:
                print('line %d: %-70s reads %s in flight from: %s' % (i, l, sorted(hz)[:4], [q[1] for q in queue if q[0] & hz][:2]))
                nrep += 1
            queue.append((set(), '%d:%s' % (i, l[:60]), i, in_asm))
            continue
        hz = allr & pending
        if hz:
            nrep += 1
            if nrep <= limit:
                print('line %d: %-70s touches %s in flight from: %s' % (i, l, sorted(hz)[:4], [q[1] for q in queue if q[0] & hz][:2]))
    print('%s: %d loads scanned, %d hazards' % (key, nload, nrep))
    return nrep


if __name__ == '__main__':
    sys.exit(1 if main() else 0)
