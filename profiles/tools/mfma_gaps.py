"""What the compiler puts BETWEEN the asm statements of a dependent MFMA chain (round 5, cluster tile): lists, for a kernel in a `hipcc -S`
listing, every run of v_mfma_f32_16x16x4 instructions (allowing only what the hand-scheduled statements themselves contain between two
MFMAs: ds_read2st64, s_waitcnt, s_nop, global_load into a[...]) and the instructions between consecutive runs, counted by mnemonic. Text
order, not execution order: a "gap" includes the code of branches not taken (slow paths).

    hipcc -O3 --offload-arch=gfx950 --cuda-device-only -S x.hip -o x.s
    python profiles/tools/mfma_gaps.py x.s <mangled kernel name prefix> [runs to list = 50] [first run = 0] [runs whose gap is printed in full, comma separated]

This is how the v_accvgpr_read in front of every statement, the 16 scalar adds per weight chunk and the taken branches were found
(profiles/DESIGN_HISTORY.md, round 5, item 14).
"""
import re,sys,collections
lines=open(sys.argv[1]).read().split('\n')
name=sys.argv[2]
start=[i for i,l in enumerate(lines) if l.startswith(name) and (': ; @' in l or l.rstrip().endswith(':'))][0]
end=[i for i,l in enumerate(lines[start:],start) if l.startswith('.Lfunc_end')][0]
seg=lines[start:end]
print('function lines',len(seg))
def isins(t): return t and not t.startswith(';') and not t.startswith('.') and not t.endswith(':')
ins=[(i,l.strip()) for i,l in enumerate(seg) if isins(l.strip())]
# mark mfma 16x16x4 positions
pos=[k for k,(i,t) in enumerate(ins) if t.startswith('v_mfma_f32_16x16x4')]
print('mfma16 count',len(pos))
# group into runs separated by >0 "foreign" instructions (not ds_read2st64/s_waitcnt/global_load a[/s_nop)
def foreign(t): return not (t.startswith('ds_read2st64') or t.startswith('s_waitcnt') or t.startswith('global_load_dwordx4 a[') or t.startswith('s_nop') or t.startswith('v_mfma_f32_16x16x4'))
runs=[]; cur=[pos[0],pos[0],1]
for p in pos[1:]:
    if any(foreign(ins[q][1]) for q in range(cur[1]+1,p)): runs.append(cur); cur=[p,p,1]
    else: cur[1]=p; cur[2]+=1
runs.append(cur)
print('runs',len(runs))
lim=int(sys.argv[3]) if len(sys.argv)>3 else 50
off=int(sys.argv[4]) if len(sys.argv)>4 else 0
for k in range(off,min(off+lim,len(runs)-1)):
    a,b=runs[k][1]+1,runs[k+1][0]
    g=[ins[q][1] for q in range(a,b)]
    c=collections.Counter(x.split()[0] for x in g)
    print(k,'mfma',runs[k][2],'gap',len(g),dict(c))
if len(sys.argv)>5:
    for k in [int(x) for x in sys.argv[5].split(',')]:
        a,b=runs[k][1]+1,runs[k+1][0]
        print('---- gap after run',k)
        for q in range(a,b): print('   ',ins[q][1])
