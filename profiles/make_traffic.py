"""profiles/rNN_traffic.json (what bench.py reports as roofline.traffic) from the PMC summaries of one round:
    python profiles/make_traffic.py profiles/r02      -> reads r02_pmc_{fetch,write,mfma,l2}.md, writes r02_traffic.json
FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 counts 64 B per 128-B request for 16 B/lane streaming loads, hence the x2 on FETCH_SIZE
(MI355X_MICROARCH.md, HBM section). A 'launch' = one march launch of bench.py's roofline (k_step or coarse k_march)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'dist-renderer_amd'))


def rows(path):
    out = {}
    for line in open(path):
        m = re.match(r'\| (?:void )?(k_step|k_march|k_tail)[^|]*\| (\w+) \| (\d+) \| ([0-9.e+]+) \|', line)
        if m:
            out[(m.group(1), m.group(2))] = (int(m.group(3)), float(m.group(4)))
    return out


def run_digest(pre):
    """csrc digest of the library the PMC passes ran on: the bench line of the SAME run (profiles/rNN_bench.json, written on the GPU box by the
    script that also made the passes) carries distr.binding.source_digest() of the box's tree; without that file: this tree's."""
    try:
        return json.load(open(pre + '_bench.json'))['roofline']['csrc_sha256']
    except (OSError, KeyError, ValueError):
        return __import__('distr.binding', fromlist=['source_digest']).source_digest()


def main():
    pre = sys.argv[1]
    f, w, mf, l2 = (rows('%s_pmc_%s.md' % (pre, k)) for k in ('fetch', 'write', 'mfma', 'l2'))
    fams = [k for k in ('k_step', 'k_march', 'k_tail') if (k, 'FETCH_SIZE') in f]      # (k_tail: only when the workload reaches the sticky regime)
    n = sum(f[(k, 'FETCH_SIZE')][0] for k in fams)
    fetch = sum(f[(k, 'FETCH_SIZE')][1] for k in fams)
    write = sum(w[(k, 'WRITE_SIZE')][1] for k in fams)
    busy = lambda k: mf[(k, 'SQ_VALU_MFMA_BUSY_CYCLES')][1] / (mf[(k, 'GRBM_GUI_ACTIVE')][1] / 8.0 * 1024.0)
    hit = lambda k: l2[(k, 'TCC_HIT_sum')][1] / (l2[(k, 'TCC_HIT_sum')][1] + l2[(k, 'TCC_MISS_sum')][1])
    # round 6: the L2's memory-side read requests split by destination (TCC_EA0_RDREQ_DRAM vs all TCC_EA0_RDREQ), if that pass exists
    ea = None
    try:
        e, dr = rows(pre + '_pmc_ea.md'), rows(pre + '_pmc_dram.md')
        rd = sum(e[(k, 'TCC_EA0_RDREQ_sum')][1] for k in fams)
        rd32 = sum(e[(k, 'TCC_EA0_RDREQ_32B_sum')][1] for k in fams)
        rdd = sum(dr[(k, 'TCC_EA0_RDREQ_DRAM_sum')][1] for k in fams)
        wrd = sum(dr[(k, 'TCC_EA0_WRREQ_DRAM_sum')][1] for k in fams)
        ea = {'TCC_EA0_RDREQ_sum': rd, 'TCC_EA0_RDREQ_32B_sum': rd32, 'TCC_EA0_RDREQ_DRAM_sum': rdd, 'TCC_EA0_WRREQ_DRAM_sum': wrd,
              'rdreq_dram_fraction': rdd / rd if rd else None, 'fetch_size_check_kb': rd * 64.0 / 1024.0,
              'hbm_bytes_per_launch': None,
              'finding': 'every memory-side read request of the L2 is counted as "destined for DRAM (MC)" (TCC_EA0_RDREQ_DRAM = TCC_EA0_RDREQ; GMI / IO: none; '
                         'FETCH_SIZE = RDREQ x 64 B exactly): the counters rocprofv3 exposes on this stack (profiles/r06_rocprof_counters.txt: no MALL / '
                         'Infinity-Cache / UMC counter) see the fabric side of the L2 only -- whether a request is then served by the 256 MiB Infinity Cache or by '
                         'HBM is NOT observable. Upper bound: HBM traffic <= the fabric figure (bytes_per_launch). The whole working set of a render (14.5 MB '
                         'of packed weights + the ray state + the mask store of the live rows) is far below 256 MiB, so in steady state the re-fetched weight '
                         'stream is expected to be served on-die; hbm_bytes_per_launch stays null rather than guessed.'}
    except (OSError, KeyError):
        pass
    name = pre.split('/')[-1]
    d = {
        'kernel': "k_step / k_march (all tile-size roles of one march step = one 'launch' of bench.py's roofline)",
        'source': 'profiles/%s_pmc_fetch.md (FETCH_SIZE) + profiles/%s_pmc_write.md (WRITE_SIZE): rocprofv3 --pmc, separate passes, '
                  '`bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-split-bf16-pass` (%d march launches: %s); profiles/make_traffic.py'
                  % (name, name, n, ' + '.join('%d %s' % (f[(k, 'FETCH_SIZE')][0], k) for k in fams)),
        'fetch_size_kb_total': fetch, 'write_size_kb_total': write, 'march_launches': n,
        'fetch_correction': 'x2: gfx950 FETCH_SIZE counts 64 B per 128-B request for 16 B/lane streaming loads (MI355X_MICROARCH.md, HBM '
                            'section); the weight fragments are buffer_load_dwordx4',
        'bytes_per_launch': int(round((2.0 * fetch + write) * 1024.0 / n)),
        # the sources the profiled libdistr.so was built from (distr.binding.source_digest): bench.py quotes bytes_per_launch only while
        # csrc/ still has this digest
        'csrc_sha256': run_digest(pre),
        'mfma_busy': {'k_step': round(busy('k_step'), 3), 'k_march_coarse': round(busy('k_march'), 3),
                      'how': 'SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), profiles/%s_pmc_mfma.md; k_step covers all 44 '
                             'full-resolution steps of a forward including the latency-bound tail' % name},
        'l2_hit_rate': {'k_step': round(hit('k_step'), 3), 'k_march_coarse': round(hit('k_march'), 3),
                        'how': 'TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum), profiles/%s_pmc_l2.md' % name},
        'note': 'fabric-side bytes (L2 misses, mostly served by the 256 MiB Infinity Cache: the 6.3 MB weight stream is re-fetched by every '
                'XCD once per tile round because it exceeds the 4 MiB L2; includes the granule exchange of the cluster tiles: ordinary cached device memory since round 5, polled with sc1 loads); '
                'algorithmic HBM bytes per launch ~27 MB (weights once + 32 B state + 512 B saved mask per decoder evaluation). At ~0.87 ms '
                'per launch this is ~3 % of HBM peak: the kernel is MFMA-bound. Experiments: aliasing all 512x512 layers onto one weight array '
                '(stream fits L2) changes the dense rate by 0.3 %; a non-temporal hint on layers 1-4 cut the coarse launches\' fetches by 15 % '
                'and cost 1 % of the dense rate (profiles/README.md).',
    }
    if ea is not None:
        d['memory_side_reads'] = ea
        d['hbm_bytes_per_launch'] = None
    json.dump(d, open('%s_traffic.json' % pre, 'w'), indent=1)
    print(json.dumps(d, indent=1))


if __name__ == '__main__':
    main()
