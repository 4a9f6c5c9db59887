"""Per-kernel resources of libdistr.so's gfx950 code object, read from its AMDGPU metadata notes (no GPU needed):
VGPR / AGPR / SGPR counts, scratch bytes per lane (`private_segment_fixed_size`), spill counts, LDS bytes.

    python profiles/tools/kernel_resources.py [path/to/libdistr.so] [--filter k_step] [--md]

`resources(path)` is what `__graft_entry__.build()` calls to assert that the march / backward kernels carry no scratch."""
import os
import re
import struct
import subprocess
import sys
import tempfile

READELF = os.environ.get('LLVM_READELF', '/opt/rocm/lib/llvm/bin/llvm-readelf')
MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'


def extract_code_objects(path):
    """[(triple, bytes)] of every bundle entry of a HIP fat binary embedded in `path`."""
    blob = open(path, 'rb').read()
    out = []
    pos = blob.find(MAGIC)
    while pos >= 0:
        n, = struct.unpack_from('<Q', blob, pos + len(MAGIC))
        q = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from('<QQQ', blob, q)
            triple = blob[q + 24:q + 24 + tlen].decode()
            q += 24 + tlen
            if size:
                out.append((triple, blob[pos + off:pos + off + size]))
        pos = blob.find(MAGIC, pos + len(MAGIC))
    return out


def demangle(names):
    try:
        p = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt'], input='\n'.join(names), capture_output=True, text=True, check=True)
        return p.stdout.split('\n')[:len(names)]
    except Exception:
        return list(names)


def resources(path):
    """{demangled kernel name: {vgpr, agpr, sgpr, scratch, vgpr_spill, sgpr_spill, lds}} for the gfx950 code object of `path`."""
    res = {}
    for triple, co in extract_code_objects(path):
        if 'gfx950' not in triple:
            continue
        with tempfile.NamedTemporaryFile(suffix='.co') as f:
            f.write(co)
            f.flush()
            txt = subprocess.run([READELF, '--notes', f.name], capture_output=True, text=True, check=True).stdout
        cur = None
        kernels = []
        # a kernel's entry lists its keys alphabetically: .agpr_count opens it, .wavefront_size closes it (argument entries carry none of the keys read here)
        for line in txt.split('\n'):
            m = re.match(r'\s*-?\s*\.(\w+):\s*(.*)$', line)
            if not m:
                continue
            k, v = m.group(1), m.group(2).strip()
            if k == 'agpr_count':
                cur = {}
                kernels.append(cur)
            if cur is None:
                continue
            if k in ('agpr_count', 'vgpr_count', 'sgpr_count', 'private_segment_fixed_size', 'vgpr_spill_count', 'sgpr_spill_count',
                     'group_segment_fixed_size'):
                cur[k] = int(v)
            elif k == 'symbol':
                cur['symbol'] = v.strip("'\"")
                cur['name'] = re.sub(r'\.kd$', '', cur['symbol'])
            elif k == 'wavefront_size':
                cur = None
        names = demangle([k.get('name', k['symbol']) for k in kernels])
        for k, n in zip(kernels, names):
            res[n] = dict(vgpr=k.get('vgpr_count', -1), agpr=k.get('agpr_count', -1), sgpr=k.get('sgpr_count', -1),
                          scratch=k.get('private_segment_fixed_size', -1), vgpr_spill=k.get('vgpr_spill_count', 0),
                          sgpr_spill=k.get('sgpr_spill_count', 0), lds=k.get('group_segment_fixed_size', -1))
    return res


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    here = os.path.dirname(os.path.abspath(__file__))
    path = args[0] if args else os.path.join(here, '..', '..', 'dist-renderer_amd', 'csrc', 'libdistr.so')
    flt = None
    if '--filter' in sys.argv:
        flt = sys.argv[sys.argv.index('--filter') + 1]
        args = [a for a in args if a != flt]
        path = args[0] if args else os.path.join(here, '..', '..', 'dist-renderer_amd', 'csrc', 'libdistr.so')
    r = resources(path)
    print('| kernel | VGPR | AGPR | SGPR | scratch B/lane | VGPR spills | SGPR spills | LDS B |')
    print('|---|---|---|---|---|---|---|---|')
    for n in sorted(r):
        if flt and flt not in n:
            continue
        k = r[n]
        short = re.sub(r'^void distr::', '', n)
        short = re.sub(r'\(.*$', '', short)
        print('| `%s` | %d | %d | %d | %d | %d | %d | %d |' % (short, k['vgpr'], k['agpr'], k['sgpr'], k['scratch'], k['vgpr_spill'], k['sgpr_spill'], k['lds']))


if __name__ == '__main__':
    main()
