"""Static instruction mix of one kernel of a HIP source, split at its workgroup barriers (no GPU needed).

    python scripts/isa_phases.py window_lean.hip 'window_lean_kernelILi4ELi16ELi512ELi1ELi6ELb0E'
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from attacking_federate_learning_amd import build_native   # noqa: E402


def main():
    src, pattern = sys.argv[1], sys.argv[2]
    out = '/tmp/isa_%s.s' % src.replace('.hip', '')
    cmd = [build_native.hipcc()] + build_native.COMMON_FLAGS + build_native.EXTRA_FLAGS.get(src, []) + [
        '-S', '--cuda-device-only', '-o', out, os.path.join(build_native.CSRC, src)]
    subprocess.run(cmd, check=True, capture_output=True)
    lines = open(out).read().split('\n')
    starts = [i for i, l in enumerate(lines) if re.match(r'^_Z\w*%s\w*:' % re.escape(pattern), l)]
    if not starts:
        raise SystemExit('no kernel label matches %r' % pattern)
    start = starts[0]
    end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
    seg, b = {}, 0
    for l in lines[start:end]:
        t = l.strip()
        if not l.startswith('\t') or t.startswith('.') or t.startswith(';'):
            continue
        op = t.split()[0]
        if op == 's_barrier':
            b += 1
        kind = ('scratch' if op.startswith('scratch_') else 'valu' if op.startswith('v_') else 'salu' if op.startswith('s_')
                else 'lds' if op.startswith('ds_') else 'mem')
        d = seg.setdefault(b, {})
        d[kind] = d.get(kind, 0) + 1
        if len(sys.argv) > 3 and sys.argv[3] == 'ops':
            d.setdefault('ops', {})
            d['ops'][op] = d['ops'].get(op, 0) + 1
    print(lines[start][:120])
    for k in sorted(seg):
        ops = seg[k].pop('ops', None)
        print('  after barrier %d: %s' % (k, seg[k]))
        if ops:
            print('      ' + ', '.join('%s %d' % kv for kv in sorted(ops.items(), key=lambda kv: -kv[1])[:14]))


if __name__ == '__main__':
    main()
