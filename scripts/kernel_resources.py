"""Static resources of every kernel in one HIP source (no GPU needed): registers, scratch, LDS, occupancy.

    python scripts/kernel_resources.py window_rows.hip [name-filter]

Compiles the file for gfx950 with the library's own flags plus -Rpass-analysis=kernel-resource-usage and prints one line
per kernel.  Scratch > 0 means spills: the first thing to look at after touching a register-resident kernel.
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from attacking_federate_learning_amd import build_native   # noqa: E402


def resources(src):
    path = os.path.join(build_native.CSRC, src)
    cmd = [build_native.hipcc()] + build_native.COMMON_FLAGS + build_native.EXTRA_FLAGS.get(src, []) + [
        '-Rpass-analysis=kernel-resource-usage', '-c', path, '-o', '/dev/null']
    text = subprocess.run(cmd, capture_output=True, text=True).stderr
    kernels, cur = [], None
    for line in text.splitlines():
        m = re.search(r'remark: [^ ]+ +(Function Name|[A-Za-z ]+): *(.*?) \[-Rpass', line) or \
            re.search(r'remark: +(Function Name|[A-Za-z \[\]/]+): *(.*?) \[-Rpass', line)
        if not m:
            m = re.search(r': +([A-Za-z][A-Za-z \[\]/]*): (.*?) \[-Rpass-analysis', line)
            if not m:
                continue
        key, val = m.group(1).strip(), m.group(2).strip()
        if key in ('Function Name', 'Name'):
            cur = {'name': val}
            kernels.append(cur)
        elif cur is not None:
            cur[key] = val
    return kernels


def demangle(names):
    out = subprocess.run(['c++filt'] + names, capture_output=True, text=True).stdout.splitlines()
    return [re.sub(r'\(.*$', '', o.replace('(anonymous namespace)::', '')).replace('byz::', '').replace('void ', '') for o in out]


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ''
    ks = resources(src)
    names = demangle([k['name'] for k in ks]) if ks else []
    print('%-64s %5s %5s %7s %7s %4s' % ('kernel', 'VGPR', 'AGPR', 'scratch', 'LDS', 'occ'))
    for k, n in zip(ks, names):
        if flt and flt not in n:
            continue
        print('%-64s %5s %5s %7s %7s %4s' % (n[:64], k.get('VGPRs', '?'), k.get('AGPRs', '?'),
                                              k.get('ScratchSize [bytes/lane]', '?'), k.get('LDS Size [bytes/block]', '?'),
                                              k.get('Occupancy [waves/SIMD]', '?')))


if __name__ == '__main__':
    main()
