"""profiles/hbm_traffic.json from the raw PMC sums of one GPU visit (scripts/gpu_round6.sh writes pmc_traffic_raw.json).

    python scripts/update_traffic.py gpurun_out/<tag>/pmc_traffic_raw.json <tag>

Every record carries the hash of the kernel's source file at the time of the measurement; bench.py reports `traffic`
only while that hash still matches (a kernel that changed since its PMC pass reports null, not a stale number).
fetch_scale: gfx950's FETCH_SIZE tallies a wide coalesced read (16 bytes per lane over segments of 128 bytes or more) at
half its size (MI355X_MICROARCH.md, HBM; calibrated in round 1 on column_stats, whose doubled count equals its
algorithmic bytes); isolated 64-byte requests are tallied at full size (calibrated on the trimmed-mean kernels of round 3 BEFORE their
tile order followed the XCDs: raw = algorithmic; with neighbouring tiles on one L2 the same bytes arrive as 128-byte requests and
the raw count is exactly half: profiles/r03t vs r03v).
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'attacking_federate_learning_amd', 'csrc')

# bench key -> (kernel name prefix in the raw file, source file, fetch_scale, launches per step, note)
RULES = {
    'c4/gram_tile': ('c4/gram_planes', 'gram_planes.hip', 2.0, 10, 'LDS-DMA pieces of 1 KiB', {'arithmetic': 'f16x2'}),   # gram_planes16_kernel since round 6
    'c4/plane_split': ('c4/plane_split_f16_stream_kernel', 'gram_planes.hip', 2.0, 10, 'f32x4 per thread over 128-byte row segments', {}),
    # (until the XCD-contiguous tile order the two 64-byte halves of a line were requested by two L2s and tallied at full size:
    #  raw = algorithmic; now neighbouring tiles meet in one L2 and the counter sees 128-byte requests, tallied at 64)
    'c4/trimmed_mean': ('c4/window_lean_kernel', 'window_lean.hip', 2.0, 1, '64-byte row segments of neighbouring tiles merged into 128-byte requests', {}),
    'c3/trimmed_mean': ('c3/window_lean_kernel', 'window_lean.hip', 2.0, 1, '64-byte row segments of neighbouring tiles merged into 128-byte requests', {}),
    'c2/gram_tile': ('c2/small_gram_kernel', 'krum_small.hip', 2.0, 1, '8 x 16-byte loads per row slice', {'arithmetic': 'f16x2'}),
    # the register-resident attack statistics read 4 bytes per lane, 128-byte row segments (one per half wave): which way the
    # counter tallies them is decided from the count itself -- a kernel cannot fetch LESS than its input (m x D x 4 bytes), so a
    # raw count below three quarters of that is the halved tally (scale 2), anything else is taken as it is (scale 1)
    'attack/column_stats': ('attack/column_resident_kernel', 'column_stats.hip', None, 1, '4-byte loads, 128-byte row segments',
                            {'algorithmic_bytes': 2400 * 1000000 * 4.0}),
}


def sha16(name):
    return hashlib.sha256(open(os.path.join(CSRC, name), 'rb').read()).hexdigest()[:16]


def main():
    raw = json.load(open(sys.argv[1]))
    tag = sys.argv[2]
    path = os.path.join(ROOT, 'profiles', 'hbm_traffic.json')
    table = json.load(open(path))
    for key, (prefix, src, scale, per_step, note, extra) in RULES.items():
        hits = [(k, v) for k, v in raw.items() if k.startswith(prefix)]
        if not hits:
            continue
        name, v = max(hits, key=lambda kv: kv[1]['FETCH_SIZE_KB_sum'])
        launches = v['launches']
        fetch, write = v['FETCH_SIZE_KB_sum'] / launches, v['WRITE_SIZE_KB_sum'] / launches
        if scale is None:
            scale = 2.0 if fetch * 1024.0 < 0.75 * extra['algorithmic_bytes'] else 1.0
        rec = {'kernel': name.split('/', 1)[1], 'source': src, 'source_sha16': sha16(src), 'measured': tag,
               'FETCH_SIZE_KB_per_launch': fetch, 'WRITE_SIZE_KB_per_launch': write, 'fetch_scale': scale,
               'hbm_bytes_per_launch': (scale * fetch + write) * 1024.0, 'launches_sampled': launches,
               'launches_per_step': per_step,
               'method': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes of the bench command '
                         '(scripts/gpu_round6.sh); KB -> bytes x 1024; fetch_scale %.1f: %s.  Fabric side of L2: '
                         'Infinity-Cache hits are included' % (scale, note)}
        rec.update(extra)
        table[key] = rec
        print('%-18s %-52s %.4g GB per launch' % (key, rec['kernel'], rec['hbm_bytes_per_launch'] / 1e9))
    json.dump(table, open(path, 'w'), indent=1)


if __name__ == '__main__':
    main()
