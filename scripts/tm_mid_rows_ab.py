"""The trimmed mean beyond the register kernels (5,632 rows): the default dispatch (tall_select.hip since round 6's last session; the LDS
sort kernel with BYZ_TM_TALL=0) against the global-memory segment sort of csrc/large_rows.hip (BYZ_TM_LARGE=1).  Needs an MI355X."""


import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from attacking_federate_learning_amd.engine import get_engine  # noqa: E402


def main():
    eng = get_engine()
    rng = np.random.default_rng(1)
    print('%-18s %14s %14s   (GB/s of 4 R D bytes)' % ('rows x cols', 'default ms', 'segment sort ms'))
    for n, cols in [(5400, 32768), (6000, 32768), (8192, 32768), (8193, 32768), (8580, 32768), (10000, 32768), (12000, 16384), (16384, 16384),
                    (6000, 100000), (10000, 100000), (16384, 50000), (20001, 50000)]:
        g = eng.to_device(rng.standard_normal((n, cols), dtype=np.float32))
        c = int(0.48 * n)
        out = []
        for forced in (False, True):
            if forced:
                os.environ['BYZ_TM_LARGE'] = '1'
            best = 1e9
            for _ in range(3):
                eng.synchronize()
                t0 = time.perf_counter()
                r = eng.trimmed_mean(g, n, c)
                eng.synchronize()
                best = min(best, time.perf_counter() - t0)
            out.append((best * 1e3, r.numpy()))
            os.environ.pop('BYZ_TM_LARGE', None)
        assert np.allclose(out[0][1], out[1][1], rtol=1e-6, atol=1e-6, equal_nan=True)
        print('%-18s %14.2f %14.2f   %8.0f %8.0f' % ('%d x %d' % (n, cols), out[0][0], out[1][0], 4e-6 * n * cols / out[0][0], 4e-6 * n * cols / out[1][0]))
        del g


if __name__ == '__main__':
    main()
