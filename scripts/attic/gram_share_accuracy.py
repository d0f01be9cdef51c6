"""Where does the clients-layout distance matrix lose accuracy?  (torch-free diagnostic for tests/test_gpu_sharded.py)"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from attacking_federate_learning_amd.engine import Engine, DeviceBuffer   # noqa: E402


def run(eng, n, d, world, panel_cols, f, near, label):
    rng = np.random.default_rng(3)
    g = rng.standard_normal((n, d), dtype=np.float32)
    g *= (1.0 + 0.5 * rng.permutation(n) / n).astype(np.float32)[:, None]
    if f:
        g[:f] = (g[:f].mean(0) - 1.5 * g[:f].std(0)).astype(np.float32)
    for a, b, eps in near:
        g[a] = g[b] + np.float32(eps) * rng.standard_normal(d).astype(np.float32)
    rows_per = [n // world + (1 if r < n % world else 0) for r in range(world)]
    rows_per[0] += 3
    rows_per[-1] -= 3
    n_max = max(rows_per)
    starts = np.concatenate([[0], np.cumsum(rows_per)])
    row_index = eng.to_device(np.concatenate([r * n_max + np.arange(rows_per[r]) for r in range(world)]).astype(np.int32))
    total = np.zeros((n, n))
    panels = []
    for lo in range(0, d, panel_cols):
        width = min(panel_cols, d - lo)
        panel = np.full((world * n_max, width), np.nan, dtype=np.float32)
        for r in range(world):
            panel[r * n_max:r * n_max + rows_per[r]] = g[starts[r]:starts[r + 1], lo:lo + width]
        buf = eng.to_device(panel)
        panels.append(buf)
        for share in range(world):
            total += eng.gram_share(buf, row_index, world, share).numpy()
            eng.check()
    gram_dev = eng.to_device(total)
    dist = eng.distances_from_gram(gram_dev, n)
    raw = dist.numpy().copy()
    count = eng.near_pairs_count()
    sq = np.zeros(count)
    for buf in panels:
        sq += eng.near_pairs_sqdist(buf, count, row_index=row_index).numpy()
    eng.near_pairs_apply(eng.to_device(sq), dist)
    final = dist.numpy()
    g64 = g.astype(np.float64)
    sample = np.unique(np.concatenate([np.arange(0, n, 131), [n - 1, n - 2, n - 7, f, f + 1, f + 2, f + 3, 0, 1, 2468 % n]]))
    sq64 = (g64 * g64).sum(1)
    d2 = sq64[sample][:, None] + sq64[None, :] - 2 * (g64[sample] @ g64.T)
    # exact for the near pairs: difference first
    want = np.sqrt(np.maximum(d2, 0))
    for a, b, _ in near:
        for (p, q) in ((a, b), (b, a)):
            if p in sample:
                want[list(sample).index(p), q] = np.linalg.norm((g[p] - g[q]).astype(np.float64))
    gram_err = np.abs(total[sample] - g64[sample] @ g64.T) / np.sqrt(sq64[sample][:, None] * sq64[None, :])
    for name, m in (('raw', raw), ('final', final)):
        got = m[sample].astype(np.float64)
        rel = np.abs(got - want) / np.maximum(want, 1e-30)
        rel[:, :f] = 0          # (the fp64 'truth' of this script is itself inexact between identical rows)
        rel[np.arange(len(sample)), sample] = 0
        print('%-20s %-6s pairs listed %d' % (label, name, count), flush=True)
        for r in (n - 1, n - 2, n - 7, f + 3, f + 2, 2468, 131):
            k = list(sample).index(r)
            j = int(rel[k].argmax())
            print('    row %4d: max rel dist err %.2e at col %4d (got %.9g want %.9g); Gram err of the row: max %.2e, c_ii %.2e' % (
                r, rel[k].max(), j, got[k, j], want[k, j], gram_err[k].max(), gram_err[k, r]), flush=True)


def main():
    eng = Engine(0)
    n = 3000
    f = 720
    near = [(f + 3, f + 2, 1e-4), (n - 1, n - 7, 3e-4)]
    run(eng, n, 32800, 2, 16400, 0, near, 'no identical rows')
    run(eng, n, 32800, 2, 16400, 0, [], 'no near rows either')


if __name__ == '__main__':
    main()
