"""Round-2 development timings (not the contract bench): plane Gram vs fused Gram, the Bulyan loop by band setting."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from attacking_federate_learning_amd.engine import get_engine

eng = get_engine()
which = sys.argv[1:] or ["gram", "loop"]
gen = torch.Generator(device='cuda').manual_seed(0)


def kernel_ms(fn, iters=2, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    eng.timing(True)
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    out = {k: (round(v['total_ms'] / iters, 3), v['launches'] // iters) for k, v in eng.timing_read().items() if v['launches']}
    eng.timing(False)
    return out

if 'gram' in which:
    for n, d in ((4000, 1000000),):
        g = torch.randn((n, d), device='cuda', generator=gen)
        flops = float(n) * n * d
        for label, env in (('fused bf16x3', {'BYZ_GRAM_PLANES': '0', 'BYZ_GRAM_MODE': 'split'}),
                           ('planes bf16x3', {'BYZ_GRAM_PLANES': '1', 'BYZ_GRAM_MODE': 'split'}),
                           ('planes f16x2', {'BYZ_GRAM_PLANES': '1', 'BYZ_GRAM_MODE': 'f16x2'}),
                           ('planes f16x2 nbuf4', {'BYZ_GRAM_PLANES': '1', 'BYZ_GRAM_MODE': 'f16x2', 'BYZ_GRAM_PLANES_VARIANT': '4'}),
                           ('planes f16x2 no-DMA', {'BYZ_GRAM_PLANES': '1', 'BYZ_GRAM_MODE': 'f16x2', 'BYZ_GRAM_PLANES_VARIANT': '10'}),
                           ('planes f16x2 no-MFMA', {'BYZ_GRAM_PLANES': '1', 'BYZ_GRAM_MODE': 'f16x2', 'BYZ_GRAM_PLANES_VARIANT': '20'})):
            os.environ.pop('BYZ_GRAM_PLANES_VARIANT', None)
            os.environ.update(env)
            t = kernel_ms(lambda: eng.gram(g))
            tile, split = t.get('gram_tile', (0, 0))[0], t.get('plane_split', (0, 0))[0]
            print('gram N=%d D=%d %-22s tile %.2f ms (%.1f TF-eq)  split %.2f ms  -> %.1f TF-eq overall' % (
                n, d, label, tile, flops / tile / 1e9, split, flops / (tile + split) / 1e9), flush=True)
        for k in ('BYZ_GRAM_PLANES_VARIANT', 'BYZ_GRAM_MODE', 'BYZ_GRAM_PLANES'):
            os.environ.pop(k, None)
        del g

if 'loop' in which:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
    for n in (4000, 10000):
        f = int(n * 0.24)
        rng = np.random.default_rng(4100 + n)
        pts = rng.standard_normal((n, 16)).astype(np.float32)
        pts *= (1.0 + 0.5 * rng.permutation(n) / n).astype(np.float32)[:, None]
        p64 = pts.astype(np.float64)
        sq = (p64 * p64).sum(1)
        dist = np.sqrt(np.maximum(sq[:, None] + sq[None, :] - 2.0 * (p64 @ p64.T), 0.0)).astype(np.float32)
        dist = np.minimum(dist, dist.T)
        np.fill_diagonal(dist, np.inf)
        ref = None
        for band in ('rigorous', '-8', '0'):
            os.environ['BYZ_BULYAN_BAND'] = band
            t = kernel_ms(lambda: eng.bulyan_select(dist, n, f), iters=1, warm=1)
            sel = eng.bulyan_select(dist, n, f).tolist()
            if ref is None:
                ref = sel
            print('bulyan loop N=%d band=%s: %s rescored=%s same_as_rigorous=%s' % (
                n, band, {k: v for k, v in t.items() if k in ('bulyan_loop', 'row_sort')}, getattr(eng, 'bulyan_rescored', lambda: '?')(), sel == ref), flush=True)
        os.environ.pop('BYZ_BULYAN_BAND')

if 'tm' in which:
    for n, d, c in ((1000, 1000000, 200), (2080, 1 << 20, 1920), (5200, 1 << 19, 4800), (256, 1 << 21, 60), (100, 1 << 22, 20)):
        g = torch.randn((n, d), device='cuda', generator=gen)
        res = {}
        for label, env in (('rows', {'BYZ_TM_ROWS': '1'}), ('columns', {'BYZ_TM_ROWS': '0'})):
            os.environ.update(env)
            t = kernel_ms(lambda: eng.trimmed_mean(g, n, c), iters=3, warm=1)
            res[label] = eng.trimmed_mean(g, n, c).clone()
            ms = t['trimmed_mean'][0]
            print('trimmed_mean N=%d D=%d keep=%d %-8s %.3f ms  %.2f TB/s (%.2f of 8)  redone tiles %s' % (
                n, d, n - c - 1, label, ms, 4.0 * n * d / ms / 1e9, 4.0 * n * d / ms / 1e9 / 8.0, eng.trimmed_mean_redone()), flush=True)
        os.environ.pop('BYZ_TM_ROWS')
        diff = (res['rows'] - res['columns']).abs().max().item()
        print('    max |rows - columns| = %.3e' % diff, flush=True)
        del g
