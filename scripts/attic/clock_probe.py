"""Shader clock and socket power while one Gram variant loops (development aid)."""
import os, sys, subprocess, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from attacking_federate_learning_amd.engine import get_engine
eng = get_engine()
n, d = 4000, 1000000
g = torch.randn((n, d), device='cuda')
samples = []
stop = False

def watch():
    while not stop:
        out = subprocess.run(['rocm-smi', '-d', '0', '--showclocks', '--showpower'], capture_output=True, text=True).stdout
        sclk = [l for l in out.splitlines() if 'sclk' in l]
        pw = [l for l in out.splitlines() if 'Power' in l]
        samples.append((sclk[0].split('(')[-1].rstrip(')') if sclk else '?', pw[0].split(':')[-1].strip() if pw else '?'))

for label, env in (('fused', {'BYZ_GRAM_PLANES': '0'}), ('planes v3', {'BYZ_GRAM_PLANES': '1', 'BYZ_GRAM_PLANES_VARIANT': '3'}),
                   ('planes v3 no-DMA', {'BYZ_GRAM_PLANES': '1', 'BYZ_GRAM_PLANES_VARIANT': '13'}),
                   ('planes v3 no-MFMA', {'BYZ_GRAM_PLANES': '1', 'BYZ_GRAM_PLANES_VARIANT': '23'}),
                   ('planes v1', {'BYZ_GRAM_PLANES': '1', 'BYZ_GRAM_PLANES_VARIANT': '1'})):
    os.environ.update(env)
    eng.gram(g); torch.cuda.synchronize()
    samples.clear(); stop = False
    th = threading.Thread(target=watch); th.start()
    t0 = time.time()
    for _ in range(25):
        eng.gram(g)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 25
    stop = True; th.join()
    mid = samples[1:-1] or samples
    print('%-20s %.1f ms per Gram; samples (sclk, W): %s' % (label, dt * 1e3, mid[:8]), flush=True)
