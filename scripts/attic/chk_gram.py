import torch, sys, os
sys.path.insert(0, ".")
from attacking_federate_learning_amd.engine import get_engine
eng = get_engine()
n, d = 4000, 600000
g = torch.randn((n, d), device="cuda")
a = eng.gram(g); torch.cuda.synchronize()
idx = torch.tensor([0, 1, 127, 128, 129, 1000, 2047, 2048, 3999], device="cuda")
sub = g[idx].double()
ref = sub @ g.double().T if False else None
# full rows for a few i
for i in idx.tolist():
    r = (g[i].double()[None, :] @ g[:, :].double().T)[0] if False else None
ref = (sub @ sub.T)
got = a[idx][:, idx]
err = (got - ref).abs()
print(os.environ.get("BYZ_GRAM_MODE"), "max abs err", float(err.max()), "signed diag err", [round(float(x),3) for x in (got-ref).diag()[:5]], "offdiag signed", [round(float(x),3) for x in (got-ref)[0,1:6]], "ref diag", float(ref[0,0]))
# row check against fp64 for row 0 vs all columns in blocks
r0 = torch.zeros(n, dtype=torch.float64, device="cuda")
for c in range(0, d, 100000):
    r0 += (g[:, c:c+100000].double() @ g[0, c:c+100000].double())
e = (a[0] - r0).abs()
print("row0 max abs err", float(e.max()), "at", int(e.argmax()), "rel", float((e / r0.abs().clamp_min(1)).max()))
