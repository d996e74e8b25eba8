import sys, torch, math
sys.path.insert(0, "/root/repo")
from sprc_amd import _lib as L, engine as E
DEV="cuda:0"
def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed); return torch.randn(shape, generator=g) * scale
def _x3(x):
    hi = x.to(torch.float16); lo = (x - hi.float()).to(torch.float16); return hi, lo
M, D, F_ = 300, 768, 3072
y32 = _rand((M, D), 1)
W1, b1 = _rand((F_, D), 4, 0.05), _rand((F_,), 5, 0.1)
hi, lo = _x3(y32); y3 = torch.cat([hi, lo, hi], 1).contiguous().to(DEV)
h, l_ = _x3(W1); w3 = torch.cat([h, h, l_], 1).contiguous().to(DEV)
for act in (L.ACT_NONE, L.ACT_GELU):
    o32 = E.gemm(y3, w3, bias=b1.to(DEV), out_dtype=L.SPRC_F32, act=act).cpu()
    hid3 = torch.zeros((M, 3 * F_), dtype=torch.float16, device=DEV)
    E.gemm(y3, w3, bias=b1.to(DEV), out_dtype=L.SPRC_F16X3, act=act, out=hid3, ldc=3 * F_)
    h3 = hid3.cpu()
    got = h3[:, :F_].double() + h3[:, F_:2*F_].double()
    z = y32.double() @ W1.double().t() + b1.double()
    want = torch.nn.functional.gelu(z) if act else z
    d32 = (o32.double() - want).abs(); dx3 = (got - want).abs(); dd = (got - o32.double()).abs()
    print(f"act {act}: fp32-out err max {d32.max():.2e}; x3-out err max {dx3.max():.2e}; x3 vs fp32-out max {dd.max():.2e}; count>4e-5: {(dx3>4e-5).sum().item()} / {(d32>4e-5).sum().item()}")
    i = dx3.argmax().item(); r, c = divmod(i, F_)
    print("  worst:", r, c, "z", z[r,c].item(), "want", want[r,c].item(), "o32", o32[r,c].item(), "hi", h3[r,c].item(), "lo", h3[r,F_+c].item())
    bad = (dx3 > 4e-5).nonzero()[:8]
    for r, c in bad.tolist():
        print("   ", r, c, "z", round(z[r,c].item(),5), "want", want[r,c].item(), "o32", o32[r,c].item(), "hi", h3[r,c].item(), "lo", h3[r,F_+c].item())
