import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multioutputihgp_amd import streams as S
L, T = 517, 1025
for dtype in (torch.float64, torch.float32):
  for kern in ("Matern52ss", "Matern32"):
    rng = np.random.default_rng(17 * L + T)
    prm = np.column_stack([rng.uniform(0.5, 2, L), rng.uniform(0.5, 2, L), rng.uniform(0.05, 0.2, L)])
    bank = S.LatentBank(0.1, prm, kernel=kern)
    t = np.arange(T)[None, :]; l = np.arange(L)[:, None]
    Ty = np.sin(0.05 * t * (1 + l % 7)) + 0.1 * rng.standard_normal((L, T))
    dev = S.alloc_stream(L, T, dtype); dev.zero_(); dev[:, :T] = torch.from_numpy(Ty).to(dtype)
    x0 = torch.from_numpy(0.2 * rng.standard_normal((L, bank.d))).to(dtype).cuda()
    for rep in range(3):
        ya, xa, na = bank.filter(dev, T=T, x=x0.clone())
        yt, xb, nb = bank.filter_tiled(S.tile_stream(dev, T), T, x=x0.clone())
        yb = S.untile_stream(yt, T)
        torch.cuda.synchronize()
        a = ya[:, :T].cpu().numpy(); b = yb[:, :T].cpu().numpy()
        d = np.argwhere(a != b)
        print(dtype, kern, "rep", rep, "mismatches", len(d), "first", d[:5].tolist(), "maxabs", float(np.abs(a - b).max()), "x eq", bool((xa == xb).all()), "nll eq", bool((na == nb).all()))
        if len(d):
            i, j = d[0]; print("   values", a[i, j], b[i, j], "latents with mismatch", sorted(set(d[:, 0].tolist()))[:10], "tick range", d[:, 1].min(), d[:, 1].max())
