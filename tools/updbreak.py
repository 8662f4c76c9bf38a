#!/usr/bin/env python3
"""Where the time of MOIHGP::update at M = L = 4096 goes: host -> device staging of the 134 MB parameter vector against the device work."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multioutputihgp_amd import MOIHGP
M = L = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rng = np.random.default_rng(0)
gp = MOIHGP(0.1, M, L, kernel="Matern52ss")
p = gp.params.copy()
dU = rng.standard_normal(M * L); p[:M * L] += 0.1 * dU / np.linalg.norm(dU)
def t(f, n=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
a = torch.from_numpy(p[:M * L])
d = torch.empty(M * L, dtype=torch.float64, device="cuda")
pin = torch.empty(M * L, dtype=torch.float64).pin_memory(); pin.copy_(a)
print(f"pageable -> device {8 * M * L / 1e6:.0f} MB: {t(lambda: d.copy_(a)):.2f} ms; pinned -> device: {t(lambda: d.copy_(pin, non_blocking=True)):.2f} ms; host memcpy into pinned: {t(lambda: pin.copy_(a)):.2f} ms")
print(f"gp.update(params): {t(lambda: gp.update(p)):.2f} ms")
pp = np.frombuffer(pin.numpy(), dtype=np.float64)  # params vector whose U part lives in pinned memory
q = torch.empty(p.size, dtype=torch.float64).pin_memory(); q.copy_(torch.from_numpy(p)); qn = q.numpy()
print(f"gp.update(params in page-locked memory): {t(lambda: gp.update(qn)):.2f} ms")
