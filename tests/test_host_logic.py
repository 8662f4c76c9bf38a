"""CPU suite: host-side logic (sharding arithmetic, alignment contract) and the N>1 reduction path on gloo."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from multioutputihgp_amd.sharded import allreduce_nll, allreduce_nll_async, gather_latent_grads, shard_bounds
from multioutputihgp_amd.streams import padded_len


def test_shard_bounds_partition():
    for L in (1, 7, 256, 4096, 32768, 1001):
        for ws in (1, 2, 3, 4, 8):
            spans = [shard_bounds(L, ws, r) for r in range(ws)]
            assert spans[0][0] == 0 and spans[-1][1] == L
            assert all(spans[i][1] == spans[i + 1][0] for i in range(ws - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_padded_len():
    assert padded_len(10, torch.float32) == 12 and padded_len(12, torch.float32) == 12
    assert padded_len(9, torch.float64) == 10 and padded_len(0, torch.float64) == 0


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, ws, port, L, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    rng = np.random.default_rng(123)
    nll_all = rng.standard_normal(L)           # identical on all ranks (same seed)
    grad_all = rng.standard_normal((L, 3))
    lo, hi = shard_bounds(L, ws, rank)
    tot = allreduce_nll(torch.from_numpy(nll_all[lo:hi].copy()))
    g = gather_latent_grads(torch.from_numpy(grad_all[lo:hi].copy()), L)
    ok = abs(tot.item() - nll_all.sum()) < 1e-12 * max(1, abs(nll_all.sum())) and np.array_equal(g.numpy(), grad_all)
    # the overlapped form: several reductions in flight, the per-latent buffer reused right after each call
    buf = torch.from_numpy(nll_all[lo:hi].copy())
    pend = []
    for k in range(4):
        pend.append(allreduce_nll_async(buf * (k + 1)))
        buf = buf.clone()
    for k, p_ in enumerate(pend):
        ok = ok and abs(p_.wait().item() - (k + 1) * nll_all.sum()) < 1e-12 * max(1, abs(nll_all.sum()) * (k + 1))
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


@pytest.mark.parametrize("L", [5, 64])
def test_nll_allreduce_and_grad_gather_gloo_world2(L):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, L, q)) for r in range(2)]
    for p in procs: p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs: p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]
