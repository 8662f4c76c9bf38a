"""Run by tests/test_gpu_configs.py::test_rccl_only_code_on_a_one_rank_communicator in a process of its own: a process group of ONE rank
on the RCCL backend, with sharded.FORCE_COLLECTIVES lifting the world == 1 short-cuts, so that the code no gloo test reaches really
executes on the GPU: all_reduce(async_op=True) on the communicator's stream, reduce_scatter_tensor with its padded blocks, all_gather of
the per-latent gradients, max-over-ranks on a device tensor, init_process_group("nccl", device_id=...).  Prints OK on success."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from multioutputihgp_amd import sharded  # noqa: E402
from multioutputihgp_amd.sharded import (ShardedMOIHGP, allreduce_nll, allreduce_nll_async, allreduce_total_async, gather_latent_grads,  # noqa: E402
                                         max_over_ranks, run_pipelined)


def main():
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)          # bench.py's N > 1 initialisation
    sharded.FORCE_COLLECTIVES = True
    assert dist.get_backend() == "nccl"
    rng = np.random.default_rng(3)
    # the 8-byte exchange, ordered and overlapped (two in flight), and the max-over-ranks timing on the device
    nll = torch.from_numpy(rng.standard_normal(1000)).to(dev)
    assert abs(allreduce_nll(nll).item() - nll.sum().item()) < 1e-12
    ring = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(4)]
    k = [0]

    def one_pass():
        t = ring[k[0] % 4]; k[0] += 1
        t.fill_(float(k[0]))
        return allreduce_total_async(t)

    tot = run_pipelined(7, one_pass, max_in_flight=2)
    assert [float(t.item()) for t in tot] == [1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0]
    assert abs(allreduce_nll_async(nll).wait().item() - nll.sum().item()) < 1e-12
    assert max_over_ranks(0.25, device=dev) == 0.25
    # per-latent gradients: all_gather of the (padded) blocks
    g = torch.from_numpy(rng.standard_normal((37, 3))).to(dev)
    assert torch.equal(gather_latent_grads(g, 37), g)
    # reduce-scatter form of the un-projection, with a time axis that needs the zero padding, through the real pipeline
    M, L, T = 24, 9, 301
    sh = ShardedMOIHGP(0.1, M, L, kernel="Matern32")
    p = np.concatenate([(np.eye(M, L) + 0.2 * rng.standard_normal((M, L))).ravel(), rng.uniform(0.5, 2, L), [0.04],
                        np.column_stack([rng.uniform(0.5, 2, L), rng.uniform(0.5, 2, L), rng.uniform(0.05, 0.2, L)]).ravel()])
    sh.update(p)
    Y = rng.standard_normal((T, M))
    Yd = torch.from_numpy(Y).to(dev)
    Ya, nll_a = sh.filter(Yd)                      # all-reduce form (forced: the collective runs)
    Ys, nll_s = sh.filter(Yd, scatter=True)        # reduce_scatter_tensor
    torch.cuda.synchronize()
    assert Ys.shape == Ya.shape and float((Ys - Ya).abs().max()) < 1e-12 and abs(nll_a - nll_s) < 1e-9 * abs(nll_a)
    part = torch.from_numpy(rng.standard_normal((2, T, M))).to(dev)
    assert torch.equal(sh._reduce_scatter_rows(part.clone()), part)
    # missing outputs across "shards": the Woodbury correction with its all-reduce of the k x k systems (forced)
    Yn = Y.copy(); Yn[3, [1, 7]] = np.nan; Yn[100, 5] = np.nan
    Yh_n, _ = sh.filter(torch.from_numpy(Yn).to(dev))
    from oracle import cref
    ref = cref.GP(0.1, M, L, "Matern32"); ref.update(p)
    x = np.zeros((L, 2)); want = np.empty((T, M))
    for t in range(T):
        x, want[t] = ref.step(x, Yn[t])
    torch.cuda.synchronize()
    assert np.abs(Yh_n.cpu().numpy() - want).max() < 1e-8 * np.abs(want).max()
    dist.destroy_process_group()
    print("OK")


if __name__ == "__main__":
    main()
