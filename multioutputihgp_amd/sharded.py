"""Latent sharding over the GPUs of one node: one process per GPU, `torch.distributed` (backend "nccl"
is RCCL on ROCm; "gloo" for the CPU rehearsal of the host logic).

After the OILMM projection the L latent processes are independent (reference moihgp.h:217-221 is a plain
loop with no cross-latent term), so rank r owns the contiguous latents [lo_r, hi_r), runs its recursion with
no data-path collective, and only the scalar sum of per-latent NLLs -- the quantity L-BFGS consumes
(moihgp.h:684 `loss += ...`) -- is all-reduced: 8 bytes over xGMI, pure latency.
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist


# Test hook (tests/test_gpu_configs.py, bench.py BENCH_FORCE_DIST=1): with a process group of ONE rank the helpers below normally return
# early; set to True they issue their collectives anyway, so that the RCCL-only code (async all-reduce on the communicator's stream,
# reduce_scatter_tensor with its padding, all_gather) runs on a one-GPU box.
FORCE_COLLECTIVES = False


def _collective(group=None) -> bool:
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or FORCE_COLLECTIVES)


def shard_bounds(L: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous balanced partition of L latents; the first L % world_size ranks get one extra."""
    q, r = divmod(L, world_size)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def allreduce_total(total: torch.Tensor, group=None) -> torch.Tensor:
    """All-reduce (SUM) of a rank's fp64 NLL total -- a 1-element tensor, e.g. the `nll_total` of LatentBank.filter -- in place."""
    if _collective(group):
        if total.is_cuda and dist.get_backend(group) == "gloo":      # CPU rehearsal of the exchange
            t = total.cpu()
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            total.copy_(t)
            return total
        dist.all_reduce(total, op=dist.ReduceOp.SUM, group=group)
    return total


def allreduce_nll(nll_local: torch.Tensor, group=None) -> torch.Tensor:
    """Sum of the per-latent NLLs over all shards: one fp64 scalar all-reduce (SUM)."""
    return allreduce_total(nll_local.sum(dtype=torch.float64).reshape(1), group)


class PendingSum:
    """Handle of an all-reduce in flight (allreduce_nll_async): `.wait()` returns the reduced fp64 scalar tensor."""

    def __init__(self, total, work=None, host=None):
        self._total, self._work, self._host = total, work, host

    def wait(self) -> torch.Tensor:
        if self._work is not None:
            self._work.wait()
            self._work = None
            if self._host is not None:
                self._total = self._host.to(self._total.device)
        return self._total


def allreduce_total_async(total: torch.Tensor, group=None) -> PendingSum:
    """As allreduce_total, but the collective runs on the communicator's own stream and the caller's stream does not wait for it.
    `total` must not be rewritten before `.wait()` (use a small ring of totals for passes in flight)."""
    if _collective(group):
        if total.is_cuda and dist.get_backend(group) == "gloo":      # CPU rehearsal of the exchange
            t = total.cpu()
            return PendingSum(total, dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True), host=t)
        return PendingSum(total, dist.all_reduce(total, op=dist.ReduceOp.SUM, group=group, async_op=True))
    return PendingSum(total)


def allreduce_nll_async(nll_local: torch.Tensor, group=None) -> PendingSum:
    """As allreduce_nll, but the collective runs on the communicator's own stream and the caller's stream does not wait for it:
    the next sweep overlaps the 8-byte exchange of this one (slabs of a long stream, pipelined objective evaluations).  The
    local sum is taken in stream order first, so the per-latent buffer may be overwritten right away."""
    return allreduce_total_async(nll_local.sum(dtype=torch.float64).reshape(1), group)


def run_pipelined(steps: int, one_pass_async, max_in_flight: int = 2):
    """The timed loop of the N-GPU bench (bench.py): every pass ends in its own all-reduce of the NLL scalar, but the exchange of pass k
    runs while pass k + 1 sweeps -- at most `max_in_flight` reductions outstanding, all of them complete on return.
    `one_pass_async()` launches one pass and returns a PendingSum.  Returns the list of reduced totals in pass order."""
    pending, totals = [], []
    for _ in range(steps):
        pending.append(one_pass_async())
        if len(pending) > max_in_flight:
            totals.append(pending.pop(0).wait().clone())     # (the caller's ring slot is reused by a later pass)
    for p_ in pending:
        totals.append(p_.wait().clone())
    return totals


def max_over_ranks(seconds: float, device="cpu", group=None) -> float:
    """Wall time of the slowest rank (the bench contract: barrier + synchronize on both sides, then the MAX over ranks)."""
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    if _collective(group):
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def time_slice_bounds(T: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Ticks [lo, hi) of the prediction stream a rank keeps after the reduce-scatter form of the un-projection (same balanced partition
    as shard_bounds, over time instead of latents)."""
    return shard_bounds(T, world_size, rank)


def gather_latent_grads(grad_local: torch.Tensor, L: int, group=None) -> torch.Tensor:
    """Per-latent gradients are disjoint across shards: all-gather [L_r, P] blocks into [L, P]
    (mode G only; moihgp.h:608-609 packs them latent-major)."""
    if not _collective(group):
        return grad_local
    ws = dist.get_world_size(group)
    P = grad_local.shape[1]
    sizes = [shard_bounds(L, ws, r)[1] - shard_bounds(L, ws, r)[0] for r in range(ws)]
    mx = max(sizes)
    pad = torch.zeros((mx, P), dtype=grad_local.dtype, device=grad_local.device)
    pad[: grad_local.shape[0]] = grad_local
    out = [torch.empty_like(pad) for _ in range(ws)]
    dist.all_gather(out, pad, group=group)
    return torch.cat([o[:n] for o, n in zip(out, sizes)], dim=0)


def missing_output_correction(U_r, sqrtS_r, Ty_r, missing, allreduce, lmax=None, chunk_elems=2 ** 28):
    """Least-squares projection of partially observed ticks (moihgp.h:167-178, `(U0^T U0).ldlt().solve(U0^T y_obs)`) when the latents
    are split over ranks.  (U0^T U0)^-1 couples ALL latents, so a rank's column slice alone gives a different vector; but U is a polar
    factor (U^T U = I), hence U0^T U0 = I - Um^T Um with Um the k rows of U at the missing outputs, and by Woodbury
        a = r + Um^T (I_k - Um Um^T)^-1 Um r,      r = U^T (y with NaN -> 0).
    r and the correction are column-local; the k x k Gram matrix G = Um Um^T and the k-vector b = Um r are sums over the column shards:
    ONE all-reduce of sum_t (k_t + k_t^2) doubles for all affected ticks of the stream, then every rank solves the same small systems
    and corrects its own rows.  (The unsharded device path does the same per tick: csrc/tick.hip ls_project_kernel; the sharded device
    path is the kernel pair moihgp_ls_shard_gram / moihgp_ls_shard_apply, which ShardedMOIHGP uses -- this function is their restatement on
    torch tensors, kept for CPU tensors: the gloo tests of the host logic.)

    U_r [M, L_r] fp64: this rank's columns;  sqrtS_r [L_r];  Ty_r [L_r, >= T]: S_r^-1/2 U_r^T y0_t in column t, corrected IN PLACE;
    missing [T, M] bool;  allreduce: tensor -> its sum over the ranks.
    lmax: the width of the WIDEST shard, ceil(L / world) -- the same number on every rank.  The affected ticks are processed in chunks
    (<= chunk_elems gathered doubles at a time) and every chunk is one all-reduce, so the chunking must not depend on the local shard
    width: shard_bounds hands out widths that differ by one, and ranks that cut the ticks differently would call the collective a
    different number of times with different shapes.  Without lmax the chunk size falls back to a fixed tick count."""
    aff = missing.any(dim=1).nonzero().flatten()
    if aff.numel() == 0:
        return Ty_r
    kmax = int(missing[aff].sum(dim=1).max())
    # <= 2 GiB of gathered rows at a time; derived from rank-invariant quantities only (kmax comes from `missing`, replicated)
    step = max(1, int(chunk_elems // max(1, kmax * int(lmax)))) if lmax is not None else max(1, int(chunk_elems // max(1, kmax * 65536)))
    eye = torch.eye(kmax, dtype=torch.float64, device=U_r.device)[None]
    for a in range(0, aff.numel(), step):
        ticks = aff[a:a + step]
        flag, order = torch.sort(missing[ticks].to(torch.int8), dim=1, descending=True, stable=True)
        idx, valid = order[:, :kmax], flag[:, :kmax].bool()                   # the missing outputs of every tick, padded to kmax
        Um = U_r[idx] * valid[..., None]                                      # [n, kmax, L_r]; padding rows are zero
        r = Ty_r[:, ticks].double().T * sqrtS_r[None, :]                      # [n, L_r] = U_r^T y0
        b = torch.bmm(Um, r[..., None])[..., 0]                               # [n, kmax]
        G = torch.bmm(Um, Um.transpose(1, 2))                                 # [n, kmax, kmax]
        n = ticks.numel()
        packed = allreduce(torch.cat([b, G.reshape(n, -1)], dim=1))           # the path's extra exchange: n (kmax + kmax^2) doubles
        b, G = packed[:, :kmax], packed[:, kmax:].reshape(n, kmax, kmax)
        w = torch.linalg.solve(eye - G, b[..., None])                         # padding rows / columns: identity
        corr = torch.bmm(Um.transpose(1, 2), w)[..., 0] / sqrtS_r[None, :]    # [n, L_r]
        Ty_r[:, ticks] += corr.T.to(Ty_r.dtype)
    return Ty_r


class ShardedMOIHGP:
    """The whole-stream pipeline of one model sharded over the ranks of a process group, one process per GPU
    (SURVEY 8e / 8f N1).

    Every rank holds the full parameter vector.  `update(params)` computes the global orthonormal factor U (the polar
    factor of moihgp.h:433-447; redundantly on every rank: deterministic, no broadcast of 8*M*L bytes) and hands this rank's
    latent columns U[:, lo:hi], S[lo:hi] and per-latent triples to a shard object.  `filter(Y)` then runs, per rank,
        Ty_r   = S_r^-1/2 U_r^T Y^T          local GEMM on the replicated observation stream Y [T, M]       (moihgp.h:181)
        sweep  over its latents               no communication                                               (ihgp.h:81-93, :204-209)
        Yhat_r = (U_r S_r^1/2 Tyhat_r)^T      local GEMM: this rank's PARTIAL prediction [T, M]              (moihgp.h:222-225)
    and combines  Yhat = sum_r Yhat_r  with one all-reduce of T*M elements (bandwidth-relevant: per-link bound on xGMI rings)
    and the NLL with the 8-byte all-reduce.  The three global NLL terms of moihgp.h:653 need U U^T y, i.e. another reduction
    of the same shape; they are evaluated from the already reduced quantities on every rank.
    """

    def __init__(self, dt, num_output, num_latent, kernel="Matern52ss", group=None, seed=20260101):
        from ._lib import load_library
        from .pywrapper import MOIHGP
        self.group = group
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        if self.world > num_latent:
            raise ValueError(f"ShardedMOIHGP: {self.world} ranks for {num_latent} latents -- every rank needs at least one latent")
        self.M, self.L = num_output, num_latent
        self.lo, self.hi = shard_bounds(num_latent, self.world, self.rank)
        self._full = MOIHGP(dt, num_output, num_latent, kernel=kernel)          # polar factor + bookkeeping of all parameters
        # the constructor draws U from std::random_device (moihgp.h:105-113): re-draw it from a fixed seed so that every rank
        # holds the SAME orthonormal factor before the first update() (a filter() on freshly built objects is then consistent)
        load_library().moihgp_reseed_U(self._full.handle, int(seed))
        self._shard = MOIHGP(dt, num_output, self.hi - self.lo, kernel=kernel)  # this rank's latent columns
        self._sync_shard()

    def _sync_shard(self):
        import ctypes as C
        import numpy as np
        from ._lib import c_double_p, load_library
        lib = load_library()
        p = self._full.params
        M, L = self.M, self.L
        U = p[:M * L].reshape(M, L)
        Us = np.ascontiguousarray(U[:, self.lo:self.hi])
        Ss = np.ascontiguousarray(p[M * L + self.lo:M * L + self.hi])
        self.sigma = float(p[M * L + L])
        self.S = p[M * L:M * L + L].copy()
        igp = np.ascontiguousarray(p[M * L + L + 1:].reshape(L, self._full.num_igp_param)[self.lo:self.hi])
        if lib.moihgp_set_mixing(self._shard.handle, Us.ctypes.data_as(c_double_p), Ss.ctypes.data_as(c_double_p), C.c_double(self.sigma)):
            raise RuntimeError("moihgp_set_mixing failed")
        if lib.moihgp_update_latents(self._shard.handle, igp.ctypes.data_as(c_double_p)):
            raise RuntimeError("moihgp_update_latents failed")

    def update(self, params):
        self._full.update(params)
        self._sync_shard()

    @property
    def params(self):
        return self._full.params

    def _allreduce(self, t):
        if self.world == 1 and not FORCE_COLLECTIVES:
            return t
        if t.is_cuda and dist.get_backend(self.group) == "gloo":     # CPU rehearsal of the exchange
            c = t.cpu()
            dist.all_reduce(c, op=dist.ReduceOp.SUM, group=self.group)
            return c.to(t.device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def _reduce_scatter_rows(self, t):
        """[2, T, M] partial slabs -> this rank's time slice [2, T_r, M] of their sum over ranks.  RCCL: one reduce-scatter (half the
        bytes of an all-reduce on the xGMI ring: every rank receives only the rows it keeps); gloo (CPU rehearsal of the host logic)
        has no reduce-scatter: all-reduce, then slice."""
        T = t.shape[1]
        lo, hi = time_slice_bounds(T, self.world, self.rank)
        if self.world == 1 and not FORCE_COLLECTIVES:
            return t
        if dist.get_backend(self.group) == "gloo":
            c = t.cpu() if t.is_cuda else t
            dist.all_reduce(c, op=dist.ReduceOp.SUM, group=self.group)
            return c[:, lo:hi].to(t.device)
        # equal blocks for reduce_scatter_tensor: pad the time axis to world * ceil(T / world) rows (rows past T are zero in every rank)
        per = (T + self.world - 1) // self.world
        pad = torch.zeros((self.world, 2, per, t.shape[2]), dtype=t.dtype, device=t.device)
        for r in range(self.world):
            a, b = time_slice_bounds(T, self.world, r)
            pad[r, :, :b - a] = t[:, a:b]
        out = torch.empty((2, per, t.shape[2]), dtype=t.dtype, device=t.device)
        dist.reduce_scatter_tensor(out, pad, op=dist.ReduceOp.SUM, group=self.group)
        return out[:, :hi - lo]

    def _project_with_missing_outputs(self, Y, missing):
        """Projection of a stream whose observation vectors hold missing outputs, latents split over the ranks: the column-local
        projection of (y with NaN -> 0), then `missing_output_correction` (one small all-reduce) on the affected ticks."""
        from .streams import project_stream
        Ty = project_stream(self._shard, torch.where(missing, torch.zeros((), dtype=Y.dtype, device=Y.device), Y))
        kmax = int(missing.sum(dim=1).max())
        if kmax > 64 or self.M - kmax < self.L:
            raise NotImplementedError(f"ShardedMOIHGP.filter: a tick with {kmax} of {self.M} outputs missing (limit: 64, and at least "
                                      f"{self.L} observed) needs the per-tick path")
        # the correction as a HIP kernel pair around the all-reduce (csrc/tick.hip ls_shard_gram_kernel / ls_shard_apply_kernel; the torch
        # function above is its CPU restatement, used by the gloo host-logic tests): per affected tick this rank's part of
        # [U_miss r | U_miss U_miss^T], one all-reduce of all the records, then the k x k solves and the correction of this rank's rows
        import ctypes as C
        from ._lib import last_error, load_library
        lib = load_library()
        aff = missing.any(dim=1).nonzero().flatten().to(torch.int32)
        dt = 0 if Y.dtype == torch.float64 else 1
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        rec = kmax + kmax * kmax
        step = max(1, (1 << 26) // rec)                                   # <= 512 MiB of records at a time; n and kmax are the same on every rank
        for a in range(0, aff.numel(), step):
            ticks = aff[a:a + step].contiguous()
            packed = torch.empty((ticks.numel(), rec), dtype=torch.float64, device=Y.device)
            if lib.moihgp_ls_shard_gram(self._shard.handle, dt, C.c_void_p(Y.data_ptr()), C.c_void_p(ticks.data_ptr()), ticks.numel(), kmax,
                                        C.c_void_p(Ty.data_ptr()), Ty.stride(0), C.c_void_p(packed.data_ptr()), stream):
                raise RuntimeError(last_error(lib) or "moihgp_ls_shard_gram failed")
            packed = self._allreduce(packed)                              # the path's extra exchange: n (kmax + kmax^2) doubles
            if lib.moihgp_ls_shard_apply(self._shard.handle, dt, C.c_void_p(Y.data_ptr()), C.c_void_p(ticks.data_ptr()), ticks.numel(), kmax,
                                         C.c_void_p(packed.data_ptr()), C.c_void_p(Ty.data_ptr()), Ty.stride(0), stream):
                raise RuntimeError(last_error(lib) or "moihgp_ls_shard_apply failed")
        return Ty

    def filter(self, Y: torch.Tensor, scatter: bool = False):
        """Y [T, M] (CUDA, fp32/fp64, replicated on every rank).  Returns (Yhat, nll_total) where nll_total is the sum over ticks of
        MOIHGP::negLogLikelihood(x, y) (moihgp.h:614-688) along the filtered trajectory.
        scatter=False: Yhat [T, M] on every rank (one all-reduce of the 2 T M partial elements).
        scatter=True : Yhat [T_r, M], this rank's time slice `time_slice_bounds(T, world, rank)` only -- the partial predictions are
                       combined by a reduce-scatter (half the ring traffic of the all-reduce) and the residual term of moihgp.h:651 is
                       formed on the local rows, then summed with the NLL scalar."""
        from .streams import LatentBank, project_stream, unproject_stream
        import math
        # the kernels behind this method take Y's base pointer and assume [T][M] rows: a view or a slice would be read wrongly
        if not (isinstance(Y, torch.Tensor) and Y.is_cuda and Y.dim() == 2 and Y.shape[1] == self.M and Y.dtype in (torch.float32, torch.float64)):
            raise ValueError(f"Y must be a CUDA tensor [T, {self.M}] of float32 or float64")
        if not Y.is_contiguous():
            Y = Y.contiguous()
        T = Y.shape[0]
        missing = torch.isnan(Y)
        if (self.world > 1 or FORCE_COLLECTIVES) and bool(missing.any()):
            Ty = self._project_with_missing_outputs(Y, missing)
        else:
            Ty = project_stream(self._shard, Y)
        bank = LatentBank.from_handle(self._shard)
        yhat_lat, _, nll = bank.filter(Ty, T=T)
        part = unproject_stream(self._shard, yhat_lat, T)                 # this rank's partial prediction
        uuty = unproject_stream(self._shard, Ty, T)                       # this rank's part of U U^T y (same GEMM shape)
        m_n = max(float(self.M - self.L), 0.0)
        glob_const = T * (0.5 * math.log(float(self.S.sum())) + 0.5 * m_n * math.log(self.sigma))
        if not scatter:
            both = self._allreduce(torch.stack([part, uuty]))             # one collective for both [T, M] slabs
            Yhat, UUty = both[0], both[1]
            nll_lat = allreduce_nll(nll, self.group)
            resid = (Y - UUty).double().norm(dim=1)                       # ||(I - U U^T) y_t||, un-squared (moihgp.h:651)
            return Yhat, float(nll_lat.item()) + glob_const + 0.5 * float(resid.sum()) / self.sigma
        lo, hi = time_slice_bounds(T, self.world, self.rank)
        both = self._reduce_scatter_rows(torch.stack([part, uuty]))
        Yhat, UUty = both[0], both[1]
        resid_local = (Y[lo:hi] - UUty).double().norm(dim=1).sum()
        scal = nll.sum(dtype=torch.float64).reshape(1) + 0.5 * resid_local.reshape(1) / self.sigma   # per-latent NLLs + this slice's residual term
        tot = allreduce_total(scal, self.group)
        return Yhat, float(tot.item()) + glob_const
