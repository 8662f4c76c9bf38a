"""The C++ host surface moihgp::MOIHGP<SS> (include/moihgp_cxx/moihgp.hpp): compiles and links against libmoihgp.so
on CPU; on the GPU box it is driven through a closed loop and compared with the oracle."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, rel_err

EXE = os.path.join(ROOT, "build", "cxx_surface")


def _build(hip_built):
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    libdir = os.path.dirname(hip_built)
    subprocess.run(["g++", "-std=c++14", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cxx", "cxx_surface.cpp"),
                    "-o", EXE, "-L", libdir, "-lmoihgp", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return EXE


def test_cxx_surface_compiles_and_links(hip_built):
    assert os.path.exists(_build(hip_built))


@pytest.mark.gpu
@pytest.mark.parametrize("kern,M,L", [("Matern32", 5, 2), ("Matern52", 7, 4)])
def test_cxx_surface_closed_loop_vs_oracle(hip_built, kern, M, L):
    from oracle import cref
    exe = _build(hip_built)
    rng = np.random.default_rng(M * 10 + L)
    ref = cref.GP(0.1, M, L, kern); ref.set_literal_ugrad(0)
    d, P = ref.igp_dim, 3
    params = np.concatenate([(np.eye(M, L) + 0.2 * rng.standard_normal((M, L))).ravel(), rng.uniform(0.5, 2, L), [0.05],
                             np.column_stack([rng.uniform(0.5, 2, L), rng.uniform(0.5, 2, L), rng.uniform(0.05, 0.2, L)]).ravel()])
    x = rng.standard_normal((L, d)); dx = rng.standard_normal((L, P, d)); nt = 4
    Y = rng.standard_normal((nt, M))
    inp = f"{M} {L} {0 if kern == 'Matern32' else 1} 0.1\n" + " ".join(repr(float(v)) for v in params) + "\n"
    inp += " ".join(repr(float(v)) for v in x.ravel()) + "\n" + " ".join(repr(float(v)) for v in dx.ravel()) + f"\n{nt}\n"
    inp += "\n".join(" ".join(repr(float(v)) for v in y) for y in Y) + "\n"
    out = subprocess.run([exe], input=inp, capture_output=True, text=True, check=True).stdout.strip().split("\n")
    assert [int(v) for v in out[0].split()] == [d, P, ref.num_param]
    ref.update(params)
    assert rel_err(np.array(out[1].split(), dtype=float), ref.params) < 1e-10
    k = 2
    for t in range(nt):
        l1, g1 = ref.negLogLikelihood(x, Y[t], dx); l2 = ref.negLogLikelihood(x, Y[t])
        xn, yh, dxn = ref.step(x, Y[t], dx)
        got = [float(v) for v in out[k].split()]
        assert abs(got[0] - l1) < 1e-9 * abs(l1) and abs(got[1] - l2) < 1e-9 * abs(l2)
        assert rel_err(np.array(out[k + 1].split(), dtype=float), g1) < 1e-8
        assert rel_err(np.array(out[k + 2].split(), dtype=float), yh) < 1e-9
        x, dx = xn, dxn; k += 3
    xp, _ = ref.step(x)
    assert rel_err(np.array(out[k].split(), dtype=float), xp.ravel()) < 1e-9
