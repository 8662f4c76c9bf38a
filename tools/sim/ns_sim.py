import numpy as np
from scipy.optimize import brentq, minimize_scalar

def minimax_cubic(l, u):
    """odd cubic p(x)=a x + b x^3 minimising max_{[l,u]} |1-p(x)|: equioscillation p(l)=1-d, p(x*)=1+d, p(u)=1-d"""
    # p(l)=p(u) => a l + b l^3 = a u + b u^3 => a = -b (u^3-l^3)/(u-l) = -b (u^2+ul+l^2)
    # let b=-c (c>0): a = c*(u^2+u l+l^2); interior max at x*^2 = a/(3c) = (u^2+ul+l^2)/3
    q = u*u+u*l+l*l
    xs = np.sqrt(q/3)
    # p(x*) = c*(q xs - xs^3) = c*xs*(q - q/3) = c*xs*2q/3 ; p(l) = c*(q l - l^3)
    # 1+d = c*A, 1-d = c*B  => c = 2/(A+B)
    A = xs*2*q/3; B = q*l - l**3
    c = 2/(A+B)
    d = c*A-1
    return c*q, -c, d

def steps(sig, mode, l_est=None, verbose=False):
    sig = sig.copy()
    n=0
    hist=[]
    while True:
        E = np.abs(1-sig**2).max()
        hist.append(E)
        if E < 1e-14: break
        if mode=='ns' or E < 0.02:
            a,b = 1.5,-0.5
        else:
            l,u = sig.min(), sig.max()
            a,b,d = minimax_cubic(l,u)
        sig = a*sig + b*sig**3
        n+=1
        if E < 1e-7: break
        if n>50: break
    return n, hist

rng=np.random.default_rng(0)
L=4096
for name, sig in [
    ("bulk1+outliers3.4", np.concatenate([np.ones(L-32)+0.01*rng.standard_normal(L-32), np.linspace(1.05,3.4,32)])),
    ("bulk1+out1.5", np.concatenate([np.ones(L-32)+0.01*rng.standard_normal(L-32), np.linspace(1.05,1.5,32)])),
    ("unif .5-1.5", rng.uniform(0.5,1.5,L)),
    ("near (0.1 step)", 1+0.02*rng.standard_normal(L)),
    ("bulk1 + few small .3", np.concatenate([np.ones(L-8)+0.005*rng.standard_normal(L-8), np.linspace(0.3,0.9,8)])),
]:
    # classical: scale so max = 1.2 (round 3)
    s = sig/ (sig.max()/1.2) if sig.max()>np.sqrt(2) else sig
    n_ns,h = steps(s,'ns')
    n_mm,h2 = steps(sig,'mm')
    print(f"{name:28s} NS {n_ns}  minimax {n_mm}", ["%.1e"%x for x in h2])
