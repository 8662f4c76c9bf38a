import numpy as np
rng=np.random.default_rng(1)
def make(M,L,sig):
    U,_=np.linalg.qr(rng.standard_normal((M,L))); V,_=np.linalg.qr(rng.standard_normal((L,L)))
    return (U*sig)@V.T, U@V.T
def ns(X, tol=1e-14, maxit=50):
    n=0
    G=X.T@X; b=min(np.abs(G).sum(1).max(), np.trace(G))
    if b>2: X=X/np.sqrt(b)*1.2
    while True:
        G=X.T@X; E=np.eye(G.shape[0])-G; err=np.abs(E).max(); einf=np.abs(E).sum(1).max()
        if err<tol: return X,n
        X=X@(np.eye(G.shape[0])+0.5*E); n+=1
        if einf<1e-7: return X,n
        if n>maxit: return X,n
def deflate(A, b=32, qmax=8, verbose=False):
    M,L=A.shape
    G=A.T@A; E=G-np.eye(L)
    V=np.sign(rng.standard_normal((L,b)))/np.sqrt(L)
    # orthonormalise start
    V,_=np.linalg.qr(V)
    passes=1
    for q in range(qmax):
        Y=E@V; passes+=1
        T=V.T@Y; T=0.5*(T+T.T)
        lam,Z=np.linalg.eigh(T)
        W=V@Z; YZ=Y@Z
        R=YZ-W*lam
        res=np.linalg.norm(R,axis=0)
        d=1/np.sqrt(1+lam)-1
        crit=np.abs(d)*res
        if verbose: print(q, "max |d| res", crit.max(), "lam range", lam.min(), lam.max())
        tolc=64*2.2e-16*max(1.0,np.abs(lam).max())
        if crit.max()<tolc: break
        # next basis: orthonormalise YZ with column scaling
        nrm=np.linalg.norm(YZ,axis=0); Ys=YZ/np.maximum(nrm,1e-300)
        S=Ys.T@Ys; th,Q=np.linalg.eigh(S)
        keep=th>1e-10*th.max()
        V=(Ys@Q[:,keep])/np.sqrt(th[keep])
        # second pass for safety
        S=V.T@V; th,Q=np.linalg.eigh(S); V=(V@Q)/np.sqrt(th)
        if V.shape[1]<b:
            extra=np.sign(rng.standard_normal((L,b-V.shape[1])))/np.sqrt(L)
            extra-=V@(V.T@extra); extra,_=np.linalg.qr(extra); V=np.hstack([V,extra])
    ok=crit<tolc
    X1=A+ (A@W[:,ok])*d[ok] @ W[:,ok].T
    return X1, passes, ok.sum(), lam[ok]
L=M=512
sig=np.ones(L)+3e-11*rng.standard_normal(L)
out=[4.684,1.2004,1.081,1.0364,1.0213,1.0125,1.0075,1.0036,0.99947,1+7.4e-6,1+1.1e-6,1+3.4e-7,1+1.1e-7,1+5e-8,1+1.3e-8,1+3.7e-9]
sig[:len(out)]=out
A,P=make(M,L,sig)
X,n=ns(A.copy()); print("plain NS steps",n,"err vs svd",np.abs(X-P).max())
X1,passes,k,lam=deflate(A,b=32,verbose=True)
print("deflated: passes over E",passes,"pairs",k)
E1=np.eye(L)-X1.T@X1; print("after deflation max|E|",np.abs(E1).max(),"einf",np.abs(E1).sum(1).max(), "spectral", np.abs(np.linalg.eigvalsh(E1)).max())
X2,n2=ns(X1); print("NS steps after deflation",n2,"err vs svd",np.abs(X2-P).max(), "orth", np.abs(X2.T@X2-np.eye(L)).max())
# a general (non low-rank) spectrum: fall back gracefully
sig=rng.uniform(0.7,1.3,L); A,P=make(M,L,sig)
X1,passes,k,lam=deflate(A,b=32)
X2,n2=ns(X1); Xp,n0=ns(A.copy()); print("wide spectrum: pairs",k,"NS after",n2,"vs plain",n0,"err",np.abs(X2-P).max())
