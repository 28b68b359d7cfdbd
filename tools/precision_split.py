#!/usr/bin/env python
"""Where does the Distance2Pre step need more than float32?  TEST INFRASTRUCTURE (CPU only, uses the oracle): one sequence of 50 positions
at dim 128 with the reference's init; the step is evaluated with float32 arithmetic in SOME of its parts and float64 in the rest, and
every variant is compared with the all-float64 one.  Parts: G = input product, F = forward recurrence, H = softmax head, B = BPTT chain,
X = dx / outer products; `store` = precision of the activations kept for the backward pass.  Result (three seeds): float32 G or F ->
2e-5 of the max-norm, float64 G + F with everything else float32 -> 2e-7: the design of csrc/te_xfwd.hip.
    python tools/precision_split.py [seed]"""
import os, sys, numpy as np, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import poi_oracle as O
F32=np.float32; F64=np.float64
def sig(x): return 1/(1+np.exp(-x))
def run(P,p,q,dp,dq,L,alpha,lam, tG, tF, tH, tB, tX, store=F32):
    """tG: dtype of input product, tF: forward chain, tH: head, tB: backward chain, tX: dx/outer products. returns lt update (touched rows), dense ui update"""
    lt,di,ui,wh,bi,vs,bs,wd=[P[k] for k in ('lt','di','ui','wh','bi','vs','bs','wd')]
    D=lt.shape[1]
    ls=O.softmax0(np.asarray(P['loss_weight'],F64))
    xs=np.concatenate((lt[p],di[dp]),1)
    n=L-1
    # G
    G=(xs[:n].astype(tG) @ ui.reshape(3*D,2*D).astype(tG).T).astype(tF) + bi.reshape(-1).astype(tF)
    whf=wh.astype(tF)
    hs=np.zeros((n+1,D),tF); zs=np.zeros((n,D),tF); rs=np.zeros((n,D),tF); cs=np.zeros((n,D),tF)
    for t in range(n):
        hp=hs[t]
        z=sig(G[t,:D]+whf[0]@hp); r=sig(G[t,D:2*D]+whf[1]@hp)
        c=np.tanh(G[t,2*D:]+whf[2]@(r*hp))
        zs[t],rs[t],cs[t]=z,r,c
        hs[t+1]=((1-z)*hp+z*c).astype(tF)
    # stored activations rounded to 'store'
    hsS=hs.astype(store); zsS=zs.astype(store); rsS=rs.astype(store); csS=cs.astype(store)
    # head
    H=hsS[1:].astype(tH)
    logits=H@vs.astype(tH).T+bs.astype(tH)
    S=np.exp(logits-logits.max(1,keepdims=True)); S/=S.sum(1,keepdims=True)
    a=dp[1:L]; b=dq[1:L]
    E=(lt[p[1:L]]-lt[q[1:L]]).astype(tH)
    idx=np.arange(n)
    us=(H*E).sum(1)+tH(wd)*(S[idx,a]-S[idx,b])
    gu=(-ls[1]*sig(-us)).astype(tH)
    ds=np.zeros_like(S)
    ds[idx,a]+=gu*tH(wd); ds[idx,b]-=gu*tH(wd)
    for t in range(n): ds[t,:a[t]+1]+=tH(ls[0])
    ds[idx,a]-=tH(ls[0])/S[idx,a]
    do=S*(ds-(ds*S).sum(1,keepdims=True))
    inj=(gu[:,None]*E + do@vs.astype(tH)).astype(tB)   # dh injection
    # backward chain
    whb=wh.astype(tB); uib=ui.astype(tX)
    DA=np.zeros((n,3*D),tB)
    dhn=np.zeros(D,tB)
    for t in range(n-1,-1,-1):
        dh=dhn+inj[t]
        z,r,c,hp=zsS[t].astype(tB),rsS[t].astype(tB),csS[t].astype(tB),hsS[t].astype(tB)
        dz=dh*(c-hp); dc=dh*z; dhp=dh*(1-z)
        dac=dc*(1-c*c)
        m=whb[2].T@dac
        dr=m*hp; dhp=dhp+m*r
        daz=dz*z*(1-z); dar=dr*r*(1-r)
        dhp=dhp+whb[0].T@daz+whb[1].T@dar
        DA[t,:D]=daz; DA[t,D:2*D]=dar; DA[t,2*D:]=dac
        dhn=dhp.astype(tB)
    DAx=DA.astype(tX)
    dx=DAx@ui.reshape(3*D,2*D).astype(tX)      # (n,2D)
    g_lt=np.zeros_like(lt); 
    np.add.at(g_lt,p[:n],dx[:,:D].astype(F64))
    gh=(gu[:,None]*H).astype(F64)
    np.add.at(g_lt,p[1:L],gh); np.add.at(g_lt,q[1:L],-gh)
    g_ui=(DAx.T@xs[:n].astype(tX)).astype(F64)
    return g_lt, g_ui, hs[-1].astype(F64), DA.astype(F64)
rng=np.random.default_rng(int(sys.argv[1]) if len(sys.argv)>1 else 0)
D=128; NB=200; N=2000; L=50
P=O.init_spatial_params(rng,N,NB,D)
for k in P:
    if isinstance(P[k],np.ndarray): P[k]=P[k].astype(F32).astype(F64)
p=rng.integers(0,N,L); q=rng.integers(0,N,L); dp=rng.integers(0,NB,L); dq=rng.integers(0,NB,L); dp[0]=NB
ref=run(P,p,q,dp,dq,L,0.01,0.001,F64,F64,F64,F64,F64,store=F64)
def rel(a,b): return np.abs(a-b).max()/np.abs(b).max()
print('max|g_lt|',np.abs(ref[0]).max())
for name,cfg in [('all f32',(F32,)*5+(F32,)),
                 ('G64 rest32',(F64,F32,F32,F32,F32,F32)),
                 ('G64 F64 rest32',(F64,F64,F32,F32,F32,F32)),
                 ('G32 F64 rest32',(F32,F64,F32,F32,F32,F32)),
                 ('G64 F64 H64 B32 X32',(F64,F64,F64,F32,F32,F32)),
                 ('fwd32 B64 X64 (G32,F32,H32)',(F32,F32,F32,F64,F64,F32)),
                 ('G64 F64 H32 B64 X32',(F64,F64,F32,F64,F32,F32)),
                 ('all64, store32',(F64,)*5+(F32,)),
                 ]:
    r=run(P,p,q,dp,dq,L,0.01,0.001,*cfg[:5],store=cfg[5])
    # error of update relative to max|lt| = 0.5 => alpha*|dg|/0.5
    print('%-32s g_lt rel %.2e  upd/maxnorm %.2e  g_ui rel %.2e  hT rel %.2e  DA rel %.2e'%(name, rel(r[0],ref[0]), 0.01*np.abs(r[0]-ref[0]).max()/0.5, rel(r[1],ref[1]), rel(r[2],ref[2]), rel(r[3],ref[3])))
