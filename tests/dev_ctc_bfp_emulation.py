"""Development check (imports oracle/: lives under tests/): numpy restatement of a block-floating-point CTC lattice (one integer
exponent per 4/8/16 adjacent states, linear arithmetic inside) against the fp64 oracle, random and sharp logits, F up to 1500.
The HIP version (tools/probes/ctc_block_floating_point.patch.txt) passes every CTC test and is NOT faster: the lattice kernel is
bound by the issue rate of one wave (~110 instructions per frame at one per 6-7 cycles), not by the arithmetic chain.  usage: python tests/dev_ctc_bfp_emulation.py"""
import numpy as np, sys
sys.path.insert(0,'/root/repo')
from oracle import ctc_ref
f32=np.float32
EMPTY=-(1<<20)
def emulate(logits,labels):
    F,V=logits.shape
    lp=(logits-np.log(np.exp(logits.astype(np.float64)).sum(-1,keepdims=True))).astype(f32)
    ext=[0]
    for l in labels: ext+=[int(l),0]
    S=len(ext); NL=64; SPL=4 if S<=256 else (8 if S<=512 else 16); NS=NL*SPL
    extp=np.array(ext+[0]*(NS-S))
    ok=(np.arange(NS)<S)
    skipA=np.array([bool(s<S and (s&1) and s>=2 and extp[s-2]!=extp[s]) for s in range(NS)])
    skipB=np.array([bool((s&1) and s+2<S and extp[s+2]!=extp[s]) for s in range(NS)])
    p=np.exp(lp).astype(f32)
    def sh(x,d): return np.ldexp(x,np.clip(d,-300,0)).astype(f32)
    def run(beta):
        rows=np.full((F,NS),-np.inf,f32)
        v=np.zeros((NL,SPL),f32); E=np.full(NL,EMPTY,np.int64)
        for i in range(F):
            t=F-1-i if beta else i
            pt=(p[t][extp]*ok).astype(f32).reshape(NL,SPL)
            if i==0:
                flat=np.zeros(NS,f32); pf=pt.reshape(-1)
                if beta:
                    flat[S-1]=pf[S-1]
                    if S>=2: flat[S-2]=pf[S-2]
                else:
                    flat[0]=pf[0]
                    if S>=2: flat[1]=pf[1]
                v=flat.reshape(NL,SPL); E[:]=0
            else:
                if not beta:
                    nv_=np.roll(v[:,SPL-1],1); ne=np.roll(E,1); nv_[0]=0; ne[0]=EMPTY
                    Ec=np.maximum(E,ne)
                    v=sh(v,(E-Ec)[:,None]); left=sh(nv_,ne-Ec); E=Ec
                    sk=skipA.reshape(NL,SPL); nv=np.zeros_like(v)
                    for k in range(SPL):
                        a=v[:,k]+(v[:,k-1] if k>=1 else left)
                        a=a+np.where(sk[:,k],(v[:,k-2] if k>=2 else left),f32(0))
                        nv[:,k]=a*pt[:,k]
                else:
                    r0=np.roll(v[:,0],-1); r1=np.roll(v[:,1],-1); ne=np.roll(E,-1); r0[-1]=0; r1[-1]=0; ne[-1]=EMPTY
                    Ec=np.maximum(E,ne)
                    v=sh(v,(E-Ec)[:,None]); r0=sh(r0,ne-Ec); r1=sh(r1,ne-Ec); E=Ec
                    sk=skipB.reshape(NL,SPL); nv=np.zeros_like(v)
                    for k in range(SPL):
                        a=v[:,k]+(v[:,k+1] if k+1<SPL else r0)
                        a=a+np.where(sk[:,k],(v[:,k+2] if k+2<SPL else (r0 if k+2==SPL else r1)),f32(0))
                        nv[:,k]=a*pt[:,k]
                v=nv.astype(f32)
            m=v.max(1)
            _,ke=np.frexp(m)
            E=np.where(m>0,E+ke,EMPTY)
            v=np.ldexp(v,np.where(m>0,-ke,0)[:,None]).astype(f32)
            with np.errstate(divide='ignore'):
                rows[t]=(np.log(v).astype(f32)+(np.where(m>0,E,0)[:,None]*0.6931471805599453).astype(f32)).reshape(-1)
        return rows
    A=run(False); B=run(True)
    a1,a2=A[F-1,S-1],A[F-1,S-2]
    m=max(a1,a2); ll=m+np.log(np.exp(a1-m)+np.exp(a2-m))
    grad=np.zeros((F,V),f32)
    for t in range(F):
        with np.errstate(invalid='ignore'):
            x=(A[t,:S]+B[t,:S]-lp[t][extp[:S]]-ll).astype(f32)
        term=np.where(np.isnan(x),0,np.exp(x))
        bins=np.zeros(V,f32); np.add.at(bins,extp[:S],term)
        grad[t]=np.exp(lp[t])-bins
    return -ll,grad
rng=np.random.default_rng(0)
for F,V,L,sharp in ((499,32,90,1),(1500,51,200,1),(60,8,7,1),(499,32,90,8),(700,20,300,1)):
    logits=(rng.standard_normal((1,F,V))*sharp).astype(np.float32)
    labels=rng.integers(1,V,(1,L))
    o_loss,o_grad,o_nll=ctc_ref.ctc_loss_and_grad(logits,labels,np.array([F],np.int32),0,"sum",True)
    nll,grad=emulate(logits[0],labels[0])
    print(F,V,L,sharp,"nll",nll,o_nll[0],"rel",abs(nll-o_nll[0])/abs(o_nll[0]),"grad err / max",np.abs(grad-o_grad[0]).max()/np.abs(o_grad).max())
