#!/usr/bin/env python3
"""Host model of csrc/stem.hip's index arithmetic (written BEFORE the kernel's first run on a device, to spend no GPU
minutes on indexing mistakes): the packed A operand, the LDS image of a block, the B fragments a lane reads, the
32x32x16 MFMA's operand / result layouts as the other kernels of this library use them, the running row maximum, the
wave-wide shifts of the x-pooling and the stores -- in numpy, against conv2d + relu + max_pool2d.  Not a test of the
library (tests/test_stem_gpu.py is); kept because design/dense.md refers to it."""
import numpy as np, torch, torch.nn.functional as F, sys
R=4; STRIP=15; WAVES=4; CONVROWS=2*R+1; INROWS=2*(CONVROWS-1)+7; ROWS=INROWS*3; ROWDW=160; USED=2*WAVES*STRIP+5; STEPS=11
def pack(w,bias):
    dst=np.zeros((STEPS,2,64,8),np.float32)
    for s in range(STEPS):
      for t in range(2):
        for lane in range(64):
          m=lane&31; hi=lane>>5
          co=32*t+16*((m>>2)&1)+4*(m>>3)+(m&3); r=2*s+hi
          for e in range(8):
            v=0.0
            if r<21:
              ky=r//3; c=r%3
              if e>=1: v=w[co,c,ky,e-1]
            elif e==0 and bias is not None: v=bias[co]
            dst[s,t,lane,e]=v
    return dst
def mfma(A,Bm,acc):
    # A[lane][8], B[lane][8] -> D layout: lane (n, hi) holds acc[q]: row m = 8*(q//4)+4*hi+q%4, col n
    Am=np.zeros((32,16),np.float32); Bk=np.zeros((16,32),np.float32)
    for lane in range(64):
        m=lane&31; g=lane>>5
        Am[m,8*g:8*g+8]=A[lane]; Bk[8*g:8*g+8,m]=Bm[lane]
    D=Am@Bk
    for lane in range(64):
        n=lane&31; hi=lane>>5
        for q in range(16):
            acc[lane,q]+=D[8*(q//4)+4*hi+q%4,n]
def run(x,w,bias):
    n_,_,H,W=x.shape
    Hc=(H-1)//2+1; Wc=(W-1)//2+1; Hp=(Hc-1)//2+1; Wp=(Wc-1)//2+1
    nbx=(Wp+WAVES*STRIP-1)//(WAVES*STRIP); nby=(Hp+R-1)//R
    wp=pack(w,bias)
    out=np.full((n_,Hp,Wp,64),np.nan,np.float32)
    LOW=-3e38
    for b in range(n_):
     for by in range(nby):
      for bx in range(nbx):
        PX0=bx*WAVES*STRIP; PY0=by*R; IXE=4*PX0-6; IY0=4*PY0-5
        lds=np.full(((ROWS+1)*ROWDW*2,),np.nan,np.float32)  # halves
        for row in range(ROWS):
            ly=row//3; c=row%3; y=IY0+ly
            for d in range(USED):
                xg=IXE+2*d
                ok = 0<=y<H and 0<=xg<W
                lds[(row*ROWDW+d)*2:(row*ROWDW+d)*2+2] = x[b,c,y,xg:xg+2] if ok else 0
        lds[ROWS*ROWDW*2:(ROWS*ROWDW+USED)*2]=1.0
        for wave in range(WAVES):
            PXs=PX0+wave*STRIP
            if PXs>=Wp: continue
            vm=[np.full((64,16),LOW,np.float32) for _ in range(2)]
            lanes=np.arange(64); nn=lanes&31; hh=lanes>>5
            col=(2*STRIP*wave+nn)
            cx=2*PXs-1+nn; col_ok=(cx>=0)&(cx<Wc)
            for j in range(CONVROWS):
                cy=2*PY0-1+j; valid=0<=cy<Hc
                acc=[np.zeros((64,16),np.float32) for _ in range(2)]
                if valid:
                    for s in range(STEPS):
                        Bv=np.zeros((64,8),np.float32)
                        for lane in range(64):
                            at=((6*j+hh[lane])*ROWDW+col[lane]+2*s*ROWDW)
                            if s==STEPS-1 and hh[lane]: at=ROWS*ROWDW+col[lane]
                            Bv[lane]=lds[at*2:at*2+8]
                        assert not np.isnan(Bv).any()
                        for t in range(2): mfma(wp[s,t],Bv,acc[t])
                    for t in range(2): vm[t]=np.maximum(vm[t],acc[t])
                if j>=2 and j%2==0:
                    py=PY0+j//2-1
                    if py<Hp:
                        for t in range(2):
                            u=vm[t].copy(); u[~col_ok]=LOW
                            nxt=lambda v: np.vstack([v[1:],np.zeros((1,16),np.float32)])
                            p=np.maximum(u,nxt(u)); r=np.maximum(np.maximum(p,nxt(p)),0)
                            for lane in range(64):
                                n=nn[lane]; px=PXs+(n>>1)
                                if n%2==0 and n<=2*(STRIP-1) and px<Wp:
                                    out[b,py,px,32*t+16*hh[lane]:32*t+16*hh[lane]+16]=r[lane]
                    for t in range(2): vm[t]=acc[t].copy() if valid else np.full((64,16),LOW,np.float32)
    return out
torch.manual_seed(0)  # small images: the model walks every lane in Python
for (n_,H,W) in [(1,20,26),(1,37,130),(2,8,8),(1,66,250)]:
    x=torch.randn(n_,3,H,W); w=torch.randn(64,3,7,7)/12; bias=torch.randn(64)
    want=F.max_pool2d(F.relu(F.conv2d(x,w,bias,2,3)),3,2,1).permute(0,2,3,1).numpy()
    got=run(x.numpy(),w.numpy(),bias.numpy())
    print((n_,H,W), got.shape, want.shape, np.isnan(got).sum(), np.abs(got-want).max())
