"""f32 rounding error of Winograd tile shapes against an f64 convolution (CPU, numpy): F(2x2), F(4,3)xF(2,3), F(2,3)xF(4,3), F(4x4)
with f32 transforms and f32 accumulation, next to the direct f32 convolution.  Numbers quoted in DESIGN.md section 4."""
import numpy as np, torch
torch.manual_seed(0)
def mats(m):
    if m==2:
        BT=np.array([[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]],np.float64)
        G=np.array([[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]],np.float64)
        AT=np.array([[1,1,1,0],[0,1,-1,-1]],np.float64)
    else:
        BT=np.array([[4,0,-5,0,1,0],[0,-4,-4,1,1,0],[0,4,-4,-1,1,0],[0,-2,-1,2,1,0],[0,2,-1,-2,1,0],[0,4,0,-5,0,1]],np.float64)
        G=np.array([[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]],np.float64)
        AT=np.array([[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,0],[0,1,-1,8,-8,1]],np.float64)
    return BT,G,AT
def wino(x,w,mh,mw,dt):
    # x: C,H,W ; w: K,C,3,3 ; pad 1; H%mh==0, W%mw==0
    BTh,Gh,ATh=[a.astype(dt) for a in mats(mh)]; BTw,Gw,ATw=[a.astype(dt) for a in mats(mw)]
    C,H,W=x.shape; K=w.shape[0]
    xp=np.zeros((C,H+2,W+2),dt); xp[:,1:-1,1:-1]=x
    U=np.einsum('ai,kcij,bj->abkc',Gh,w.astype(dt),Gw).astype(dt)
    y=np.zeros((K,H,W),dt)
    for th in range(H//mh):
        for tw in range(W//mw):
            d=xp[:,th*mh:th*mh+mh+2, tw*mw:tw*mw+mw+2]
            V=np.einsum('ai,cij,bj->abc',BTh,d,BTw).astype(dt)
            M=np.einsum('abkc,abc->abk',U,V).astype(dt)  # accumulate in dt
            y[:,th*mh:(th+1)*mh, tw*mw:(tw+1)*mw]=np.einsum('ia,abk,jb->kij',ATh,M,ATw).astype(dt)
    return y
for C,H,W in ((512,4,12),(256,8,24)):
    x=np.maximum(np.random.randn(C,H,W),0).astype(np.float32)   # post-ReLU like
    w=(np.random.randn(C,C,3,3)*np.sqrt(2/(9*C))).astype(np.float32)
    ref=torch.nn.functional.conv2d(torch.from_numpy(x).double()[None],torch.from_numpy(w).double(),padding=1)[0].numpy()
    d32=torch.nn.functional.conv2d(torch.from_numpy(x)[None],torch.from_numpy(w),padding=1)[0].numpy()
    s=np.abs(ref).max()
    print(C,H,W,'direct f32 err', np.abs(d32-ref).max()/s)
    for mh,mw in ((2,2),(4,2),(2,4),(4,4)):
        y=wino(x,w,mh,mw,np.float32)
        print('  F(%d,%d) max rel-to-max err %.3e  rms %.3e'%(mh,mw,np.abs(y-ref).max()/s, np.sqrt(((y-ref)**2).mean())/s))
