"""Rounding error of Winograd F(4x4, 3x3) in fp32 (numpy, CPU) against an fp64 direct convolution on layer-sized problems: forward and
weight gradient, with the transform matrices csrc/wino4.hip uses.  usage: python tools/wino43_error.py"""
import numpy as np, torch
torch.manual_seed(0)
# F(4x4,3x3) matrices (Lavin & Gray)
Bt = np.array([[4,0,-5,0,1,0],[0,-4,-4,1,1,0],[0,4,-4,-1,1,0],[0,-2,-1,2,1,0],[0,2,-1,-2,1,0],[0,4,0,-5,0,1]],dtype=np.float64)
G = np.array([[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]],dtype=np.float64)
At = np.array([[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,0],[0,1,-1,8,-8,1]],dtype=np.float64)
def wino(x, w, dt):
    # x [C,H,W] (H,W multiple of 4, padded by 1 outside), w [K,C,3,3]
    C,H,W = x.shape; K = w.shape[0]
    xp = np.zeros((C,H+2,W+2),dtype=dt); xp[:,1:-1,1:-1] = x
    U = np.einsum('ij,kcjl,ml->kcim', G.astype(dt), w.astype(dt), G.astype(dt)).astype(dt)   # [K,C,6,6]
    out = np.zeros((K,H,W),dtype=dt)
    for ty in range(H//4):
        for tx in range(W//4):
            d = xp[:,4*ty:4*ty+6,4*tx:4*tx+6]
            V = np.einsum('ij,cjl,ml->cim', Bt.astype(dt), d, Bt.astype(dt)).astype(dt)
            M = np.einsum('kcim,cim->kim', U, V).astype(dt)
            Y = np.einsum('ij,kjl,ml->kim', At.astype(dt), M, At.astype(dt)).astype(dt)
            out[:,4*ty:4*ty+4,4*tx:4*tx+4] = Y
    return out
for C,K,H in ((128,128,8),(256,64,8),(512,32,8)):
    x = torch.randn(C,H,H).numpy().astype(np.float64); x = np.maximum(x,0)      # post-ReLU activations
    w = (torch.randn(K,C,3,3)/ (9*C)**0.5).numpy().astype(np.float64)
    ref = torch.nn.functional.conv2d(torch.from_numpy(x)[None], torch.from_numpy(w), padding=1)[0].numpy()
    y32 = wino(x.astype(np.float32), w.astype(np.float32), np.float32)
    d32 = torch.nn.functional.conv2d(torch.from_numpy(x.astype(np.float32))[None], torch.from_numpy(w.astype(np.float32)), padding=1)[0].numpy()
    s = np.abs(ref).max()
    print(C, "F(4,3) fp32 max err / max|y|:", np.abs(y32-ref).max()/s, " direct fp32:", np.abs(d32-ref).max()/s)

def wino_wgrad(x, dy, dt):
    C,H,W = x.shape; K = dy.shape[0]
    xp = np.zeros((C,H+2,W+2),dtype=dt); xp[:,1:-1,1:-1] = x
    dU = np.zeros((K,C,6,6),dtype=dt)
    A = At.T.astype(dt)
    for ty in range(H//4):
        for tx in range(W//4):
            d = xp[:,4*ty:4*ty+6,4*tx:4*tx+6]
            V = np.einsum('ij,cjl,ml->cim', Bt.astype(dt), d, Bt.astype(dt)).astype(dt)
            Yp = np.einsum('ij,kjl,ml->kim', A, dy[:,4*ty:4*ty+4,4*tx:4*tx+4], A).astype(dt)
            dU += np.einsum('kim,cim->kcim', Yp, V).astype(dt)
    return np.einsum('ji,kcjl,lm->kcim', G.astype(dt), dU, G.astype(dt)).astype(dt)
print("weight gradient:")
for C,K,H,N in ((128,16,28,2),(256,16,16,4)):
    tot32 = 0; tot64 = 0
    for n in range(N):
        x = np.maximum(torch.randn(C,H,H).numpy().astype(np.float64),0)
        dy = torch.randn(K,H,H).numpy().astype(np.float64)
        tot32 = tot32 + wino_wgrad(x.astype(np.float32), dy.astype(np.float32), np.float32).astype(np.float64)
        xr = torch.from_numpy(x)[None]; 
        wr = torch.zeros(K,C,3,3,dtype=torch.float64,requires_grad=True)
        torch.nn.functional.conv2d(xr, wr, padding=1).backward(torch.from_numpy(dy)[None])
        tot64 = tot64 + wr.grad.numpy()
    print(C, H, "F(4,3) wgrad fp32 max err / max|dW|:", np.abs(tot32-tot64).max()/np.abs(tot64).max())
