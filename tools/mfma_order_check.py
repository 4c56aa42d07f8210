import sys, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import oracle
import experimental_gpu_programming_for_a_spectral_numerical_integration_amd as sri
from conftest import rod_strain_panel, uniform_panel
for n,rods in ((64,64),(256,64),(1024,32)):
    op=sri.SpectralApply(n, sri.OP_DN, device=0)
    A=oracle.dn(n)
    for kind in ('rod','uniform'):
        X = rod_strain_panel(n, rods) if kind=='rod' else uniform_panel(n,4*rods)
        Xt=torch.from_numpy(np.ascontiguousarray(X.T)).cuda()
        Y=op.apply(Xt).cpu().numpy().T
        Yx=oracle.apply_exact(A,X); Yo=oracle.apply(A,X)
        f=lambda a,b: np.linalg.norm(a-b)/np.linalg.norm(b)
        match={m: bool(np.array_equal(Y, oracle.model_apply(A,X,m))) for m in range(6)}
        print(n,kind,'gpu vs oracle %.3e  gpu vs exact %.3e  oracle vs exact %.3e'%(f(Y,Yo),f(Y,Yx),f(Yo,Yx)),'bit-exact models:',[m for m,v in match.items() if v], op.last_kernel)
