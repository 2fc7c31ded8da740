import torch, sys
sys.path.insert(0,'.')
from gpplus_amd.backend import get_context, square_buffer
ctx=get_context('cuda:0')
for D in (8, 49, 64):
    N=700
    U=torch.randn(N,D,dtype=torch.float64,device='cuda'); w=torch.rand(D,dtype=torch.float64,device='cuda')*0.05
    sf2=torch.tensor([0.9],dtype=torch.float64,device='cuda')
    K=square_buffer(N,'cuda'); ctx.kernel_build(U,w,sf2,None,None,K,uplo=0)
    d=((U[:,None,:]-U[None,:,:])**2*w).sum(-1); ref=0.9*torch.exp(-d)
    print(D, float((K-ref).abs().max()))
