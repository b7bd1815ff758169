"""Yardstick only (NOT used by the product): fp32 torch.matmul (rocBLAS/hipBLASLt) on the shapes of the path,
to put the hand-written GEMM's TFLOP/s in context.  Run on the GPU box: python tools/torch_matmul_yardstick.py"""
import torch, time
torch.backends.cuda.matmul.allow_tf32=False
for (M,N,K) in [(4096,4096,4096),(8192,8192,2048),(4410,5808,1936),(4410,1936,1936),(2240,5808,1936),(2240,512,12544),(2640,5808,1936),(330,5808,1936)]:
    a=torch.randn(M,K,device='cuda'); b=torch.randn(N,K,device='cuda')
    for _ in range(3): c=a@b.T
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): c=a@b.T
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)*100
    print(M,N,K, round(us,1),'us', round(2*M*N*K/us/1e6,1),'TF (torch.matmul fp32 = rocBLAS/hipBLASLt)')
