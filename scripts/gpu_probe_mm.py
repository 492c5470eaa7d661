import torch, time
for n in (8192, 16384):
    a=torch.randn(n,n,device="cuda",dtype=torch.bfloat16); b=torch.randn(n,n,device="cuda",dtype=torch.bfloat16)
    for _ in range(3): c=a@b
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(10): c=a@b
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/10
    print(f"torch bf16 matmul {n}^3: {dt*1e3:.2f} ms  {2*n**3/dt/1e12:.0f} TFLOP/s", flush=True)
