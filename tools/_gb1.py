import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from dotaclient_amd import ops
dev = torch.device('cuda:0')
def run(M, N, K, a_km, b_km, aux=False):
    A = torch.randn((K, M) if a_km else (M, K), device=dev)
    B = torch.randn((K, N) if b_km else (N, K), device=dev)
    C = torch.empty(M, N, device=dev)
    ax = torch.randn(M, N, device=dev) if aux else None
    for _ in range(3):
        ops.gemm(A, B, C, M, N, K, M if a_km else K, N if b_km else K, N, a_km, b_km, aux=ax, ldaux=N)
    torch.cuda.synchronize()
run(262144, 128, 128, False, False)
run(16384, 256, 896, False, False)
run(262144, 128, 128, False, True, aux=True)
