"""Run torch.matmul on the BERT-large forward shapes once so that a kernel trace shows which hipBLASLt kernels it picks."""
import torch
T, H = 131072, 1024
for (M, N, K, layout) in [(T, 3 * H, H, "nt"), (T, H, H, "nt"), (T, 4 * H, H, "nt"), (T, H, 4 * H, "nt"), (T, H, 3 * H, "nn"), (T, 4 * H, H, "nn")]:
    A = torch.randn(M, K, device="cuda", dtype=torch.float16)
    B = torch.randn(N, K, device="cuda", dtype=torch.float16) if layout == "nt" else torch.randn(K, N, device="cuda", dtype=torch.float16)
    for _ in range(3):
        C = A @ (B.t() if layout == "nt" else B)
    torch.cuda.synchronize()
