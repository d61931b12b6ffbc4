"""What a plain device memset reaches on this GPU, for the size of one egocentric observation batch (65 536 x 133 x 117 B):
the store-bound floor of ego_sparse_kernel.  torch's zero_() (a tuned fill kernel) and hipMemsetAsync."""
import torch
n = 65536 * 133 * 117
x = torch.empty(n, dtype=torch.uint8, device='cuda')
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for name, fn in (("torch zero_()", lambda: x.zero_()), ("torch fill_(7)", lambda: x.fill_(7))):
    for _ in range(3):
        fn()
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("%-16s %.4f ms  %.1f GB/s" % (name, ms, n / ms / 1e6), flush=True)
