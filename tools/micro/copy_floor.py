import torch, time
dev = torch.device("cuda:0")
n = 2_280_000_000 // 4
a = torch.empty(n, dtype=torch.int32, device=dev); b = torch.empty(n, dtype=torch.int32, device=dev)
a.fill_(7)
for _ in range(3): b.copy_(a)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): b.copy_(a)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print("device copy of %.2f GB: %.3f ms = %.2f TB/s read+write" % (n * 4 / 1e9, ms, 2 * n * 4 / ms / 1e9))
