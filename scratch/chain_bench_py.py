import ctypes, os, sys, torch
x = torch.zeros(1024, device="cuda"); torch.cuda.synchronize()
print({k: v for k, v in os.environ.items() if k.startswith(("HIP", "HSA", "AMD", "ROC", "GPU"))})
L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libchain_bench.so"))
L.bench_main(0, None)
