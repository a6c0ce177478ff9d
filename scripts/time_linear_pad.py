"""Does the 64 KB row stride of the 16384-wide weight hurt?  Forward GEMM with the weight's rows padded by `pad` floats."""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from disentangle_mlp_amd import ops, _lib
def timeit(fn, n=20):
    for _ in range(60): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return statistics.median(ts)
M, N, K = 128, 2048, 16384
lib = _lib.load()
x, b = torch.randn(M, K, device="cuda"), torch.randn(N, device="cuda")
for pad in (0, 32, 64, 256, 1024):
    wfull = torch.randn(N, K + pad, device="cuda") / K ** 0.5
    w = wfull[:, :K]
    y = torch.empty(M, N, device="cuda")
    need = lib.vg_gemm_nt_bf16split_workspace_bytes(M, N, K)
    ws = ops.workspace(need, x.device)
    def split():
        _lib.check(lib.vg_gemm_nt_bf16split(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), M, N, K, K, 1, K + pad, 1, 3,
                                            ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream), "gemm")
    t1 = timeit(split)
    ref = torch.nn.functional.linear(x, w, b)
    err = float((y - ref).norm() / ref.norm())
    t2 = timeit(lambda: torch.nn.functional.linear(x, w, b))
    print(f"pad {pad:5d}: split {t1*1e3:7.1f} us   vendor {t2*1e3:7.1f} us   (split vs vendor rel diff {err:.1e})", flush=True)
