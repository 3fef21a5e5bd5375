import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ser_amd
from ser_amd import _lib as L
f = L.lib.ser_debug_gemm_pair
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p] * 2 + [C.c_void_p]
torch.manual_seed(0)
def mk(M, N, K):
    a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    return a, w, torch.full((M, N), float("nan"), device="cuda")
for (M0, N0, K0), (M1, N1, K1) in [((24, 384, 128), (140, 384, 128)), ((24, 128, 128), (140, 128, 128)), ((24, 128, 256), (140, 128, 256)),
                                   ((100, 384, 128), (140, 384, 128)), ((24, 384, 128), (24, 384, 128)), ((512, 768, 768), (3184, 768, 768)),
                                   ((64, 384, 128), (140, 384, 128)), ((65, 200, 128), (140, 200, 128))]:
    a0, w0, c0 = mk(M0, N0, K0); a1, w1, c1 = mk(M1, N1, K1)
    rc = f(a0.data_ptr(), w0.data_ptr(), M0, N0, K0, c0.data_ptr(), a1.data_ptr(), w1.data_ptr(), M1, N1, K1, c1.data_ptr(),
           torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    r0 = a0.float() @ w0.float().T; r1 = a1.float() @ w1.float().T
    e0 = (c0 - r0).abs(); e1 = (c1 - r1).abs()
    bad0 = (~(e0 < 1e-3)).nonzero()
    print((M0, N0, K0), (M1, N1, K1), rc, "err0", e0.max().item(), "err1", e1.max().item(), "bad0 count", len(bad0),
          "rows", sorted(set(bad0[:, 0].tolist()))[:8], "cols", sorted(set(bad0[:, 1].tolist()))[:8])
