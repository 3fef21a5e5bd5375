#!/usr/bin/env python3
"""A/B of compile-time variants of gemm_bf16.hip in ONE process (interleaved rounds; cdna_hip_programming.md rule 24).

    python scripts/gemm_ab.py --build spread=-DSER_GEMM_SPREAD_DMA=1 [name=-Dflag ...]     (cross-compiles, no GPU)
    python scripts/gemm_ab.py spread [...]                                                (on the MI355X)

Interleaved three-product mode, layer / conv shapes of BASELINE config 2, every classic tile height; prints the
median of the rounds per (shape, height, variant)."""
import ctypes as C
import glob
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "multilingual-multimodal-speech-emotion-recognition_amd", "csrc")
DIAG = os.path.join(ROOT, "scripts", "diag")
SHAPES = [("qkv", 3696, 2304, 768), ("ffn1", 3696, 3072, 768), ("ffn2", 3696, 768, 3072), ("oproj", 3696, 768, 768),
          ("conv1", 102384, 512, 1536), ("conv3", 25584, 512, 1536)]


def build(specs):
    os.makedirs(DIAG, exist_ok=True)
    subprocess.check_call(["make", "-C", CSRC])
    others = [o for o in glob.glob(os.path.join(CSRC, "*.o")) if not o.endswith("gemm_bf16.o")]
    for spec in specs:
        name, flags = spec.split("=", 1)
        obj = os.path.join(DIAG, f"gemm_{name}.o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17"] + flags.split(",") +
                              ["-c", os.path.join(CSRC, "gemm_bf16.hip"), "-o", obj])
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                               os.path.join(DIAG, f"libser_{name}.so"), obj] + others)


def load(name):
    path = os.path.join(ROOT, "multilingual-multimodal-speech-emotion-recognition_amd", "libser_hip.so") if name == "product" \
        else os.path.join(DIAG, f"libser_{name}.so")
    lib = C.CDLL(path)
    lib.ser_debug_gemm_il_cfg.restype = C.c_int
    lib.ser_debug_gemm_il_cfg.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    return lib


def main():
    if sys.argv[1] == "--build":
        return build(sys.argv[2:])
    import torch
    names = ["product"] + sys.argv[1:]
    libs = {n: load(n) for n in names}
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(0)
    for sname, M, N, K in SHAPES:
        a = (torch.randn(M, 2 * K, generator=g) * 0.5).to("cuda", torch.bfloat16)
        w = (torch.randn(N, 2 * K, generator=g) * 0.05).to("cuda", torch.bfloat16)
        out = torch.empty(M, N, dtype=torch.float32, device="cuda")
        for bm in (64, 96, 128, 160, 192):
            res = {n: [] for n in names}
            for rnd in range(5):
                for n in names:
                    lib = libs[n]
                    run = lambda: lib.ser_debug_gemm_il_cfg(a.data_ptr(), w.data_ptr(), M, N, K, bm, 1, out.data_ptr(), st)
                    assert run() == 0
                    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
                    e0.record()
                    for _ in range(6):
                        run()
                    e1.record()
                    torch.cuda.synchronize()
                    res[n].append(e0.elapsed_time(e1) / 6 * 1e3)
            ideal = 2.0 * M * N * K * 3 / 2.5e15 * 1e6
            print(f"{sname:6s} BM={bm:3d}  " + "  ".join(f"{n} {statistics.median(res[n]):7.1f} us ({ideal / statistics.median(res[n]):5.1%})" for n in names), flush=True)


if __name__ == "__main__":
    main()
