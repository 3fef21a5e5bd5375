#!/usr/bin/env python3
"""Where does the time of one encoder GEMM go?  Times the product library and four probe builds of gemm_bf16.hip
(-DSER_GEMM_DIAG bits: 1 = no epilogue, 2 = no fragment reads / MFMAs, 4 = no global->LDS staging, 8 = no global stores
in the epilogue; 6 = "skeleton": launch, prologue, barriers and epilogue only; 7 = "floor": launch, prologue, barriers) on the encoder-layer shapes, standalone on an idle chip.

Build the probe libraries first (cross-compiles without a GPU):
    python scripts/gemm_phase_probe.py --build
then on the MI355X:
    python scripts/gemm_phase_probe.py
"""
import ctypes as C
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "multilingual-multimodal-speech-emotion-recognition_amd", "csrc")
DIAG = os.path.join(ROOT, "scripts", "diag")
VARIANTS = {0: "full", 1: "no epilogue", 2: "no reads/MFMA", 4: "no staging", 6: "skeleton", 7: "floor", 14: "skeleton, no stores",
            "direct": "full, register-direct epilogue"}


def build():
    os.makedirs(DIAG, exist_ok=True)
    subprocess.check_call(["make", "-C", CSRC])
    others = [o for o in glob.glob(os.path.join(CSRC, "*.o")) if not o.endswith("gemm_bf16.o")]
    for d in VARIANTS:
        if d == 0:
            continue
        obj = os.path.join(DIAG, f"gemm_diag{d}.o")
        define = "-DSER_GEMM_EPI_DIRECT=1" if d == "direct" else f"-DSER_GEMM_DIAG={d}"
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17",
                               define, "-c", os.path.join(CSRC, "gemm_bf16.hip"), "-o", obj])
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                               os.path.join(DIAG, f"libser_diag{d}.so"), obj] + others)


def load(d):
    path = os.path.join(ROOT, "multilingual-multimodal-speech-emotion-recognition_amd", "libser_hip.so") if d == 0 \
        else os.path.join(DIAG, f"libser_diag{d}.so")
    lib = C.CDLL(path)
    vp, i32 = C.c_void_p, C.c_int
    lib.ser_gemm_bf16_nt.restype = i32
    lib.ser_gemm_bf16_nt.argtypes = [vp, vp, i32, vp, vp, i32, i32, i32, i32, vp, i32, vp, i32, vp, vp, vp, i32, vp]
    lib.ser_debug_set_gemm_bm.argtypes = [i32]
    return lib


def main():
    if "--build" in sys.argv:
        return build()
    import torch
    libs = {d: load(d) for d in VARIANTS}
    st = torch.cuda.current_stream().cuda_stream
    shapes = [(3184, 2304, 768, "QKV", 192, 0), (3184, 2304, 768, "QKV", 128, 0), (3184, 768, 768, "out", 64, 0),
              (3184, 3072, 768, "FFN1", 160, 1), (3184, 3072, 768, "FFN1", 160, 0), (3184, 768, 3072, "FFN2", 64, 0)]
    print("standalone, bf16 planes in / bf16 plane out, bias (+ GELU where marked) epilogue; us per launch (20 launches averaged)")
    for M, N, K, name, bm, act in shapes:
        a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
        bias = torch.randn(N, device="cuda")
        ch = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        row = {}
        for d, lib in libs.items():
            lib.ser_debug_set_gemm_bm(bm)

            def run():
                rc = lib.ser_gemm_bf16_nt(a.data_ptr(), None, K, w.data_ptr(), None, K, M, N, K, bias.data_ptr(), act, None, 0,
                                          None, ch.data_ptr(), None, N, st)
                assert rc == 0
            for _ in range(3):
                run()
            e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
            e0.record()
            for _ in range(20):
                run()
            e1.record()
            torch.cuda.synchronize()
            row[d] = e0.elapsed_time(e1) / 20 * 1e3
        tiles = -(-M // bm) * -(-N // 128)
        ideal = 2.0 * M * N * K / 2.5e15 * 1e6
        print(f"{name:5s} {M}x{N}x{K} {'gelu' if act else 'bias'} bm={bm:3d} tiles={tiles:4d} MFMA-ideal {ideal:5.1f}us | " +
              " | ".join(f"{VARIANTS[d]} {row[d]:6.1f}" for d in VARIANTS) + f" | {2.0 * M * N * K / row[0] / 1e6:5.0f} TF", flush=True)


if __name__ == "__main__":
    main()
