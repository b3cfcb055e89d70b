#!/usr/bin/env python3
"""Diagnostic: s_memtime stamps inside the training GEMM (needs a library built with -DEECT_TL:
tools/build_variant.sh tl "-DEECT_TL" train_kernels.hip; EEC_LIB=.../libeec_tl.so).  Prints, for 64 workgroups spread over the
launch, cycles from kernel entry to: k-loop start, per k-tile (after the store + barrier, after the MFMAs) for the first four
k-tiles, k-loop end, epilogue end."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

lib = C.CDLL(os.environ["EEC_LIB"])
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
lib.eec_train_gemm.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
lib.eect_debug_gemm_epi.argtypes = [C.c_void_p] * 5 + [C.c_int] * 5 + [C.c_float, C.c_void_p]
EPI = {"plain": 0, "silu (second output)": 1, "silu' x mask (reads aux)": 2, "residual + dropout (reads aux)": 4}
cases = [(2048, 2048, 256, "ffn1 at M 2048 (C = 16 MB stays in the caches)", 3, "plain", 0), (16384, 2048, 256, "ffn1", 3, "plain", 0), (16384, 2048, 256, "ffn1", 1, "plain", 0), (16384, 256, 2048, "ffn2", 3, "plain", 0),
         (16384, 256, 2048, "ffn2", 1, "plain", 0), (16384, 768, 256, "in_proj", 3, "plain", 0), (16384, 768, 256, "in_proj", 1, "plain", 0),
         (16384, 2048, 256, "ffn1 fwd", 3, "silu (second output)", 0), (16384, 2048, 256, "ffn dact, B transposed", 3, "plain", 1),
         (16384, 2048, 256, "ffn dact, B transposed", 3, "silu' x mask (reads aux)", 1), (16384, 256, 2048, "ffn2 fwd", 3, "residual + dropout (reads aux)", 0)]
for M, N, K, what, passes, epi, bt in cases:
    A = torch.randn(M, K, device="cuda")
    B = torch.randn(K, N, device="cuda") if bt else torch.randn(N, K, device="cuda")
    out = torch.empty(M, N, device="cuda")
    aux, out2 = torch.randn(M, N, device="cuda"), torch.empty(M, N, device="cuda")
    what = f"{what}, epilogue {epi}"
    if True:
        def launch():
            if epi == "plain" and not bt:
                lib.eec_train_gemm(A.data_ptr(), B.data_ptr(), None, out.data_ptr(), M, N, K, passes, 0, 0, st)
            else:
                lib.eect_debug_gemm_epi(A.data_ptr(), B.data_ptr(), out.data_ptr(), aux.data_ptr(), out2.data_ptr(), M, N, K, EPI[epi], bt, 0.1, st)
        for _ in range(3):
            launch()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        launch()
        e1.record()
        torch.cuda.synchronize()
        buf = (C.c_ulonglong * (64 * 16))()
        lib.eect_debug_tl(buf)
        t = np.array(buf, dtype=np.int64).reshape(64, 16)
        rel = t[:, 1:12] - t[:, :1]
        rel = np.concatenate([rel[:, :1], rel[:, 3:11], rel[:, 1:3]], axis=1)  # start, 8 k-tile stamps, loop end, epilogue end
        start = t[:, 0] - t[:, 0].min()
        print(f"{what} M{M} N{N} K{K} passes {passes}: {e0.elapsed_time(e1) * 1e3:.1f} us; s_memtime ticks (100 MHz on gfx950? compare the total with the us figure)")
        print("   workgroup start offsets (first 16 sampled):", start[:16].tolist())
        print("   median over sampled workgroups: loop start %d | k-tiles (stored, mfma done) %s | loop end %d | epilogue end %d" % (
            np.median(rel[:, 0]), [int(np.median(rel[:, i])) for i in range(1, 9)], np.median(rel[:, 9]), np.median(rel[:, 10])))
        print("   last sampled workgroup ends at", int((t[:, 3] - t[:, 0].min()).max()))
