#!/usr/bin/env python3
"""Microbenchmark of the training GEMM (csrc/train_kernels.hip) through eec_train_gemm at the shapes of the default model:
C[M][N] = A . B^T, fp32 in HBM, bf16 hi/lo split on the fly.  Prints us and algorithmic TFLOP/s per shape / layout / passes."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from early_exit_transformer_amd import capi  # noqa: E402

lib = capi.load()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
shapes = [(16384, 2048, 256, "ffn1 fwd / dact"), (16384, 256, 2048, "ffn2 fwd / dln"), (16384, 768, 256, "in_proj"), (16384, 256, 256, "out_proj"),
          (2048, 256, 16384, "dW ffn (one split would be K=1024)")]
for M, N, K, what in shapes:
    A = torch.randn(M, K, device="cuda")
    B = torch.randn(N, K, device="cuda")
    At, Bt = A.t().contiguous(), B.t().contiguous()
    out = torch.empty(M, N, device="cuda")
    for at, bt in ((0, 0), (0, 1), (1, 1)):
        for passes in (3, 1):
            a, b = (At if at else A), (Bt if bt else B)
            def run():
                lib.eec_train_gemm(a.data_ptr(), b.data_ptr(), None, out.data_ptr(), M, N, K, passes, at, bt, st)
            for _ in range(3):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 100
            print(f"{what:36s} M{M} N{N} K{K} A{'t' if at else 'n'} B{'t' if bt else 'n'} passes {passes}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s", flush=True)
