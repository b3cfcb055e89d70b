// Kernels of the TRAINING step (train.py:53-70: forward in train mode, loss.backward()): launch wrappers.
// Everything lives in HBM as fp32 (the reference trains in fp32); GEMM operands are split into bf16 hi / lo planes on the
// way into LDS and multiplied on v_mfma_f32_32x32x16_bf16 with fp32 accumulation: 3 products (hi.hi + hi.lo + lo.hi,
// ~2^-16 relative per product: "bf16x3") or 1 ("bf16").  bf16 rather than the inference path's fp16: gradients span the
// whole fp32 exponent range.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace eect {

// dropout site: keep(i) = hash(seed, site, i) >= p * 2^32; kept values are scaled by 1 / (1 - p).  p == 0: identity.
struct Drop {
  float p;
  uint64_t seed;
  uint32_t site;
};

// C[z](m, n) = alpha * sum_k A[z](m, k) * B[z](n, k)  (+ bias[n])  (+ C[z](m, n) if accumulate)
// A(m, k) = A[a_m * m + a_k * k], B(n, k) = B[b_n * n + b_k * k]: for each operand ONE of its two strides must be 1
// (k-contiguous operands are staged with 8-byte LDS stores, row-contiguous ones are transposed on the way in).
// Batch z in [0, nz): offsets (z / zdiv) * x_z0 + (z % zdiv) * x_z1 for x in {a, b, c}.  C is row-major with leading
// dimension c_m.  Any M, N, K >= 1 (edges are zero-filled / masked).
struct GemmArgs {
  const float* A; long a_m, a_k;
  const float* B; long b_n, b_k;
  float* C; long c_m;
  int M, N, K;
  int nz, zdiv;
  int ktot;  // > 0: split-K over the batch: batch z contracts k in [0, min(K, ktot - (z / zdiv) * K))
  long a_z0, a_z1, b_z0, b_z1, c_z0, c_z1;
  float alpha;
  const float* bias;
  int accumulate;
  // epilogue (single [M][N] problems only): 0 none; 1: C2 = drop(silu(C)); 2: C = C * dropmask * silu'(aux); 3: C = max(C, 0);
  // 4: C = aux + res_scale * drop(C)  (the residual connection around a sub-module); 5: C2 = drop(max(C, 0)); 6: C = C * dropmask * (aux > 0)
  int epi;
  float* C2;
  const float* aux;
  Drop drop;
  float res_scale;
  // row sums of A over the contraction (A row-contiguous only): rowsum[(z / zdiv) * rowsum_z + m] = sum_k A[z](m, k) -- the bias
  // gradient of a Linear falls out of its weight-gradient GEMM (A = dY^T) without another pass over dY
  float* rowsum;
  long rowsum_z;
  // set by launch_gemm: the output (and aux / old C read by the epilogue) is a large single-use tensor -- stores and those loads carry
  // the non-temporal hint, so that they do not push the operand tiles out of the L2 (EEC_TRAIN_NT_MB: threshold, 0 = never)
  int stream_out;
  int stream_a, stream_b;  // ... and which operand, if any, is loaded with the hint (EEC_TRAIN_NT_IN_MB, default 64)
};
inline GemmArgs gemm_args(const float* A, long a_m, long a_k, const float* B, long b_n, long b_k, float* C, long c_m, int M, int N, int K) {
  GemmArgs g{};
  g.A = A, g.a_m = a_m, g.a_k = a_k, g.B = B, g.b_n = b_n, g.b_k = b_k, g.C = C, g.c_m = c_m, g.M = M, g.N = N, g.K = K;
  g.nz = 1, g.zdiv = 1, g.alpha = 1.0f;
  return g;
}
hipError_t launch_gemm(const GemmArgs& g, int np, hipStream_t st);

// LayerNorm over the last dimension (eps 1e-5, affine): y = (x - mean) * rstd * g + b; mean / rstd [M] are kept
hipError_t launch_ln_fwd(const float* x, const float* g, const float* b, float* y, float* mean, float* rstd, int M, int D, hipStream_t st);
// dx = (dres ? dres : 0) + LN'(dy); part: [ln_bwd_blocks(M)][2][D] partial sums of (dy * xhat, dy) for dg / db
int ln_bwd_blocks(int M);
hipError_t launch_ln_bwd(const float* dy, const float* x, const float* g, const float* mean, const float* rstd, const float* dres,
                         float* dx, float* part, int M, int D, hipStream_t st);
// out[j] = sum_{s < S} part[s * stride + j], j < n
hipError_t launch_reduce_leading(const float* part, int S, long stride, long n, float* out, hipStream_t st);
// out0[j] = sum_s part[s][0][j], out1[j] = sum_s part[s][1][j], j < n  (part [S][2][n]: the LayerNorm / BatchNorm partials)
hipError_t launch_reduce_leading2(const float* part, int S, long n, float* out0, float* out1, hipStream_t st);
// general two-output form: columns [0, n0) of the n reduced columns go to out0, [n0, n) to out1
hipError_t launch_reduce_leading_split(const float* part, int S, long stride, long n, long n0, float* out0, float* out1, hipStream_t st);
// part[blk][n] = column sums of X[M][N] over the block's rows; blocks = colsum_blocks(M)
int colsum_blocks(int M);
hipError_t launch_colsum_partial(const float* X, int M, int N, float* part, hipStream_t st);

hipError_t launch_scale_drop(const float* dx, float scale, float* dh, long n, Drop d, hipStream_t st);   // dh = scale * dropmask * dx
hipError_t launch_glu_fwd(const float* u, float* g, int M, int D, hipStream_t st);                        // g = u[:, :D] * sigmoid(u[:, D:])
hipError_t launch_glu_bwd(const float* dg, const float* u, float* du, int M, int D, hipStream_t st);

// depthwise Conv1d over time (groups = D, zero padding (K - 1) / 2 at the sequence ends), x [B][T][D], w [D][K], b [D]
hipError_t launch_dw_fwd(const float* x, const float* w, const float* b, float* y, int B, int T, int D, int K, hipStream_t st);
hipError_t launch_dw_bwd_data(const float* dy, const float* w, float* dx, int B, int T, int D, int K, hipStream_t st);
int dw_bwd_weight_blocks(int B, int T);
hipError_t launch_dw_bwd_weight(const float* dy, const float* x, float* part /*[blocks + 1][K + 1][D]: taps, then bias*/, float* dw /*[D][K]*/, float* db /*[D]*/,
                                int B, int T, int D, int K, hipStream_t st);

// BatchNorm1d in train mode + SiLU: stats [2][D] = (batch mean, rstd = 1 / sqrt(biased var + 1e-5)); mv [2][D] = (mean, biased var)
hipError_t launch_bn_stats(const float* c, int M, int D, float* part /*[colsum_blocks(M) + 1][2][D]*/, float* stats, float* mv, hipStream_t st);
hipError_t launch_bn_silu_fwd(const float* c, const float* stats, const float* g, const float* b, float* s, int M, int D, hipStream_t st);
// sums [2][D] = (sum dy, sum dy * xhat) with dy = ds * silu'(bn(c)); then dc
hipError_t launch_bn_silu_bwd(const float* ds, const float* c, const float* stats, const float* g, const float* b, float* part, float* sums,
                              float* dc, int M, int D, hipStream_t st);

// P[z][tq][:] = softmax(scale * S[z][tq][:] + (tk >= len[z / H] ? -inf : 0)) in place
// Pd (optional): the dropped copy drop(P) written in the same pass
hipError_t launch_softmax_fwd(float* S, float* Pd, const int32_t* key_len, int B, int H, int T, float scale, Drop d, hipStream_t st);
// dS = scale * P * (dPd * dropmask - sum_k(dPd * dropmask * P)) in place over dP
hipError_t launch_softmax_bwd(const float* P, float* dP, int B, int H, int T, float scale, Drop d, hipStream_t st);

hipError_t launch_logsoftmax_fwd(const float* logits, float* logp, int M, int V, hipStream_t st);

// Fused self-attention of the training step (train_attention.hip; head dim 32 or 64): qkv [B*Tq][3D] (in_proj output),
// ctx / d_ctx [B*Tq][D], lse / delta [B*H][Tq] (row log-sum-exp kept for the backward; delta is scratch), dqkv [B*Tq][3D].
// No [B, H, Tq, Tq] tensor exists; the backward recomputes the probabilities.  Dropout stream positions are those of the
// flat [B*H][Tq][Tq] probability tensor.
bool attn_fused_supported(int D, int H);
hipError_t launch_attn_fwd_fused(const float* qkv, const int32_t* key_len, float* ctx, float* lse, int B, int H, int Tq, int D, int np, Drop d,
                                 hipStream_t st);
hipError_t launch_attn_bwd_fused(const float* qkv, const int32_t* key_len, const float* ctx, const float* d_ctx, const float* lse, float* delta,
                                 float* dqkv, int B, int H, int Tq, int D, int np, Drop d, hipStream_t st);

// AED decoder helpers (decoder.hip).  P[z][tq][:] = softmax(scale * S[z][tq][:] + mask) in place over [B*H][Tq][Tk] scores:
// key tk is masked for query tq of batch b = z / H when (causal && tk > tq) || (key_pad && key_pad[b * Tk + tk])
hipError_t launch_softmax_masked(float* S, int B, int H, int Tq, int Tk, float scale, int causal, const unsigned char* key_pad, hipStream_t st);
// the same with the dropped copy Pd = drop(P) written in the same pass (Pd may be null), and the backward of any such softmax over
// `rows` rows of Tk probabilities: dS = scale * P * (dPd * dropmask - sum_k(dPd * dropmask * P)) in place over dP
hipError_t launch_softmax_masked_drop(float* S, float* Pd, int B, int H, int Tq, int Tk, float scale, int causal, const unsigned char* key_pad,
                                      Drop d, hipStream_t st);
hipError_t launch_softmax_bwd_rows(const float* P, float* dP, long rows, int Tk, float scale, Drop d, hipStream_t st);
// train mode: x = drop(emb[tok] + pe) (positional_encoder_2's dropout); demb[v] = sum of the (masked) dx rows of the tokens equal to v
hipError_t launch_embed_pe_drop(const long long* tok, const float* emb, const float* pe, float* x, unsigned char* pad, long n_tok, int S, int D, int V,
                                int pad_idx, Drop d, hipStream_t st);
hipError_t launch_embed_bwd(const long long* tok, const float* dx, float* demb, long n_tok, int D, int V, Drop d, hipStream_t st);
// x[i][:] = emb[tok[i]][:] + pe[i % S][:];  pad[i] = (tok[i] == pad_idx)
hipError_t launch_embed_pe(const long long* tok, const float* emb, const float* pe, float* x, unsigned char* pad, long n_tok, int S, int D, int V,
                           int pad_idx, hipStream_t st);

// stem helpers: im2col of the first conv (mel [B][C][T] -> [B*T1][C*3], column c*3 + j = mel[b][c][2*t1 + j]);
// [O][C][3] <-> [O][3][C] weight permutes; gather of the second conv's input gradient from G [B*T2][3][D]
hipError_t launch_im2col_mel(const float* mel, float* a, int B, int C, int T, int T1, hipStream_t st);
hipError_t launch_permute_w3(const float* w, float* wp, int O, int C, int to_jc, hipStream_t st);
hipError_t launch_col2im_stride2(const float* G, float* dout1, int B, int T1, int T2, int D, hipStream_t st);
// x[b][t][:] = drop(x[b][t][:] + pe[t][:])
hipError_t launch_add_pe_drop(float* x, const float* pe, int B, int T, int D, Drop d, hipStream_t st);
hipError_t launch_axpy(float* y, const float* x, float a, long n, hipStream_t st);  // y += a * x

}  // namespace eect
