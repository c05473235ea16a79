// layer_fused.hip -- one C entry per GraphSAGE / GCN layer pass.
//
// The dense part of a GraphSAGE layer (shaDow/layers.py:471-483),
//     out = norm_0(act(X Ws^T + bs)) + norm_1(act((A X) Wn^T + bn)),
// was six kernels forward (SpMM, two weight packs, two split-bf16 GEMMs, fused bias/act/norm) and about ten backward.
// Launched one by one from Python through ctypes they cost more host time than GPU time at the reference's own batch
// sizes (16-256 roots per step).  These entries enqueue the whole pass with ONE call; they own no kernel, only the choice and
// the order of the existing ones.  Round 3: forward = weight pack, SpMM, ONE GEMM-epilogue kernel (gemm_fused.hip); backward =
// [act_norm backward,] transposed SpMM, the input-gradient GEMM with the act_norm backward of the layer below in its
// epilogue, ONE launch for both weight gradients (sl_gemm_tn_f16_pair, the neighbour branch as (A^T dZn)^T X).
#include <string.h>

#include <algorithm>

#include "common.h"

using namespace shadow;

namespace {

// Rows wider than 128 floats on the pipelined CSR kernel (a row per wavefront, the gathers out of the L2) instead of the
// block-diagonal LDS kernel, for batches whose rows may be long (round 5, after the CSR kernel's long-row loop took 6 gathers in
// flight).  One launch at 256 floats, plain row-normalised adjacency (scripts/micro/ab_spmm_long.sh): PPR batch (153 k rows, root
// rows of up to 199 entries: the LDS kernel walks them with one thread per float4 column) 122 -> 94 us, k-hop batch (290 k rows)
// 155 -> 141; inside the training step, with the drop-edge mask and the transposed passes' permutation (scripts/ab_spmm_wide.sh):
// PPR 139 -> 112 us (step 5.98 -> 5.76 ms), but k-hop 151 -> 163 (6.11 -> 6.18) and arxiv 41 -> 52 (2.26 -> 2.41): by the bound.
static int g_spmm_wide_pipe = 1;                     // 0: never, 1: batches with long rows, 2: always
// (... and enough rows to fill the persistent kernel's 4 096 wavefronts with chunks of six 4-row groups: the papers100M PPR batches,
//  ~40 k rows, run 0.051 ms on the LDS kernel and 0.072 on the pipelined one)
constexpr uint32_t kWidePipeRowEntries = 64, kWidePipeMinRows = 98304;
static bool spmm_wide_pipe(const sl_norm_adj *a) {
  return g_spmm_wide_pipe == 2 || (g_spmm_wide_pipe == 1 && a->row_entries_bound > kWidePipeRowEntries && a->n >= kWidePipeMinRows);
}

// does the SpMM of this shape run on a kernel that can join row maxima with what the array holds?
bool spmm_joins(const sl_norm_adj *a, uint32_t F, const float *X, int64_t ldx, const float *Y, int64_t ldy) {
  if (spmm_wide_pipe(a) && spmm_csr_whole_rows(F, X, ldx, Y, ldy)) return true;     // (a row per wavefront: plain read-modify-write)
  return a->subg_node_off && F >= 96 && (F % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) && !(reinterpret_cast<uintptr_t>(X) & 15) &&
         !(reinterpret_cast<uintptr_t>(Y) & 15);
}

// Row maxima of an operand (the fp16 scales of the GEMM-epilogue kernels) are handed from kernel to kernel for batches of
// at least this many rows; below, the GEMM reads its rows once more itself (they sit in the L2, and the steps of such
// batches are bound by the host's launch rate: every pass or memset saved counts more than the re-read).
constexpr uint32_t kAmaxHandoverRows = 32768;

// amax (may be NULL): joined with the row maxima of Y.  amax_state 0: zeroed here first; 1: the caller has zeroed it; 2: it
// holds the maxima of other columns of the same operand.  Kernels without the atomic form get a separate pass (not for 2).
int spmm_any(const sl_norm_adj *a, bool transposed, const float *X, int64_t ldx, float *Y, int64_t ldy, uint32_t F, void *st,
             float *amax = nullptr, int amax_state = 0, bool zero_pad = false) {
  const uint32_t *ip = transposed ? a->t_indptr : a->indptr, *ix = transposed ? a->t_indices : a->indices;
  const uint32_t *perm = (transposed && a->edge_w) ? a->t_perm : nullptr;
  // (diag(rs) W diag(cs))^T = diag(cs) W^T diag(rs)
  const float *rs = transposed ? a->col_scale : a->row_scale, *cs = transposed ? a->row_scale : a->col_scale;
  // algorithmic bytes: indptr + indices (+ edge values) + read X + write A.X
  SHD_PROF_FMT(4.0 * (a->n + 1) + 4.0 * a->e + (a->edge_w ? 4.0 * a->e : 0.0) + 8.0 * a->n * F, 0, st, "spmm_F%u", F);
  if (!zero_pad && spmm_wide_pipe(a) && spmm_csr_whole_rows(F, X, ldx, Y, ldy)) {
    // (the kernel writes every row's maximum: nothing to clear; amax_state 2: joined with what the array holds)
    return spmm_csr_amax(ip, ix, a->edge_w, perm, rs, cs, X, ldx, Y, ldy, a->n, F, amax, amax_state == 2 ? 1 : 0, st);
  }
  if (a->subg_node_off && F >= 96) {
    float *am = spmm_joins(a, F, X, ldx, Y, ldy) ? amax : nullptr;
    if (am && amax_state == 0) { const int frc = fill_words(am, 0u, (size_t)a->n, (hipStream_t)st); if (frc != SG_OK) return frc; }
    const int rc = zero_pad ? spmm_blockdiag_padded(ip, ix, a->edge_w, perm, rs, cs, X, ldx, Y, ldy, a->n, F, a->subg_node_off, a->subg_edge_off,
                                                    a->num_subg, a->max_subg_nodes, am, st)
                            : sl_spmm_blockdiag_f32(ip, ix, a->edge_w, perm, rs, cs, X, ldx, Y, ldy, a->n, F, a->subg_node_off, a->subg_edge_off,
                                                    a->num_subg, a->max_subg_nodes, am, st);
    if (rc != SG_OK || !amax || am) return rc;
  } else {
    const int rc = sl_spmm_csr_f32(ip, ix, a->edge_w, perm, rs, cs, X, ldx, Y, ldy, a->n, F, st);
    if (rc != SG_OK || !amax) return rc;
  }
  if (amax_state == 2) return set_error(SG_ERR_INVALID, "spmm: joined row maxima need the block-diagonal vector kernel");
  return sl_row_amax(Y, ldy, a->n, F, amax, st);
}

// the primitives, each under its profiling scope (names and byte / flop counts as ops.py gives them to KernelTimer)
int nt_gemm(const float *A, int64_t lda, const void *pk, float *Cm, int64_t ldc, uint32_t M, uint32_t N, uint32_t K, void *st) {
  SHD_PROF_FMT(4.0 * M * (K + N), 2.0 * M * K * N, st, "gemm_nt_split_N%u%s", N, K % 32 ? "_Ktail" : "");
  return sl_gemm_nt_f32(A, lda, pk, Cm, ldc, M, N, K, st);
}

int tn_gemm(const float *A, int64_t lda, const float *B, int64_t ldb, float *Cm, uint32_t M, uint32_t N, uint32_t K, float *partial, void *st) {
  SHD_PROF_FMT(4.0 * M * (N + K), 2.0 * M * N * K, st, "gemm_tn_split_N%u%s", N, K <= 128 ? "_K128" : "");
  return sl_gemm_tn_f32(A, lda, B, ldb, Cm, M, N, K, partial, nullptr, st);
}

// zeros into the F-wide column slices [b] of a pitched buffer (F % 4 == 0, 16-byte aligned): the rows of dZs / dZn a sparse
// read-out gradient does not reach (hipMemset2DAsync on a pitched region ran at 0.8 TB/s)
__global__ void __launch_bounds__(256) zero_slices_kernel(float *a, float *b, int64_t ld, uint32_t n, uint32_t F) {
  const uint32_t f4 = F / 4;
  const uint64_t total = (uint64_t)n * f4 * 2;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (uint64_t)gridDim.x * 256) {
    const uint64_t half = (uint64_t)n * f4;
    float *base = i < half ? a : b;
    const uint64_t j = i < half ? i : i - half;
    *reinterpret_cast<float4 *>(base + (j / f4) * ld + (j % f4) * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

}  // namespace

// Zeros into the F-wide column slices a and b (may be NULL) of a pitched buffer: what a caller that fills a few rows of a
// [dZs | . | dZn] buffer itself (the row-sparse top-layer backward of ops._SageDense) clears around them.
extern "C" int sl_zero_slices(float *d_a, float *d_b, int64_t ld, uint32_t n, uint32_t F, void *stream) {
  if (n == 0 || F == 0 || (!d_a && !d_b)) return SG_OK;
  if ((F & 3) || (ld & 3) || (reinterpret_cast<uintptr_t>(d_a) & 15) || (reinterpret_cast<uintptr_t>(d_b) & 15))
    return set_error(SG_ERR_INVALID, "sl_zero_slices: needs F %% 4 == 0, ld %% 4 == 0 and 16-byte aligned slices");
  float *a = d_a ? d_a : d_b, *b = d_b ? d_b : d_a;          // (one slice: cleared twice -- idempotent)
  SHD_PROF_FMT(2.0 * 4.0 * n * F, 0, stream, "zero_slices_F%u", F);
  hipLaunchKernelGGL(zero_slices_kernel, dim3(256 * 8), dim3(256), 0, (hipStream_t)stream, a, b, ld, n, F);
  SHD_HIP(hipGetLastError());
  return SG_OK;
}

namespace {

// the GEMM-epilogue forms (gemm_fused.hip) take 16-byte aligned operands with row pitches of whole float4s
bool fused_epilogue_ok(uint32_t Fout, uint32_t Fin, const float *A0, int64_t lda0, const float *A1, int64_t lda1) {
  auto ok = [](const float *p, int64_t ld) { return !p || ((reinterpret_cast<uintptr_t>(p) & 15) == 0 && (ld & 3) == 0); };
  return sl_gemm_act_norm_supported(Fout, Fin) && ok(A0, lda0) && ok(A1, lda1);
}

}  // namespace

// scratch of one layer pass: the weight images, then (16-byte aligned) the operands' row scales (2 n floats)
static size_t images_bytes_sage(uint32_t Fin, uint32_t Fout) {
  // forward: Ws and Wn images ([Fout, Fin] each); backward: the [Fin, 2 Fout] image of [Ws^T | Wn^T]
  // (fp16 epilogue-kernel images or, for widths those do not take, the bf16 images of sl_gemm_nt_f32)
  const size_t fwd = std::max(2 * sl_gemm_act_norm_pack_bytes(Fout, Fin), 2 * sl_gemm_pack_bytes(Fout, Fin));
  const size_t bwd = std::max(sl_gemm_act_norm_pack_bytes(Fin, 2 * Fout), sl_gemm_pack_bytes(Fin, 2 * Fout));
  return (std::max(fwd, bwd) + 15) & ~(size_t)15;
}

extern "C" int sl_set_spmm_wide_pipe(int on) {
  const int prev = g_spmm_wide_pipe;
  if (on >= 0) g_spmm_wide_pipe = on > 2 ? 2 : on;
  return prev;
}

extern "C" size_t sl_sage_pack_bytes(uint32_t n, uint32_t Fin, uint32_t Fout) {
  return images_bytes_sage(Fin, Fout) + (size_t)2 * n * 4;
}

extern "C" int sl_sage_fwd(const sl_norm_adj *adj, const float *d_X, int64_t ldx, uint32_t Fin, uint32_t Fout,
                           const float *d_Ws, int64_t ldws, const float *d_bs, const float *d_Wn, int64_t ldwn,
                           const float *d_bn, const float *d_scale, const float *d_offset, int act, float drop_p,
                           uint64_t drop_seed, float *d_AX, int64_t ldax, float *d_Zs, float *d_Zn, float *d_out,
                           float *d_out_dropped, const float *d_x_amax, float *d_out_amax, void *d_pack, int x_pad_zero,
                           float *d_row_stats, void *stream) {
  if (!adj || !d_X || !d_Ws || !d_Wn || !d_scale || !d_offset || !d_AX || !d_Zs || !d_Zn || !d_out || !d_pack)
    return set_error(SG_ERR_INVALID, "sl_sage_fwd: null argument");
  if (Fout > 256 || (Fout & 3) || Fin == 0) return set_error(SG_ERR_INVALID, "sl_sage_fwd: Fout = %u (multiple of 4, at most 256)", Fout);
  const uint32_t n = adj->n;
  if (n == 0) return SG_OK;
  int rc;
  char *pk = (char *)d_pack;
  const bool fused = fused_epilogue_ok(Fout, Fin, d_X, ldx, d_AX, ldax);
  const int64_t ldz[2] = {Fout, Fout};
  const float *bias[2] = {d_bs, d_bn};
  const int acts[2] = {act, act};
  if (fused) {
    // both products in one launch, bias / act / norm / branch sum (/ dropout) in its epilogue (gemm_fused.hip).
    // Row maxima of the two A operands (fp16 split, gemm_common.h): X's come with it when its producer wrote them, the SpMM
    // joins those of A X into an array the weight pack has cleared on its way.
    float *amx = reinterpret_cast<float *>(pk + images_bytes_sage(Fin, Fout));
    const bool hand = n >= kAmaxHandoverRows;
    const bool joins = hand && spmm_joins(adj, Fin, d_X, ldx, d_AX, ldax);
    const float *W[2] = {d_Ws, d_Wn};
    const int64_t ldw[2] = {ldws, ldwn};
    if ((rc = sl_gemm_act_norm_pack(2, W, ldw, Fout, Fin, pk, joins ? amx + n : nullptr, joins ? n : 0, stream)) != SG_OK) return rc;
    // Rows narrower than their 128-byte lines (layer 0: 100 floats in 512 B) whose pad is known to be zero (x_pad_zero: the
    // caller's word for X; the aggregation writes A X's): the products read them at the padded width -- no K tail, no
    // predicated A loads (371 -> 319 us at 289 k rows) -- against the SAME weight images (their tail columns are zero).
    const uint32_t Fp = (Fin + 31u) & ~31u;
    const bool kpad = x_pad_zero && Fin % 32 && ldx >= (int64_t)Fp && ldax >= (int64_t)Fp && adj->subg_node_off && Fin >= 96 &&
                      spmm_blockdiag_lines_ok(Fin, d_X, ldx, d_AX, ldax);
    if ((rc = spmm_any(adj, false, d_X, ldx, d_AX, ldax, Fin, stream, hand ? amx + n : nullptr, joins ? 1 : 0, kpad)) != SG_OK) return rc;
    if (hand && !d_x_amax && (rc = sl_row_amax(d_X, ldx, n, Fin, amx, stream)) != SG_OK) return rc;
    const float *A[2] = {d_X, d_AX};
    const float *asc[2] = {d_x_amax ? d_x_amax : (hand ? amx : nullptr), hand ? amx + n : nullptr};
    const int64_t lda[2] = {ldx, ldax};
    float *Zw[2] = {d_Zs, d_Zn};
    SHD_PROF_FMT(4.0 * n * (2 * Fin + 2 * Fout + Fout * (d_out_dropped ? 2 : 1)), 2.0 * 2 * n * Fin * Fout, stream, "gemm_act_norm_fwd_nb%d_N%u%s", 2, Fout, Fin % 32 ? "_Ktail" : "");
    return sl_gemm_act_norm_fwd(2, A, lda, asc, pk, n, Fout, kpad ? Fp : Fin, Zw, ldz, bias, acts, d_scale, d_offset, 1.0f, d_out, Fout, drop_p,
                                drop_seed, d_out_dropped, Fout, d_out_amax, d_row_stats, stream);
  }
  if (d_row_stats) return set_error(SG_ERR_INVALID, "sl_sage_fwd: row statistics come from the GEMM-epilogue kernel only (sl_gemm_act_norm_supported, "
                                     "16-byte aligned operands): pass d_row_stats = NULL for this shape");
  if ((rc = spmm_any(adj, false, d_X, ldx, d_AX, ldax, Fin, stream)) != SG_OK) return rc;
  const size_t pb = sl_gemm_pack_bytes(Fout, Fin);
  if ((rc = sl_gemm_pack_b(d_Ws, ldws, Fout, Fin, pk, stream)) != SG_OK) return rc;
  if ((rc = sl_gemm_pack_b(d_Wn, ldwn, Fout, Fin, pk + pb, stream)) != SG_OK) return rc;
  if ((rc = nt_gemm(d_X, ldx, pk, d_Zs, Fout, n, Fout, Fin, stream)) != SG_OK) return rc;
  if ((rc = nt_gemm(d_AX, ldax, pk + pb, d_Zn, Fout, n, Fout, Fin, stream)) != SG_OK) return rc;
  const float *Z[2] = {d_Zs, d_Zn};
  SHD_PROF_FMT((2 + 1) * 4.0 * n * Fout, 0, stream, "act_norm_fwd_nb%d_F%u", 2, Fout);
  return sl_act_norm_fwd(2, Z, ldz, bias, acts, d_scale, d_offset, n, Fout, Fout, 1.0f, d_out, Fout, drop_p, drop_seed, d_out_dropped,
                         Fout, d_out_amax, stream);
}

extern "C" size_t sl_sage_chain_partial_floats(uint32_t n, uint32_t F) { return sl_gemm_an_bwd_partial_floats(n, F, 2); }

// dz_ready: the layer above already left this layer's dZs / dZn in d_buf and its dscale / doffset / dbias (its own call
// had `below` pointing here); below != NULL: the input gradient is not written -- the epilogue of the K = 2 Fout product
// turns it into the dZs / dZn of the layer below on the fly (sl_gemm_an_bwd).
extern "C" int sl_sage_bwd_chain(const sl_norm_adj *adj, const float *d_X, int64_t ldx, const float *d_AX, int64_t ldax,
                                 const float *d_Zs, const float *d_Zn, uint32_t Fin, uint32_t Fout, const float *d_Ws, int64_t ldws,
                                 const float *d_bs, const float *d_Wn, int64_t ldwn, const float *d_bn, const float *d_scale,
                                 const float *d_offset, int act, float drop_p, uint64_t drop_seed, const float *d_dout,
                                 const float *d_dout_dropped, float *d_dX, float *d_dWs, float *d_dWn, float *d_dbias,
                                 float *d_dscale, float *d_doffset, float *d_buf, float *d_an_partial, float *d_tn_partial,
                                 void *d_pack, int dz_ready, const sl_sage_below *below, float *d_dzs_amax,
                                 const uint32_t *d_dout_rows, uint32_t num_dout_rows, const float *d_x_amax, const uint32_t *d_dout_map,
                                 void *stream) {
  if (d_dout_map && (d_dout_rows || dz_ready || !d_dout)) return set_error(SG_ERR_INVALID, "sl_sage_bwd: a gradient table comes with d_dout alone");
  // (d_dWs == d_dWn == NULL: the caller computes the weight gradients itself -- on the few rows dZ is non-zero on, see
  //  ops._SageDense: the layer below a row-sparse top pass)
  const bool want_dw = d_dWs || d_dWn;
  if (!adj || !d_X || !d_AX || !d_Ws || !d_Wn || (want_dw && (!d_dWs || !d_dWn || !d_tn_partial)) || !d_buf || !d_pack)
    return set_error(SG_ERR_INVALID, "sl_sage_bwd: null argument");
  if (!dz_ready && (!d_Zs || !d_Zn || !d_scale || !d_offset || !d_dscale || !d_doffset || !d_an_partial || (!d_dout && !d_dout_dropped)))
    return set_error(SG_ERR_INVALID, "sl_sage_bwd: null argument");
  if (Fout > 256 || (Fout & 3) || Fin > 256 || (Fin & 3)) return set_error(SG_ERR_INVALID, "sl_sage_bwd: widths %u -> %u unsupported", Fin, Fout);
  if ((d_dX || below) && ((2 * Fout) % 32 || Fout % 32)) return set_error(SG_ERR_INVALID, "sl_sage_bwd: the input gradient needs Fout %% 32 == 0");
  if (below && (below->F != Fin || !below->Zs || !below->Zn || !below->scale || !below->offset || !below->buf || !below->dscale ||
                !below->doffset || !below->partial || !below->amax))
    return set_error(SG_ERR_INVALID, "sl_sage_bwd: incomplete description of the layer below");
  const uint32_t n = adj->n;
  if (n == 0) return SG_OK;
  int rc;
  // dZs lives in the left third of buf [n, 3 Fout], dZn in the right third (the transposed SpMM reads it and writes
  // A^T dZn into the middle third): [dZs | A^T dZn | dZn]
  float *dZs = d_buf, *dZn = d_buf + 2 * (size_t)Fout;
  const int64_t ld3 = 3 * (int64_t)Fout;
  // (the row maxima of the K = 2 Fout operand of the epilogue form, see below: the act_norm backward writes those of dZs)
  float *amx = reinterpret_cast<float *>(reinterpret_cast<char *>(d_pack) + images_bytes_sage(Fin, Fout));
  const bool hand = n >= kAmaxHandoverRows;
  // the input-gradient product on the fp16 kernels: with the lower layer's act_norm backward in its epilogue (below), or plain
  const bool f16dx = below || (d_dX && fused_epilogue_ok(Fin, 2 * Fout, d_buf, ld3, nullptr, 0));
  const bool join = f16dx && hand && adj->t_indptr && spmm_joins(adj, Fout, dZn, ld3, d_buf + Fout, ld3);
  if (!dz_ready) {
    const float *Z[2] = {d_Zs, d_Zn};
    const int64_t ldz[2] = {Fout, Fout};
    const float *bias[2] = {d_bs, d_bn};
    const int acts[2] = {act, act};
    float *dZ[2] = {dZs, dZn};
    const int64_t lddz[2] = {ld3, ld3};
    if (d_dout_rows) {
      // The output gradient exists for a few rows only (a read-out that takes the roots' rows: the reference's feat[rows]
      // hands autograd a zero-filled [n, F] tensor).  Every other row of dZs / dZn is zero: cleared here, and the act_norm
      // backward runs on the selected rows (compact gradient, row indirection).
      SHD_PROF_FMT(2.0 * 4.0 * n * Fout + 5.0 * 4.0 * num_dout_rows * Fout, 0, stream, "act_norm_bwd_rows_nb%d_F%u", 2, Fout);
      hipLaunchKernelGGL(zero_slices_kernel, dim3(256 * 8), dim3(256), 0, (hipStream_t)stream, dZs, dZn, ld3, n, Fout);
      if (join && (rc = fill_words(amx, 0u, (size_t)n, (hipStream_t)stream)) != SG_OK) return rc;
      if (num_dout_rows == 0) {
        SHD_HIP(hipMemsetAsync(d_dscale, 0, (size_t)2 * Fout * 4, (hipStream_t)stream));
        SHD_HIP(hipMemsetAsync(d_doffset, 0, (size_t)2 * Fout * 4, (hipStream_t)stream));
        if (d_dbias) SHD_HIP(hipMemsetAsync(d_dbias, 0, (size_t)2 * Fout * 4, (hipStream_t)stream));
      } else if ((rc = sl_act_norm_bwd(2, Z, ldz, bias, acts, d_scale, d_offset, num_dout_rows, Fout, Fout, 1.0f, d_dout, Fout, dZ, lddz,
                                       d_dscale, d_doffset, d_dbias, d_an_partial, drop_p, drop_seed, d_dout_dropped, Fout,
                                       join ? amx : nullptr, d_dout_rows, stream)) != SG_OK) {
        return rc;
      }
    } else {
      SHD_PROF_FMT((2 * 2 + 1) * 4.0 * n * Fout, 0, stream, "act_norm_bwd_nb%d_F%u", 2, Fout);
      if ((rc = sl_act_norm_bwd_map(2, Z, ldz, bias, acts, d_scale, d_offset, n, Fout, Fout, 1.0f, d_dout, Fout, d_dout_map, dZ, lddz, d_dscale,
                                    d_doffset, d_dbias, d_an_partial, drop_p, drop_seed, d_dout_dropped, Fout, join ? amx : nullptr,
                                    stream)) != SG_OK)
        return rc;
    }
  }
  if (d_dX || below) {
    if (!adj->t_indptr) return set_error(SG_ERR_INVALID, "sl_sage_bwd: the input gradient needs the transposed adjacency");
    // The K = 2 Fout operand [dZs | A^T dZn] of the epilogue form needs its row maxima: those of dZs come from the kernel
    // that wrote it (the layer above when dz_ready, the act_norm backward otherwise); the transposed SpMM joins those of its half.
    if (join && dz_ready) {
      if (d_dzs_amax) amx = d_dzs_amax;
      else if ((rc = sl_row_amax(dZs, ld3, n, Fout, amx, stream)) != SG_OK) return rc;
    }
    if ((rc = spmm_any(adj, true, dZn, ld3, d_buf + Fout, ld3, Fout, stream, join ? amx : nullptr, 2)) != SG_OK) return rc;
    if (f16dx && hand && !join && (rc = sl_row_amax(d_buf, ld3, n, 2 * Fout, amx, stream)) != SG_OK) return rc;
    // dX = [dZs | A^T dZn] . [Ws ; Wn]   (K = 2 Fout)
    if (f16dx) rc = sl_gemm_act_norm_pack_b2(d_Ws, 1, ldws, Fout, d_Wn, 1, ldwn, Fin, 2 * Fout, d_pack, stream);
    else rc = sl_gemm_pack_b2(d_Ws, 1, ldws, Fout, d_Wn, 1, ldwn, Fin, 2 * Fout, d_pack, stream);
    if (rc != SG_OK) return rc;
    if (below) {
      const uint32_t Fb = below->F;
      const float *Zb[2] = {below->Zs, below->Zn};
      const int64_t ldzb[2] = {Fb, Fb};
      const float *biasb[2] = {below->bs, below->bn};
      const int actsb[2] = {below->act, below->act};
      float *dZb[2] = {below->buf, below->buf + 2 * (size_t)Fb};
      const int64_t lddzb[2] = {3 * (int64_t)Fb, 3 * (int64_t)Fb};
      // read [dZs | A^T dZn] and both Z of the layer below, write its two dZ
      SHD_PROF_FMT(4.0 * n * (2.0 * Fout + 4.0 * Fb), 2.0 * n * (2.0 * Fout) * Fin, stream, "gemm_an_bwd_nb2_N%u", Fin);
      if ((rc = sl_gemm_an_bwd_plain(d_buf, ld3, hand ? amx : nullptr, d_pack, n, Fin, 2 * Fout, 2, Zb, ldzb, biasb, actsb, below->scale, below->offset, 1.0f,
                                     dZb, lddzb, below->dscale, below->doffset, below->dbias, below->partial, below->drop_p, below->drop_seed,
                                     hand ? below->amax : nullptr, below->stats, nullptr, 0, nullptr, 0, below->dout_plain, below->plain_ld ? below->plain_ld : (int64_t)Fb, below->plain_row, stream)) != SG_OK)
        return rc;
    } else if (f16dx) {
      const float *A1[1] = {d_buf};
      const int64_t lda1[1] = {ld3}, ldc1[1] = {Fin};
      const float *am1[1] = {hand ? amx : nullptr};
      float *C1[1] = {d_dX};
      SHD_PROF_FMT(4.0 * n * (2.0 * Fout + Fin), 2.0 * n * (2.0 * Fout) * Fin, stream, "gemm_nt_f16_N%u", Fin);
      if ((rc = sl_gemm_nt2_f32(1, A1, lda1, am1, d_pack, n, Fin, 2 * Fout, nullptr, C1, ldc1, stream)) != SG_OK) return rc;
    } else if ((rc = nt_gemm(d_buf, ld3, d_pack, d_dX, Fin, n, Fin, 2 * Fout, stream)) != SG_OK) {
      return rc;
    }
  }
  if (!want_dw) return SG_OK;
  // Weight gradients.  With the row maxima of [dZs | A^T dZn] (amx: the K = 2 Fout operand of the input gradient above) and
  // of X in hand, both run on two fp16 pieces (sl_gemm_tn_f16) -- the neighbour branch as
  //     dWn = dZn^T (A X) = (A^T dZn)^T X
  // over the transposed aggregate that is already there: same operand X, no row maxima of dZn / A X needed.
  if (d_x_amax && f16dx && hand && Fin == 256 && Fout == 256 && (ldx & 3) == 0 && (reinterpret_cast<uintptr_t>(d_X) & 15) == 0 &&
      (n + sl_gemm_tn_slices(n) - 1) / sl_gemm_tn_slices(n) <= 3024) {
    // (one launch for both: the two workgroups of a row slice share an XCD's L2, X comes from HBM once -- sl_gemm_tn_f16_pair)
    SHD_PROF_FMT(4.0 * n * (2 * Fout + Fin), 2.0 * 2 * n * Fout * Fin, stream, "gemm_tn_f16_pair_N%u", Fout);
    return sl_gemm_tn_f16_pair(dZs, d_buf + Fout, ld3, amx, d_X, ldx, d_x_amax, d_dWs, d_dWn, n, Fout, Fin, d_tn_partial, nullptr, nullptr, stream);
  }
  if ((rc = tn_gemm(dZs, ld3, d_X, ldx, d_dWs, n, Fout, Fin, d_tn_partial, stream)) != SG_OK) return rc;
  return tn_gemm(dZn, ld3, d_AX, ldax, d_dWn, n, Fout, Fin, d_tn_partial, stream);
}

extern "C" int sl_sage_bwd(const sl_norm_adj *adj, const float *d_X, int64_t ldx, const float *d_AX, int64_t ldax,
                           const float *d_Zs, const float *d_Zn, uint32_t Fin, uint32_t Fout, const float *d_Ws, int64_t ldws,
                           const float *d_bs, const float *d_Wn, int64_t ldwn, const float *d_bn, const float *d_scale,
                           const float *d_offset, int act, float drop_p, uint64_t drop_seed, const float *d_dout,
                           const float *d_dout_dropped, float *d_dX, float *d_dWs, float *d_dWn, float *d_dbias,
                           float *d_dscale, float *d_doffset, float *d_buf, float *d_an_partial, float *d_tn_partial,
                           void *d_pack, void *stream) {
  return sl_sage_bwd_chain(adj, d_X, ldx, d_AX, ldax, d_Zs, d_Zn, Fin, Fout, d_Ws, ldws, d_bs, d_Wn, ldwn, d_bn, d_scale, d_offset, act,
                           drop_p, drop_seed, d_dout, d_dout_dropped, d_dX, d_dWs, d_dWn, d_dbias, d_dscale, d_doffset, d_buf,
                           d_an_partial, d_tn_partial, d_pack, 0, nullptr, nullptr, nullptr, 0, nullptr, nullptr, stream);
}

// ---------------------------------------------------------------------------------------------------------------
// A whole stack of GraphSAGE layers per call (the conv loop of shaDow/models.py:193-197 over layers.py:471-483 layers whose only
// connection is out_l -> X_{l+1}: residue 'none' + centre pooling).  The same per-layer entries in the same order with the same
// arguments the host would pass one by one (identical results); what goes away is the host work between them -- at the
// reference's own batch sizes (16-256 roots) a step is bound by it.
// ---------------------------------------------------------------------------------------------------------------
extern "C" size_t sl_sage_stack_pack_bytes(uint32_t n, uint32_t L, const sl_sage_stack_layer *ly) {
  size_t b = 0;
  for (uint32_t l = 0; ly && l < L; ++l) b = std::max(b, sl_sage_pack_bytes(n, ly[l].Fin, ly[l].Fout));
  return b;
}

static int stack_check(const sl_norm_adj *adj, const float *d_X0, uint32_t L, const sl_sage_stack_layer *ly, const char *who) {
  if (!adj || !d_X0 || !ly || L == 0) return set_error(SG_ERR_INVALID, "%s: null argument", who);
  for (uint32_t l = 1; l < L; ++l)
    if (ly[l].Fin != ly[l - 1].Fout) return set_error(SG_ERR_INVALID, "%s: layer %u reads %u columns, layer %u writes %u", who, l, ly[l].Fin, l - 1, ly[l - 1].Fout);
  return SG_OK;
}

extern "C" int sl_sage_stack_fwd(const sl_norm_adj *adj, const float *d_X0, int64_t ldx0, const float *d_x0_amax, int x0_pad_zero,
                                 uint32_t L, const sl_sage_stack_layer *ly, void *d_pack, void *stream) {
  int rc;
  if ((rc = stack_check(adj, d_X0, L, ly, "sl_sage_stack_fwd")) != SG_OK) return rc;
  const float *X = d_X0, *xam = d_x0_amax;
  int64_t ldx = ldx0;
  for (uint32_t l = 0; l < L; ++l) {
    const sl_sage_stack_layer &y = ly[l];
    if ((rc = sl_sage_fwd(adj, X, ldx, y.Fin, y.Fout, y.Ws, y.ldws, y.bs, y.Wn, y.ldwn, y.bn, y.scale, y.offset, y.act, y.drop_p, y.drop_seed,
                          y.AX, y.ldax, y.Zs, y.Zn, y.out, nullptr, xam, y.out_amax, d_pack, l == 0 ? x0_pad_zero : 0, y.row_stats,
                          stream)) != SG_OK)
      return rc;
    X = y.out, ldx = y.Fout, xam = y.out_amax;
  }
  return SG_OK;
}

extern "C" int sl_sage_stack_bwd(const sl_norm_adj *adj, const float *d_X0, int64_t ldx0, const float *d_x0_amax, uint32_t L,
                                 const sl_sage_stack_layer *ly, const float *d_dout, const uint32_t *d_dout_rows, uint32_t num_dout_rows,
                                 float *d_dX0, float *d_buf, float *d_amax, float *d_an_partial, float *d_chain_partial,
                                 float *d_tn_partial, void *d_pack, void *stream) {
  int rc;
  if ((rc = stack_check(adj, d_X0, L, ly, "sl_sage_stack_bwd")) != SG_OK) return rc;
  if (!d_dout || !d_buf || !d_amax || !d_an_partial || !d_tn_partial || !d_pack || (L > 1 && !d_chain_partial))
    return set_error(SG_ERR_INVALID, "sl_sage_stack_bwd: null argument");
  const uint32_t n = adj->n;
  uint32_t Fmax = 0;
  for (uint32_t l = 0; l < L; ++l) Fmax = std::max(Fmax, ly[l].Fout);
  // [dZs | A^T dZn | dZn] of layer l lives in half (l & 1) of d_buf: written by layer l + 1's call (or layer l's own act_norm
  // backward at the top), read by layer l's -- two halves in turn do for the whole stack
  const size_t half = (size_t)n * 3 * Fmax;
  for (uint32_t l = L; l-- > 0;) {
    const sl_sage_stack_layer &y = ly[l];
    const bool top = l + 1 == L;
    float *buf = d_buf + (l & 1u) * half, *am = d_amax + (l & 1u) * (size_t)n;
    sl_sage_below below;
    if (l > 0) {
      const sl_sage_stack_layer &b = ly[l - 1];
      if (y.Fout % 32) return set_error(SG_ERR_INVALID, "sl_sage_stack_bwd: chained layers need Fout %% 32 == 0 (layer %u: %u)", l, y.Fout);
      below.Zs = b.Zs, below.Zn = b.Zn, below.bs = b.bs, below.bn = b.bn, below.scale = b.scale, below.offset = b.offset;
      below.act = b.act, below.drop_p = b.drop_p, below.drop_seed = b.drop_seed, below.F = b.Fout;
      below.buf = d_buf + ((l - 1) & 1u) * half, below.dscale = b.dscale, below.doffset = b.doffset, below.dbias = b.dbias;
      below.partial = d_chain_partial, below.amax = d_amax + ((l - 1) & 1u) * (size_t)n, below.stats = b.row_stats; below.dout_plain = nullptr; below.plain_row = nullptr; below.plain_ld = 0;
    }
    const float *X = l ? ly[l - 1].out : d_X0;
    if ((rc = sl_sage_bwd_chain(adj, X, l ? (int64_t)ly[l - 1].Fout : ldx0, y.AX, y.ldax, y.Zs, y.Zn, y.Fin, y.Fout, y.Ws, y.ldws, y.bs, y.Wn,
                                y.ldwn, y.bn, y.scale, y.offset, y.act, y.drop_p, y.drop_seed, top ? d_dout : nullptr, nullptr,
                                l == 0 ? d_dX0 : nullptr, y.dWs, y.dWn, y.dbias, y.dscale, y.doffset, buf, d_an_partial, d_tn_partial, d_pack,
                                top ? 0 : 1, l > 0 ? &below : nullptr, top ? nullptr : am, top ? d_dout_rows : nullptr,
                                top ? num_dout_rows : 0, l ? ly[l - 1].out_amax : d_x0_amax, nullptr, stream)) != SG_OK)
      return rc;
  }
  return SG_OK;
}

// The lower part of a stack whose top layers ran row-sparse (round 5): layers 0 .. L - 1 of `ly`, layer L - 1's
// [dZs | . | dZn], its row maxima and its dscale / doffset / dbias already produced by the caller (the K = F product with the
// sparse addend of the layer above, sl_gemm_an_bwd_corr) in d_top_buf / d_top_amax; from there down the chained dense passes
// of sl_sage_stack_bwd.
extern "C" int sl_sage_stack_bwd_ready(const sl_norm_adj *adj, const float *d_X0, int64_t ldx0, const float *d_x0_amax, uint32_t L,
                                       const sl_sage_stack_layer *ly, float *d_top_buf, const float *d_top_amax, float *d_dX0, float *d_buf,
                                       float *d_amax, float *d_chain_partial, float *d_tn_partial, void *d_pack, void *stream) {
  int rc;
  if ((rc = stack_check(adj, d_X0, L, ly, "sl_sage_stack_bwd_ready")) != SG_OK) return rc;
  if (!d_top_buf || !d_buf || !d_amax || !d_tn_partial || !d_pack || (L > 1 && !d_chain_partial))
    return set_error(SG_ERR_INVALID, "sl_sage_stack_bwd_ready: null argument");
  const uint32_t n = adj->n;
  uint32_t Fmax = 0;
  for (uint32_t l = 0; l < L; ++l) Fmax = std::max(Fmax, ly[l].Fout);
  const size_t half = (size_t)n * 3 * Fmax;
  for (uint32_t l = L; l-- > 0;) {
    const sl_sage_stack_layer &y = ly[l];
    const bool top = l + 1 == L;
    float *buf = top ? d_top_buf : d_buf + (l & 1u) * half;
    float *am = top ? const_cast<float *>(d_top_amax) : d_amax + (l & 1u) * (size_t)n;
    sl_sage_below below;
    if (l > 0) {
      const sl_sage_stack_layer &b = ly[l - 1];
      if (y.Fout % 32) return set_error(SG_ERR_INVALID, "sl_sage_stack_bwd_ready: chained layers need Fout %% 32 == 0 (layer %u: %u)", l, y.Fout);
      below.Zs = b.Zs, below.Zn = b.Zn, below.bs = b.bs, below.bn = b.bn, below.scale = b.scale, below.offset = b.offset;
      below.act = b.act, below.drop_p = b.drop_p, below.drop_seed = b.drop_seed, below.F = b.Fout;
      below.buf = d_buf + ((l - 1) & 1u) * half, below.dscale = b.dscale, below.doffset = b.doffset, below.dbias = b.dbias;
      below.partial = d_chain_partial, below.amax = d_amax + ((l - 1) & 1u) * (size_t)n, below.stats = b.row_stats; below.dout_plain = nullptr; below.plain_row = nullptr; below.plain_ld = 0;
    }
    const float *X = l ? ly[l - 1].out : d_X0;
    if ((rc = sl_sage_bwd_chain(adj, X, l ? (int64_t)ly[l - 1].Fout : ldx0, y.AX, y.ldax, y.Zs, y.Zn, y.Fin, y.Fout, y.Ws, y.ldws, y.bs, y.Wn,
                                y.ldwn, y.bn, y.scale, y.offset, y.act, y.drop_p, y.drop_seed, nullptr, nullptr,
                                l == 0 ? d_dX0 : nullptr, y.dWs, y.dWn, y.dbias, y.dscale, y.doffset, buf, nullptr, d_tn_partial, d_pack,
                                1, l > 0 ? &below : nullptr, am, nullptr, 0, l ? ly[l - 1].out_amax : d_x0_amax, nullptr, stream)) != SG_OK)
      return rc;
  }
  return SG_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// GCN (shaDow/layers.py:417-444): out = norm(act((A X) W^T + b)).  Forward: SpMM, weight pack, GEMM, fused bias / act /
// norm; backward: act_norm backward, dAX = dZ W, dX = A^T dAX, dW = dZ^T (A X).
// ---------------------------------------------------------------------------------------------------------------
static size_t images_bytes_gcn(uint32_t Fin, uint32_t Fout) {
  const size_t fwd = std::max(sl_gemm_act_norm_pack_bytes(Fout, Fin), sl_gemm_pack_bytes(Fout, Fin)), bwd = sl_gemm_pack_bytes(Fin, Fout);
  return (std::max(fwd, bwd) + 15) & ~(size_t)15;
}

extern "C" size_t sl_gcn_pack_bytes(uint32_t n, uint32_t Fin, uint32_t Fout) { return images_bytes_gcn(Fin, Fout) + (size_t)n * 4; }

extern "C" int sl_gcn_fwd(const sl_norm_adj *adj, const float *d_X, int64_t ldx, uint32_t Fin, uint32_t Fout, const float *d_W,
                          int64_t ldw, const float *d_b, const float *d_scale, const float *d_offset, int act, float drop_p,
                          uint64_t drop_seed, float *d_AX, int64_t ldax, float *d_Z, float *d_out, float *d_out_dropped,
                          void *d_pack, void *stream) {
  if (!adj || !d_X || !d_W || !d_scale || !d_offset || !d_AX || !d_Z || !d_out || !d_pack)
    return set_error(SG_ERR_INVALID, "sl_gcn_fwd: null argument");
  if (Fout > 256 || (Fout & 3) || Fin == 0) return set_error(SG_ERR_INVALID, "sl_gcn_fwd: Fout = %u (multiple of 4, at most 256)", Fout);
  const uint32_t n = adj->n;
  if (n == 0) return SG_OK;
  int rc;
  const bool fused = fused_epilogue_ok(Fout, Fin, d_AX, ldax, nullptr, 0);
  float *amx = reinterpret_cast<float *>(reinterpret_cast<char *>(d_pack) + images_bytes_gcn(Fin, Fout));
  const bool hand = fused && n >= kAmaxHandoverRows;
  const bool joins = hand && spmm_joins(adj, Fin, d_X, ldx, d_AX, ldax);
  if (fused) {
    const float *W[1] = {d_W};
    const int64_t ldws[1] = {ldw};
    if ((rc = sl_gemm_act_norm_pack(1, W, ldws, Fout, Fin, d_pack, joins ? amx : nullptr, joins ? n : 0, stream)) != SG_OK) return rc;
  }
  if ((rc = spmm_any(adj, false, d_X, ldx, d_AX, ldax, Fin, stream, hand ? amx : nullptr, joins ? 1 : 0)) != SG_OK) return rc;
  const int64_t ldz[1] = {Fout};
  const float *bias[1] = {d_b};
  const int acts[1] = {act};
  if (fused) {
    const float *A[1] = {d_AX};
    const float *asc[1] = {hand ? amx : nullptr};
    const int64_t lda[1] = {ldax};
    float *Zw[1] = {d_Z};
    SHD_PROF_FMT(4.0 * n * (1 * Fin + 1 * Fout + Fout * (d_out_dropped ? 2 : 1)), 2.0 * 1 * n * Fin * Fout, stream, "gemm_act_norm_fwd_nb%d_N%u%s", 1, Fout, Fin % 32 ? "_Ktail" : "");
    return sl_gemm_act_norm_fwd(1, A, lda, asc, d_pack, n, Fout, Fin, Zw, ldz, bias, acts, d_scale, d_offset, 1.0f, d_out, Fout, drop_p,
                                drop_seed, d_out_dropped, Fout, nullptr, nullptr, stream);
  }
  if ((rc = sl_gemm_pack_b(d_W, ldw, Fout, Fin, d_pack, stream)) != SG_OK) return rc;
  if ((rc = nt_gemm(d_AX, ldax, d_pack, d_Z, Fout, n, Fout, Fin, stream)) != SG_OK) return rc;
  const float *Z[1] = {d_Z};
  SHD_PROF_FMT((1 + 1) * 4.0 * n * Fout, 0, stream, "act_norm_fwd_nb%d_F%u", 1, Fout);
  return sl_act_norm_fwd(1, Z, ldz, bias, acts, d_scale, d_offset, n, Fout, Fout, 1.0f, d_out, Fout, drop_p, drop_seed,
                         d_out_dropped, Fout, nullptr, stream);
}

extern "C" int sl_gcn_bwd(const sl_norm_adj *adj, const float *d_AX, int64_t ldax, const float *d_Z, uint32_t Fin, uint32_t Fout,
                          const float *d_W, int64_t ldw, const float *d_b, const float *d_scale, const float *d_offset, int act,
                          float drop_p, uint64_t drop_seed, const float *d_dout, const float *d_dout_dropped, float *d_dX,
                          int64_t lddx, float *d_dW, float *d_dbias, float *d_dscale, float *d_doffset, float *d_buf,
                          float *d_an_partial, float *d_tn_partial, void *d_pack, void *stream) {
  if (!adj || !d_AX || !d_Z || !d_W || !d_scale || !d_offset || !d_dW || !d_dscale || !d_doffset || !d_buf || !d_an_partial ||
      !d_tn_partial || !d_pack || (!d_dout && !d_dout_dropped))
    return set_error(SG_ERR_INVALID, "sl_gcn_bwd: null argument");
  if (Fout > 256 || (Fout & 3) || Fin > 256 || (Fin & 3)) return set_error(SG_ERR_INVALID, "sl_gcn_bwd: widths %u -> %u unsupported", Fin, Fout);
  const uint32_t n = adj->n;
  if (n == 0) return SG_OK;
  int rc;
  // buf = [dZ (n x Fout) | dAX (n x Fin)], both dense
  float *dZ = d_buf, *dAX = d_buf + (size_t)n * Fout;
  const float *Z[1] = {d_Z};
  const int64_t ldz[1] = {Fout};
  const float *bias[1] = {d_b};
  const int acts[1] = {act};
  float *dZs[1] = {dZ};
  const int64_t lddz[1] = {Fout};
  {
    SHD_PROF_FMT((2 * 1 + 1) * 4.0 * n * Fout, 0, stream, "act_norm_bwd_nb%d_F%u", 1, Fout);
    if ((rc = sl_act_norm_bwd(1, Z, ldz, bias, acts, d_scale, d_offset, n, Fout, Fout, 1.0f, d_dout, Fout, dZs, lddz, d_dscale,
                              d_doffset, d_dbias, d_an_partial, drop_p, drop_seed, d_dout_dropped, Fout, nullptr, nullptr, stream)) != SG_OK)
      return rc;
  }
  if (d_dX) {
    if (!adj->t_indptr) return set_error(SG_ERR_INVALID, "sl_gcn_bwd: the input gradient needs the transposed adjacency");
    // dAX = dZ . W: B image of W^T ([Fin, Fout] read transposed out of W), then dX = A^T dAX
    if ((rc = sl_gemm_pack_b2(d_W, 1, ldw, Fout, nullptr, 0, 0, Fin, Fout, d_pack, stream)) != SG_OK) return rc;
    if ((rc = nt_gemm(dZ, Fout, d_pack, dAX, Fin, n, Fin, Fout, stream)) != SG_OK) return rc;
    if ((rc = spmm_any(adj, true, dAX, Fin, d_dX, lddx, Fin, stream)) != SG_OK) return rc;
  }
  return tn_gemm(dZ, Fout, d_AX, ldax, d_dW, n, Fout, Fin, d_tn_partial, stream);
}

// ---------------------------------------------------------------------------------------------------------------
// A whole stack of GCN layers per call (the conv loop of shaDow/models.py:193-197 over layers.py:417-444 layers, residue 'none' +
// centre pooling): sl_gcn_fwd / sl_gcn_bwd layer by layer with the arguments the per-layer nodes pass; the input gradient of layer
// l is the output gradient of layer l - 1 (two buffers in turn).  The top layer's output gradient may come as (rows, values) of
// the read-out's row select: it is scattered into a cleared [n, Fout] buffer -- what autograd's zero-filled index_add builds.
// ---------------------------------------------------------------------------------------------------------------
namespace {
__global__ void scatter_rows_kernel(float *__restrict__ dst, int64_t ld, const uint32_t *__restrict__ rows, const float *__restrict__ src,
                                    uint32_t r, uint32_t F4) {
  const uint64_t total = (uint64_t)r * F4;
  for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t i = (uint32_t)(t / F4), c = (uint32_t)(t % F4) * 4u;
    *reinterpret_cast<float4 *>(dst + (int64_t)rows[i] * ld + c) = *reinterpret_cast<const float4 *>(src + (size_t)i * F4 * 4 + c);
  }
}
}  // namespace

extern "C" size_t sl_gcn_stack_pack_bytes(uint32_t n, uint32_t L, const sl_gcn_stack_layer *ly) {
  size_t b = 0;
  for (uint32_t l = 0; ly && l < L; ++l) b = std::max(b, sl_gcn_pack_bytes(n, ly[l].Fin, ly[l].Fout));
  return b;
}

static int gcn_stack_check(const sl_norm_adj *adj, uint32_t L, const sl_gcn_stack_layer *ly, const char *who) {
  if (!adj || !ly || L == 0) return set_error(SG_ERR_INVALID, "%s: null argument", who);
  for (uint32_t l = 1; l < L; ++l)
    if (ly[l].Fin != ly[l - 1].Fout) return set_error(SG_ERR_INVALID, "%s: layer %u reads %u columns, layer %u writes %u", who, l, ly[l].Fin, l - 1, ly[l - 1].Fout);
  return SG_OK;
}

extern "C" int sl_gcn_stack_fwd(const sl_norm_adj *adj, const float *d_X0, int64_t ldx0, uint32_t L, const sl_gcn_stack_layer *ly,
                                void *d_pack, void *stream) {
  int rc;
  if ((rc = gcn_stack_check(adj, L, ly, "sl_gcn_stack_fwd")) != SG_OK) return rc;
  if (!d_X0 || !d_pack) return set_error(SG_ERR_INVALID, "sl_gcn_stack_fwd: null argument");
  const float *X = d_X0;
  int64_t ldx = ldx0;
  for (uint32_t l = 0; l < L; ++l) {
    const sl_gcn_stack_layer &y = ly[l];
    if ((rc = sl_gcn_fwd(adj, X, ldx, y.Fin, y.Fout, y.W, y.ldw, y.b, y.scale, y.offset, y.act, y.drop_p, y.drop_seed, y.AX, y.ldax, y.Z,
                         y.out, nullptr, d_pack, stream)) != SG_OK)
      return rc;
    X = y.out, ldx = y.Fout;
  }
  return SG_OK;
}

extern "C" int sl_gcn_stack_bwd(const sl_norm_adj *adj, uint32_t L, const sl_gcn_stack_layer *ly, const float *d_dout,
                                const uint32_t *d_dout_rows, uint32_t num_dout_rows, float *d_dX0, float *d_grad, float *d_buf,
                                float *d_an_partial, float *d_tn_partial, void *d_pack, void *stream) {
  int rc;
  if ((rc = gcn_stack_check(adj, L, ly, "sl_gcn_stack_bwd")) != SG_OK) return rc;
  if (!d_dout || !d_grad || !d_buf || !d_an_partial || !d_tn_partial || !d_pack) return set_error(SG_ERR_INVALID, "sl_gcn_stack_bwd: null argument");
  const uint32_t n = adj->n;
  if (n == 0) return SG_OK;
  uint32_t Fmax = 0;
  for (uint32_t l = 0; l < L; ++l) Fmax = std::max(Fmax, std::max(ly[l].Fout, ly[l].Fin));
  const size_t half = (size_t)n * Fmax;
  // the gradient of layer l's output lives in half (l & 1) of d_grad: written by layer l + 1's call as its input gradient
  const float *dout = d_dout;
  if (d_dout_rows) {
    const uint32_t F = ly[L - 1].Fout;
    if (F & 3) return set_error(SG_ERR_INVALID, "sl_gcn_stack_bwd: the scattered output gradient needs Fout %% 4 == 0");
    float *dense = d_grad + ((L - 1) & 1u) * half;
    SHD_HIP(hipMemsetAsync(dense, 0, (size_t)n * F * 4, (hipStream_t)stream));
    if (num_dout_rows)
      hipLaunchKernelGGL(scatter_rows_kernel, dim3(std::min<uint32_t>((num_dout_rows * (F / 4) + 255) / 256, 2048)), dim3(256), 0,
                         (hipStream_t)stream, dense, (int64_t)F, d_dout_rows, d_dout, num_dout_rows, F / 4);
    dout = dense;
  }
  for (uint32_t l = L; l-- > 0;) {
    const sl_gcn_stack_layer &y = ly[l];
    float *dX = l ? d_grad + ((l - 1) & 1u) * half : d_dX0;
    if ((rc = sl_gcn_bwd(adj, y.AX, y.ldax, y.Z, y.Fin, y.Fout, y.W, y.ldw, y.b, y.scale, y.offset, y.act, y.drop_p, y.drop_seed, dout, nullptr,
                         dX, y.Fin, y.dW, y.dbias, y.dscale, y.doffset, d_buf, d_an_partial, d_tn_partial, d_pack, stream)) != SG_OK)
      return rc;
    dout = dX;
  }
  return SG_OK;
}
