// C-ABI entries of the 256 x 32 TN-tile single-pass bf16 product (gemm_rows.h): the GPT path's activation-by-weight products with row-major
// bf16 (or tiled-image) activations, fp32 / fp32 + residual / bf16 results, and the lm_head product with the cross-entropy statistics in
// its epilogue and the logits kept as bf16.
#include <stdlib.h>
#include "halo_common.h"
#include "halo_internal.h"
#include "gemm_rows.h"

namespace {

using namespace halo_gr;

// columns per tile for an [M x N] result: the choice that needs the fewest MFMA phases per CU, rounds x (TN + 1) over the device's CUs
// (the + 1: a tile's prologue and epilogue); ties go to the wider tile (fewer operand bytes through L2 per product)
int pick_tn(int M, int N) {
    const int cus = halo_cu_count() > 0 ? halo_cu_count() : 256;
    const long tm = (M + 255) / 256;
    int best = 9;
    long best_cost = -1;
    for (int tn = 9; tn >= 3; tn -= 3) {
        const long tiles = tm * ((N + 32 * tn - 1) / (32 * tn));
        const long cost = ((tiles + cus - 1) / cus) * (tn + 1);
        if (best_cost < 0 || cost < best_cost) { best = tn; best_cost = cost; }
    }
    const char *e = getenv("HALO_GEMM_ROWS_TN");
    if (e && (atoi(e) == 3 || atoi(e) == 6 || atoi(e) == 9)) best = atoi(e);
    return best;
}

template <int EPI, bool AIMG>
int launch_tn(int tn, Args &a, hipStream_t st) {
    a.tiles_m = (a.M + 255) / 256;
    a.tiles_n = (a.N + 32 * tn - 1) / (32 * tn);
    if (a.kslices < 1) { a.kslices = 1; a.ktper = a.KT; }
    hipError_t e;
    if (tn == 9) e = launch<9, EPI, AIMG>(a, st);
    else if (tn == 6) e = launch<6, EPI, AIMG>(a, st);
    else e = launch<3, EPI, AIMG>(a, st);
    return e == hipSuccess ? HALO_OK : HALO_ELAUNCH;
}

// C [M][ldc] = slab 0 + slab 1 (+ ...), four columns per thread (N % 4 == 0): the K-slices of a long contraction
__global__ __launch_bounds__(256) void rows_slab_sum_kernel(const float *__restrict__ slab, long slab_stride, int nslab, float *__restrict__ C, long ldc,
                                                            int M, int N4) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)M * N4) return;
    const int r = (int)(idx / N4), c = (int)(idx % N4) * 4;
    f32x4 s = *reinterpret_cast<const f32x4 *>(slab + (long)r * (N4 * 4) + c);
    for (int k = 1; k < nslab; ++k) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(slab + k * slab_stride + (long)r * (N4 * 4) + c);
        s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
    }
    *reinterpret_cast<f32x4 *>(C + (long)r * ldc + c) = s;
}

// loss[row] = logsumexp over the tile columns' (max, sum exp) - target logit; 0 (and lse 0) on ignored rows; one wave per row
__global__ __launch_bounds__(256) void rows_ce_merge_kernel(const float *__restrict__ part, const float *__restrict__ tlogit,
                                                            const int64_t *__restrict__ target, float *__restrict__ loss,
                                                            float *__restrict__ lse_out, int rows, int strips, long ignore_index) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    if (target[row] == ignore_index) {
        if (lane == 0) { loss[row] = 0.f; if (lse_out) lse_out[row] = 0.f; }
        return;
    }
    const float *pr = part + (long)row * strips * 2;
    float m = -INFINITY;
    for (int t = lane; t < strips; t += 64) m = fmaxf(m, pr[2 * t]);
    m = wave_max(m);
    float s = 0.f;
    for (int t = lane; t < strips; t += 64) s += pr[2 * t + 1] * __expf(pr[2 * t] - m);
    s = wave_sum(s);
    if (lane == 0) {
        const float l = m + logf(s);
        loss[row] = l - tlogit[row];
        if (lse_out) lse_out[row] = l;
    }
}

// logits [rows][ld] bf16 <- (softmax - onehot(target)) * grad[row], in place (0 on ignored rows): the lm_head's output gradient as the
// row-major bf16 operand its two gradient products read.  One workgroup per row, 8 columns per thread and pass.
__global__ __launch_bounds__(256) void ce_bwd_bf16_kernel(__bf16 *__restrict__ logits, const int64_t *__restrict__ target, const float *__restrict__ lse,
                                                          const float *__restrict__ grad, long grad_stride, int V, long ld, long ignore_index) {
    const int n = blockIdx.x;
    const long tgt = target[n];
    __bf16 *row = logits + (long)n * ld;
    const bool ign = tgt == ignore_index;
    const float l = ign ? 0.f : lse[n], g = ign ? 0.f : grad[(long)n * grad_stride];
    for (int c8 = threadIdx.x; c8 < V / 8; c8 += 256) {
        bf16x8 v = *reinterpret_cast<const bf16x8 *>(row + 8 * c8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = 8 * c8 + e;
            v[e] = ign ? (__bf16)0.f : (__bf16)((__expf((float)v[e] - l) - (c == tgt ? 1.0f : 0.f)) * g);
        }
        *reinterpret_cast<bf16x8 *>(row + 8 * c8) = v;
    }
}

}  // namespace

extern "C" {

int halo_gemm_rows_supported(int M, int N, int K) {
    return halo_math_mode() == HALO_MATH_BF16 && M > 0 && N > 0 && K > 0 && K % 32 == 0 && N % 8 == 0;
}

static int gemm_rows_impl(const void *a_image, const void *a_bf16, long lda, const void *b_image, int M, int N, int K, float *C, long ldc,
                          const float *residual, long ldr, void *out_bf16, long ldo, int act, void *pre_bf16, halo_stream_t stream) {
    HALO_CHECK_ARG(b_image && M > 0 && N > 0 && K > 0 && (a_image != nullptr) != (a_bf16 != nullptr) && (C != nullptr) != (out_bf16 != nullptr));
    HALO_CHECK_ARG(K % 32 == 0 && N % 8 == 0);
    if (halo_math_mode() != HALO_MATH_BF16) return HALO_ENOTSUP;
    if (a_bf16) HALO_CHECK_ARG(lda >= K && lda % 8 == 0 && (uintptr_t)a_bf16 % 16 == 0);
    if (C) HALO_CHECK_ARG(ldc >= N && ldc % 4 == 0 && (uintptr_t)C % 16 == 0 && (!residual || (ldr >= N && ldr % 4 == 0 && (uintptr_t)residual % 16 == 0)));
    if (out_bf16) HALO_CHECK_ARG(ldo >= N && ldo % 8 == 0 && (uintptr_t)out_bf16 % 16 == 0 && !residual);
    HALO_CHECK_ARG(((uintptr_t)a_image | (uintptr_t)b_image) % 16 == 0);
    Args a = {};
    a.a_img = (const char *)a_image; a.a_rm = (const __bf16 *)a_bf16; a.lda = lda; a.b_img = (const char *)b_image;
    a.M = M; a.N = N; a.KT = K / 32;
    a.C = C; a.ldc = ldc; a.R = residual; a.ldr = ldr; a.O = (__bf16 *)out_bf16; a.ldo = ldo;
    a.act = act; a.O2 = (__bf16 *)pre_bf16;
    int tn = pick_tn(M, N);
    hipStream_t st = (hipStream_t)stream;
    // A very long contraction under few output tiles (the lm_head's input gradient: K = 50304 under 8192 x 768): the narrow tile that fills
    // the chip in one round is paced by its operand stream (22 KB per 384 MFMA cycles and CU), so it runs on tiles twice as wide and the
    // k-blocks in two slices instead -- fp32 slabs in the scratch lent by halo_set_scratch, added by a second launch
    if (C && !residual && !getenv("HALO_GEMM_ROWS_TN") && tn == 3 && K >= 16384 && N % 4 == 0) {
        const int cus = halo_cu_count() > 0 ? halo_cu_count() : 256;
        const long tiles6 = (long)((M + 255) / 256) * ((N + 191) / 192);
        void *scratch; size_t bytes;
        halo_get_scratch(&scratch, &bytes);
        if (2 * tiles6 <= cus && scratch && bytes >= (size_t)2 * M * N * sizeof(float)) {
            a.kslices = 2; a.ktper = (a.KT + 1) / 2;
            a.C = (float *)scratch; a.ldc = N; a.slab_stride = (long)M * N;
            const int rc = a_image ? launch_tn<EPI_F32, true>(6, a, st) : launch_tn<EPI_F32, false>(6, a, st);
            if (rc != HALO_OK) return rc;
            const long n4 = (long)M * (N / 4);
            hipLaunchKernelGGL(rows_slab_sum_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, (const float *)scratch, a.slab_stride, 2, C, ldc, M, N / 4);
            return halo_launch_status();
        }
    }
    if (a_image) {
        if (out_bf16) return launch_tn<EPI_BF16, true>(tn, a, st);
        return residual ? launch_tn<EPI_RESID, true>(tn, a, st) : launch_tn<EPI_F32, true>(tn, a, st);
    }
    if (out_bf16) return launch_tn<EPI_BF16, false>(tn, a, st);
    return residual ? launch_tn<EPI_RESID, false>(tn, a, st) : launch_tn<EPI_F32, false>(tn, a, st);
}

int halo_gemm_rows(const void *a_image, const void *a_bf16, long lda, const void *b_image, int M, int N, int K, float *C, long ldc,
                   const float *residual, long ldr, void *out_bf16, long ldo, halo_stream_t stream) {
    return gemm_rows_impl(a_image, a_bf16, lda, b_image, M, N, K, C, ldc, residual, ldr, out_bf16, ldo, 0, nullptr, stream);
}

int halo_gemm_rows_gelu(const void *a_image, const void *a_bf16, long lda, const void *b_image, int M, int N, int K, void *out_bf16,
                        void *pre_bf16, long ldo, int exact, halo_stream_t stream) {
    HALO_CHECK_ARG(out_bf16 && (!pre_bf16 || (uintptr_t)pre_bf16 % 16 == 0));
    return gemm_rows_impl(a_image, a_bf16, lda, b_image, M, N, K, nullptr, 0, nullptr, 0, out_bf16, ldo, exact ? HALO_GEMM_GELU_ERF : HALO_GEMM_GELU, pre_bf16, stream);
}

static int ce_tn(int M, int N) { return pick_tn(M, N); }       // (the lm_head at V = 50304: 288 columns -- 914 against 1018 us with the logits kept)

size_t halo_gemm_rows_ce_workspace_bytes(int M, int N) {
    if (M <= 0 || N <= 0) return 0;
    return ((size_t)M * ((N + 95) / 96) * 2 + (size_t)M) * sizeof(float);       // (sized for the narrowest tile)
}

int halo_gemm_rows_ce(const void *a_image, const void *a_bf16, long lda, const void *b_image, int M, int N, int K, const int64_t *targets,
                      long ignore_index, void *workspace, float *loss, float *lse, void *logits_bf16, long ldo, halo_stream_t stream) {
    HALO_CHECK_ARG(b_image && targets && workspace && loss && M > 0 && N > 0 && K > 0 && (a_image != nullptr) != (a_bf16 != nullptr));
    HALO_CHECK_ARG(K % 32 == 0 && N % 8 == 0);
    if (halo_math_mode() != HALO_MATH_BF16) return HALO_ENOTSUP;
    if (a_bf16) HALO_CHECK_ARG(lda >= K && lda % 8 == 0 && (uintptr_t)a_bf16 % 16 == 0);
    if (logits_bf16) HALO_CHECK_ARG(ldo >= N && ldo % 8 == 0 && (uintptr_t)logits_bf16 % 16 == 0);
    Args a = {};
    a.a_img = (const char *)a_image; a.a_rm = (const __bf16 *)a_bf16; a.lda = lda; a.b_img = (const char *)b_image;
    a.M = M; a.N = N; a.KT = K / 32;
    a.O = (__bf16 *)logits_bf16; a.ldo = ldo;
    const int tn = ce_tn(M, N);
    const int strips = (N + 32 * tn - 1) / (32 * tn);
    a.ce_target = targets; a.ce_part = (float *)workspace; a.ce_tlogit = a.ce_part + (size_t)M * strips * 2;
    hipStream_t st = (hipStream_t)stream;
    const int rc = a_image ? launch_tn<EPI_CE, true>(tn, a, st) : launch_tn<EPI_CE, false>(tn, a, st);
    if (rc != HALO_OK) return rc;
    hipLaunchKernelGGL(rows_ce_merge_kernel, dim3((M + 3) / 4), dim3(256), 0, st, a.ce_part, a.ce_tlogit, targets, loss, lse, M, strips, ignore_index);
    return halo_launch_status();
}

int halo_cross_entropy_bwd_bf16(void *logits_bf16, const int64_t *targets, const float *lse, const float *grad, long grad_stride, int rows,
                                int V, long ld, long ignore_index, halo_stream_t stream) {
    HALO_CHECK_ARG(logits_bf16 && targets && lse && grad && rows > 0 && V > 0 && V % 8 == 0 && ld >= V && ld % 8 == 0);
    HALO_CHECK_ARG((grad_stride == 0 || grad_stride == 1) && (uintptr_t)logits_bf16 % 16 == 0);
    hipLaunchKernelGGL(ce_bwd_bf16_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, (__bf16 *)logits_bf16, targets, lse, grad, grad_stride, V, ld,
                       ignore_index);
    return halo_launch_status();
}

}  // extern "C"
