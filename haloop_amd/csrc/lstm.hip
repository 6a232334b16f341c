// Multi-layer LSTM forward/backward for gfx950.
//
// Structure (per layer): one exact-f32 MFMA GEMM for the input projection of all T frames, then T
// dependent "step" kernels, each fusing the recurrent GEMM h_{t-1} W_hh^T (v_mfma_f32_16x16x4_f32),
// the gate non-linearities and the state update.  The chain of step launches is what the caller
// captures in a hipGraph: on this chip a dependent kernel boundary (~1.5 us) is cheaper than a
// grid-wide barrier inside a persistent kernel (~4-5 us; MI355X_MICROARCH.md price list, rows
// "boundary" vs "barrier-xcd"), so the time recursion is cut at every step.
//
// Step kernel tiling: one workgroup owns a 16(batch) x 16(hidden unit) tile and all four gates
// of it.  Waves are (gate, k-slice) pairs for the forward, k-slices of the 4H-deep contraction
// for the backward; partial 16x16 accumulators meet in LDS, then 256 threads do the pointwise
// cell update for their (b, j) element.
//
// Operand layout: both MFMA operands are read from MFMA-PACKED images, [tile][k-block][lane][4]:
// the float4 a lane feeds to four consecutive 16x16x4 MFMAs sits at lane*16 bytes of a contiguous
// 1 KiB block, so every wave-level load is one fully coalesced 1 KiB request.  (Reading the
// row-major [B,H]/[4H,H] arrays directly makes each load touch 16 rows 4 KiB apart: same L1 set,
// same L2 channel -- measured 18 us per step instead of ~5.)  W_hh is packed once per call; the
// packed copy of h_t (and of the gate gradients in the backward) is written by the previous step's
// epilogue next to the row-major copy that the batched GEMMs consume.
// Row-major buffers are time-major: h/c [T+1,B,H], gates [T,B,4H].
#include "halo_common.h"
#include "halo_internal.h"

namespace {

enum YMode { Y_NONE = 0, Y_PLAIN = 1, Y_RELU = 2, Y_DROPOUT = 3 };

constexpr int CHUNK = 4;   // k-blocks (of 16) per register stage

__device__ __forceinline__ void load_chunk(f32x4 (&a)[CHUNK], f32x4 (&w)[CHUNK], const float *ap, const float *bp,
                                           int blk0) {
#pragma unroll
    for (int i = 0; i < CHUNK; ++i) {
        a[i] = *reinterpret_cast<const f32x4 *>(ap + (long)(blk0 + i) * 256);
        w[i] = *reinterpret_cast<const f32x4 *>(bp + (long)(blk0 + i) * 256);
    }
}

__device__ __forceinline__ void mma_chunk(const f32x4 (&a)[CHUNK], const f32x4 (&w)[CHUNK], f32x4 &acc0, f32x4 &acc1) {
#pragma unroll
    for (int i = 0; i < CHUNK; i += 2) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][m], w[i][m], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i + 1][m], w[i + 1][m], acc1, 0, 0, 0);
        }
    }
}

// sum over k-blocks [0, nblk) of A-block x B-block; ap/bp point at this lane's float4 of block 0
__device__ __forceinline__ f32x4 packed_dot(const float *ap, const float *bp, int nblk) {
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    if (nblk % (2 * CHUNK) == 0) {
        // two register stages, branch-free so that hipcc can keep the next stage's loads in flight
        // behind a counted vmcnt while the MFMAs of the current stage run
        f32x4 a0[CHUNK], w0[CHUNK], a1[CHUNK], w1[CHUNK];
        load_chunk(a0, w0, ap, bp, 0);
        for (int c = 0; c < nblk; c += 2 * CHUNK) {
            load_chunk(a1, w1, ap, bp, c + CHUNK);
            mma_chunk(a0, w0, acc0, acc1);
            load_chunk(a0, w0, ap, bp, min(c + 2 * CHUNK, nblk - CHUNK));   // last pass: harmless re-read
            mma_chunk(a1, w1, acc0, acc1);
        }
    } else {
        for (int blk = 0; blk < nblk; ++blk) {
            const f32x4 a = *reinterpret_cast<const f32x4 *>(ap + (long)blk * 256);
            const f32x4 w = *reinterpret_cast<const f32x4 *>(bp + (long)blk * 256);
#pragma unroll
            for (int m = 0; m < 4; ++m) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], w[m], acc0, 0, 0, 0);
        }
    }
    return acc0 + acc1;
}

struct StepFwdArgs {
    const float *hp_prev; // packed h_{t-1}  [BT/16][H/16][64][4]
    const float *cprev;   // [B,H]
    const float *wp;      // packed W_hh     [H/16][4][H/16][64][4]
    float *gates;         // [B,4H] in: x W_ih^T + b ; out: activated i,f,g,o
    float *hout;          // [B,H] row-major h_t
    float *hp_out;        // packed h_t
    float *cout;          // [B,H]
    float *y;             // optional second output of h (strided)
    long y_stride_b;
    int y_mode;
    uint64_t drop_base;   // flat index of (t, b=0, j=0) in the time-major dropout tensor
    DropoutCfg drop;
    int B, H;
};

template <int KS>
__global__ __launch_bounds__(256 * KS) void lstm_step_fwd_kernel(const StepFwdArgs p) {
    constexpr int NW = 4 * KS;
    __shared__ float red[NW][256];
    const int jt = blockIdx.x, bt = blockIdx.y;
    const int j0 = jt * 16, b0 = bt * 16;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int gate = wave & 3, ks = wave >> 2;
    const int H = p.H, nkb = H >> 4;
    const int nblk = nkb / KS;

    const float *ap = p.hp_prev + ((long)bt * nkb + ks * nblk) * 256 + lane * 4;
    const float *bp = p.wp + (((long)jt * 4 + gate) * nkb + ks * nblk) * 256 + lane * 4;
    const f32x4 acc = packed_dot(ap, bp, nblk);
    // D layout: col = lane&15 (hidden unit), row = 4*(lane>>4) + reg (batch)
    {
        const int r = lane & 15, q = lane >> 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) red[wave][(4 * q + e) * 16 + r] = acc[e];
    }
    __syncthreads();

    if (threadIdx.x < 256) {
        const int i = threadIdx.x >> 4, j = threadIdx.x & 15;
        const int b = b0 + i;
        float h = 0.f;
        if (b < p.B) {
            float pre[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < KS; ++k) s += red[k * 4 + g][threadIdx.x];
                pre[g] = s + p.gates[(long)b * 4 * H + (long)g * H + j0 + j];
            }
            const float ig = sigmoidf_(pre[0]), fg = sigmoidf_(pre[1]), gg = tanhf(pre[2]), og = sigmoidf_(pre[3]);
            const long e = (long)b * H + j0 + j;
            const float c = fg * p.cprev[e] + ig * gg;
            h = og * tanhf(c);
            float *gp = p.gates + (long)b * 4 * H + j0 + j;
            gp[0] = ig; gp[H] = fg; gp[2 * (long)H] = gg; gp[3 * (long)H] = og;
            p.cout[e] = c;
            p.hout[e] = h;
            if (p.y_mode != Y_NONE) {
                float v = h;
                if (p.y_mode == Y_RELU) v = fmaxf(h, 0.f);
                else if (p.y_mode == Y_DROPOUT) v = h * dropout_mult(p.drop, p.drop_base + (uint64_t)e);
                p.y[(long)b * p.y_stride_b + j0 + j] = v;
            }
        }
        // packed copy for the next step: k-block = this hidden tile, lane = (j>>2)*16 + i, elem j&3
        p.hp_out[((long)bt * nkb + jt) * 256 + ((j >> 2) * 16 + i) * 4 + (j & 3)] = h;
    }
}

struct StepBwdArgs {
    const float *dgp_next; // packed gate gradients of step t+1 [BT/16][4H/16][64][4], or NULL at t = T-1
    const float *wpT;      // packed W_hh^T  [H/16][4H/16][64][4]
    float *gates;          // [B,4H] in: activated gates of step t ; out: gradients w.r.t. pre-activations
    float *dgp_out;        // packed copy of the gate gradients written here
    const float *c;        // [B,H] c_t
    const float *cprev;    // [B,H] c_{t-1}
    float *dc;             // [B,H] carry, in/out
    const float *dy;       // gradient arriving from above for step t (strided), may be NULL
    long dy_stride_b;
    int dy_relu;           // dy is w.r.t. relu(h): mask with h > 0
    const float *dhinit;   // [B,H] extra dh added at this step (dhn at t = T-1), or NULL
    int first;             // t == T-1: dc carry starts from dcinit (or 0)
    const float *dcinit;   // [B,H] or NULL
    int B, H;
};

template <int NW>
__global__ __launch_bounds__(64 * NW) void lstm_step_bwd_kernel(const StepBwdArgs p) {
    __shared__ float red[NW][256];
    const int jt = blockIdx.x, bt = blockIdx.y;
    const int j0 = jt * 16, b0 = bt * 16;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int H = p.H, K = 4 * H, nkb4 = K >> 4;

    if (p.dgp_next) {
        const int nblk = nkb4 / NW;
        const float *ap = p.dgp_next + ((long)bt * nkb4 + wave * nblk) * 256 + lane * 4;
        const float *bp = p.wpT + ((long)jt * nkb4 + wave * nblk) * 256 + lane * 4;
        const f32x4 acc = packed_dot(ap, bp, nblk);
        const int r = lane & 15, q = lane >> 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) red[wave][(4 * q + e) * 16 + r] = acc[e];
    }
    __syncthreads();

    if (threadIdx.x < 256) {
        const int i = threadIdx.x >> 4, j = threadIdx.x & 15;
        const int b = b0 + i;
        float dg[4] = {0.f, 0.f, 0.f, 0.f};
        if (b < p.B) {
            const long e = (long)b * H + j0 + j;
            float dh = 0.f;
            if (p.dgp_next) {
#pragma unroll
                for (int k = 0; k < NW; ++k) dh += red[k][threadIdx.x];
            }
            if (p.dhinit) dh += p.dhinit[e];
            float *gp = p.gates + (long)b * K + j0 + j;
            const float ig = gp[0], fg = gp[H], gg = gp[2 * (long)H], og = gp[3 * (long)H];
            const float c = p.c[e];
            const float tc = tanhf(c);
            if (p.dy) {
                float d = p.dy[(long)b * p.dy_stride_b + j0 + j];
                if (p.dy_relu && !(og * tc > 0.f)) d = 0.f;
                dh += d;
            }
            float dcc = p.first ? (p.dcinit ? p.dcinit[e] : 0.f) : p.dc[e];
            dcc += dh * og * (1.f - tc * tc);
            const float d_o = dh * tc;
            const float d_i = dcc * gg, d_f = dcc * p.cprev[e], d_g = dcc * ig;
            p.dc[e] = dcc * fg;
            dg[0] = d_i * ig * (1.f - ig);
            dg[1] = d_f * fg * (1.f - fg);
            dg[2] = d_g * (1.f - gg * gg);
            dg[3] = d_o * og * (1.f - og);
            gp[0] = dg[0]; gp[H] = dg[1]; gp[2 * (long)H] = dg[2]; gp[3 * (long)H] = dg[3];
        }
        const int nkb = H >> 4;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            p.dgp_out[((long)bt * nkb4 + g * nkb + jt) * 256 + ((j >> 2) * 16 + i) * 4 + (j & 3)] = dg[g];
    }
}

// wp[jt][g][kb][lane][e] = W[g*H + jt*16 + (lane&15)][kb*16 + 4*(lane>>4) + e]
__global__ __launch_bounds__(256) void pack_whh_kernel(const float *__restrict__ w, float *__restrict__ wp, int H) {
    const int nkb = H >> 4;
    const long total = (long)4 * H * H / 4;     // float4 units
    for (long u = blockIdx.x * 256L + threadIdx.x; u < total; u += (long)gridDim.x * 256) {
        const int lane = (int)(u & 63);
        long blk = u >> 6;
        const int kb = (int)(blk % nkb); blk /= nkb;
        const int g = (int)(blk & 3);
        const int jt = (int)(blk >> 2);
        const long row = (long)g * H + jt * 16 + (lane & 15);
        const int k = kb * 16 + 4 * (lane >> 4);
        *reinterpret_cast<f32x4 *>(wp + u * 4) = *reinterpret_cast<const f32x4 *>(w + row * H + k);
    }
}

// wpT[jt][kb][lane][e] = W[kb*16 + 4*(lane>>4) + e][jt*16 + (lane&15)],  kb over the 4H rows of W
__global__ __launch_bounds__(256) void pack_whhT_kernel(const float *__restrict__ w, float *__restrict__ wpT, int H) {
    const int nkb4 = (4 * H) >> 4;
    const long total = (long)4 * H * H / 4;
    for (long u = blockIdx.x * 256L + threadIdx.x; u < total; u += (long)gridDim.x * 256) {
        const int lane = (int)(u & 63);
        long blk = u >> 6;
        const int kb = (int)(blk % nkb4);
        const int jt = (int)(blk / nkb4);
        const long row = (long)kb * 16 + 4 * (lane >> 4);
        const int col = jt * 16 + (lane & 15);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = w[(row + e) * H + col];
        *reinterpret_cast<f32x4 *>(wpT + u * 4) = v;
    }
}

// packed[bt][kb][lane][e] = x[bt*16 + (lane&15)][kb*16 + 4*(lane>>4) + e]  (rows >= B -> 0); x NULL -> zeros
__global__ __launch_bounds__(256) void pack_rows_kernel(const float *__restrict__ x, float *__restrict__ xp, int B, int W) {
    const int nkb = W >> 4;
    const int BT = (B + 15) / 16;
    const long total = (long)BT * 16 * W / 4;
    for (long u = blockIdx.x * 256L + threadIdx.x; u < total; u += (long)gridDim.x * 256) {
        const int lane = (int)(u & 63);
        const long blk = u >> 6;
        const int kb = (int)(blk % nkb);
        const int bt = (int)(blk / nkb);
        const int b = bt * 16 + (lane & 15);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (x && b < B) v = *reinterpret_cast<const f32x4 *>(x + (long)b * W + kb * 16 + 4 * (lane >> 4));
        *reinterpret_cast<f32x4 *>(xp + u * 4) = v;
    }
}

struct LayerBufs {
    float *h, *c, *gates, *ydrop, *hp;
};

inline size_t bt16(int B) { return (size_t)((B + 15) / 16) * 16; }

inline size_t layer_floats(int T, int B, int H) {
    return (size_t)(2 * (T + 1) + 5 * T) * B * H + (size_t)(T + 1) * bt16(B) * H;
}

// reserve = [ packed W_hh scratch (4H*H) | layer 0 | layer 1 | ... | tiled images for the input projection ]
inline LayerBufs layer_bufs(float *reserve, int l, int T, int B, int H) {
    float *base = reserve + (size_t)4 * H * H + (size_t)l * layer_floats(T, B, H);
    LayerBufs lb;
    lb.h = base;
    lb.c = lb.h + (size_t)(T + 1) * B * H;
    lb.gates = lb.c + (size_t)(T + 1) * B * H;
    lb.ydrop = lb.gates + (size_t)T * B * 4 * H;
    lb.hp = lb.ydrop + (size_t)T * B * H;
    return lb;
}

inline int pick_fwd_ks(int H) { return (H % 64 == 0) ? 4 : (H % 32 == 0) ? 2 : 1; }
inline int pick_bwd_nw(int H) { return (H % 64 == 0) ? 16 : (H % 32 == 0) ? 8 : 4; }

int launch_step_fwd(const StepFwdArgs &a, hipStream_t st) {
    dim3 grid(a.H / 16, (a.B + 15) / 16);
    switch (pick_fwd_ks(a.H)) {
        case 4: hipLaunchKernelGGL(lstm_step_fwd_kernel<4>, grid, dim3(1024), 0, st, a); break;
        case 2: hipLaunchKernelGGL(lstm_step_fwd_kernel<2>, grid, dim3(512), 0, st, a); break;
        default: hipLaunchKernelGGL(lstm_step_fwd_kernel<1>, grid, dim3(256), 0, st, a); break;
    }
    return halo_launch_status();
}

int launch_step_bwd(const StepBwdArgs &a, hipStream_t st) {
    dim3 grid(a.H / 16, (a.B + 15) / 16);
    switch (pick_bwd_nw(a.H)) {
        case 16: hipLaunchKernelGGL(lstm_step_bwd_kernel<16>, grid, dim3(1024), 0, st, a); break;
        case 8: hipLaunchKernelGGL(lstm_step_bwd_kernel<8>, grid, dim3(512), 0, st, a); break;
        default: hipLaunchKernelGGL(lstm_step_bwd_kernel<4>, grid, dim3(256), 0, st, a); break;
    }
    return halo_launch_status();
}

inline unsigned pack_grid(size_t units) {
    size_t g = (units + 255) / 256;
    return (unsigned)(g > 2048 ? 2048 : (g < 1 ? 1 : g));
}

#define HALO_TRY(expr)            \
    do {                          \
        int rc_ = (expr);         \
        if (rc_ != HALO_OK) return rc_; \
    } while (0)

inline int copy_d2d(float *dst, const float *src, size_t n, hipStream_t st) {
    return hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, st) == hipSuccess ? HALO_OK : HALO_ELAUNCH;
}

}  // namespace

extern "C" {

size_t halo_lstm_reserve_bytes(int T, int B, int in0, int H, int L) {
    if (T <= 0 || B <= 0 || in0 <= 0 || H <= 0 || L <= 0) return 0;
    const int kin = in0 > H ? in0 : H;
    return ((size_t)4 * H * H + (size_t)L * layer_floats(T, B, H)) * sizeof(float) +
           halo_tiled_image_bytes(T * B, kin) + halo_tiled_image_bytes(4 * H, kin);
}

size_t halo_lstm_bwd_workspace_bytes(int T, int B, int in0, int H, int L) {
    if (T <= 0 || B <= 0 || H <= 0 || L <= 0) return 0;
        // packed W_hh^T [H,4H] + dc carry [B,H] + gradient w.r.t. a layer's input [T,B,H]
    // + two packed gate-gradient images [BT16, 4H] + tiled images for the batched gradient GEMMs
    const int kin = in0 > H ? in0 : H;
    return ((size_t)H * 4 * H + (size_t)B * H + (size_t)T * B * H + 2 * bt16(B) * 4 * H) * sizeof(float) +
           halo_tiled_image_bytes(4 * H, T * B) + halo_tiled_image_bytes(T * B, 4 * H) +
           2 * halo_tiled_image_bytes(kin, T * B) + halo_tiled_image_bytes(kin, 4 * H);
}

int halo_lstm_fwd(const float *x, const float *const *w_ih, const float *const *w_hh, const float *const *b_ih,
                  const float *const *b_hh, const float *h0, const float *c0, float *y, long y_stride_t,
                  long y_stride_b, int y_relu, float *hn, float *cn, float *reserve, int T, int B, int in0, int H,
                  int L, float p_drop, uint64_t seed, uint32_t offset, const uint32_t *offset_dev,
                  halo_stream_t stream) {
    HALO_CHECK_ARG(x && w_ih && w_hh && b_ih && b_hh && reserve);
    HALO_CHECK_ARG(T > 0 && B > 0 && in0 > 0 && H > 0 && L > 0);
    if (H % 16 != 0) return HALO_ENOTSUP;
    hipStream_t st = (hipStream_t)stream;
    const size_t BH = (size_t)B * H, PH = bt16(B) * H;
    float *wp = reserve;
    const int kin = in0 > H ? in0 : H;
    char *img_in = (char *)(reserve + (size_t)4 * H * H + (size_t)L * layer_floats(T, B, H));
    char *img_w = img_in + halo_tiled_image_bytes(T * B, kin);
    for (int l = 0; l < L; ++l) {
        HALO_CHECK_ARG(w_ih[l] && w_hh[l] && b_ih[l] && b_hh[l]);
        const LayerBufs lb = layer_bufs(reserve, l, T, B, H);
        const bool last = (l == L - 1);
        const bool drop_out = !last && p_drop > 0.f;
        const float *in;
        int in_dim;
        if (l == 0) { in = x; in_dim = in0; }
        else {
            const LayerBufs pb = layer_bufs(reserve, l - 1, T, B, H);
            in = p_drop > 0.f ? pb.ydrop : pb.h + BH;
            in_dim = H;
        }
        // gates[T*B, 4H] = in[T*B, in_dim] * W_ih^T + b_ih + b_hh
        if (halo_math_mode() == HALO_MATH_BF16X3 && in_dim >= 64) {
            HALO_TRY(halo_prep_tiles(in, T * B, in_dim, in_dim, 0, img_in, st));
            HALO_TRY(halo_prep_tiles(w_ih[l], 4 * H, in_dim, in_dim, 0, img_w, st));
            HALO_TRY(halo_gemm_bf16x3_tiled(img_in, img_w, T * B, 4 * H, in_dim, lb.gates, 4 * H, b_ih[l], b_hh[l], 0,
                                            nullptr, st));
        } else {
            HALO_TRY(halo_gemm_f32(1, 1, T * B, 4 * H, in_dim, in, in_dim, w_ih[l], in_dim, lb.gates, 4 * H, b_ih[l],
                                   b_hh[l], 0, 0.f, 0, 0, 0, nullptr, stream));
        }
        hipLaunchKernelGGL(pack_whh_kernel, dim3(pack_grid((size_t)H * H)), dim3(256), 0, st, w_hh[l], wp, H);
        HALO_TRY(halo_launch_status());
        const float *h0l = h0 ? h0 + (size_t)l * BH : nullptr;
        if (h0l) HALO_TRY(copy_d2d(lb.h, h0l, BH, st));
        else HALO_TRY(halo_fill(lb.h, BH, 0.f, st));
        hipLaunchKernelGGL(pack_rows_kernel, dim3(pack_grid(PH / 4)), dim3(256), 0, st, h0l, lb.hp, B, H);
        HALO_TRY(halo_launch_status());
        if (c0) HALO_TRY(copy_d2d(lb.c, c0 + (size_t)l * BH, BH, st));
        else HALO_TRY(halo_fill(lb.c, BH, 0.f, st));
        const DropoutCfg dc = make_dropout(drop_out ? p_drop : 0.f, seed, HALO_STREAM_LSTM_LAYER0 + (uint32_t)l, offset, offset_dev);
        for (int t = 0; t < T; ++t) {
            StepFwdArgs a;
            a.hp_prev = lb.hp + (size_t)t * PH;
            a.cprev = lb.c + (size_t)t * BH;
            a.wp = wp;
            a.gates = lb.gates + (size_t)t * B * 4 * H;
            a.hout = lb.h + (size_t)(t + 1) * BH;
            a.hp_out = lb.hp + (size_t)(t + 1) * PH;
            a.cout = lb.c + (size_t)(t + 1) * BH;
            a.B = B; a.H = H;
            a.drop = dc;
            a.drop_base = (uint64_t)t * BH;
            if (last && y) {
                a.y = y + (long)t * y_stride_t;
                a.y_stride_b = y_stride_b;
                a.y_mode = y_relu ? Y_RELU : Y_PLAIN;
            } else if (drop_out) {
                a.y = lb.ydrop + (size_t)t * BH;
                a.y_stride_b = H;
                a.y_mode = Y_DROPOUT;
            } else {
                a.y = nullptr; a.y_stride_b = 0; a.y_mode = Y_NONE;
            }
            HALO_TRY(launch_step_fwd(a, st));
        }
        if (hn) HALO_TRY(copy_d2d(hn + (size_t)l * BH, lb.h + (size_t)T * BH, BH, st));
        if (cn) HALO_TRY(copy_d2d(cn + (size_t)l * BH, lb.c + (size_t)T * BH, BH, st));
    }
    return HALO_OK;
}

int halo_lstm_bwd(const float *x, const float *const *w_ih, const float *const *w_hh, const float *dy,
                  long y_stride_t, long y_stride_b, int y_relu, const float *dhn, const float *dcn, float *reserve,
                  float *workspace, float *dx, float *const *dw_ih, float *const *dw_hh, float *const *db_ih,
                  float *const *db_hh, int T, int B, int in0, int H, int L, float p_drop, uint64_t seed,
                  uint32_t offset, const uint32_t *offset_dev, halo_stream_t stream) {
    HALO_CHECK_ARG(x && w_ih && w_hh && reserve && workspace && dw_ih && dw_hh && db_ih && db_hh);
    HALO_CHECK_ARG(dy || dhn || dcn);
    HALO_CHECK_ARG(T > 0 && B > 0 && in0 > 0 && H > 0 && L > 0);
    if (H % 16 != 0) return HALO_ENOTSUP;
    hipStream_t st = (hipStream_t)stream;
    const size_t BH = (size_t)B * H, PG = bt16(B) * 4 * H;
    float *wpT = workspace;
    float *dcarry = wpT + (size_t)H * 4 * H;
    float *din = dcarry + BH;          // [T,B,H] gradient w.r.t. the current layer's input
    float *dgp = din + (size_t)T * BH; // two packed gate-gradient images, ping-pong
    const int kin = in0 > H ? in0 : H;
    char *img_gT = (char *)(dgp + 2 * PG);
    char *img_g = img_gT + halo_tiled_image_bytes(4 * H, T * B);
    char *img_hT = img_g + halo_tiled_image_bytes(T * B, 4 * H);
    char *img_inT = img_hT + halo_tiled_image_bytes(kin, T * B);
    char *img_wT = img_inT + halo_tiled_image_bytes(kin, T * B);
    for (int l = L - 1; l >= 0; --l) {
        HALO_CHECK_ARG(w_ih[l] && w_hh[l] && dw_ih[l] && dw_hh[l] && db_ih[l] && db_hh[l]);
        const LayerBufs lb = layer_bufs(reserve, l, T, B, H);
        const bool last = (l == L - 1);
        hipLaunchKernelGGL(pack_whhT_kernel, dim3(pack_grid((size_t)H * H)), dim3(256), 0, st, w_hh[l], wpT, H);
        HALO_TRY(halo_launch_status());
        for (int t = T - 1; t >= 0; --t) {
            StepBwdArgs a;
            a.dgp_next = (t == T - 1) ? nullptr : dgp + (size_t)((t + 1) & 1) * PG;
            a.dgp_out = dgp + (size_t)(t & 1) * PG;
            a.wpT = wpT;
            a.gates = lb.gates + (size_t)t * B * 4 * H;
            a.c = lb.c + (size_t)(t + 1) * BH;
            a.cprev = lb.c + (size_t)t * BH;
            a.dc = dcarry;
            if (last) {
                a.dy = dy ? dy + (long)t * y_stride_t : nullptr;
                a.dy_stride_b = y_stride_b;
                a.dy_relu = y_relu;
            } else {
                a.dy = din + (size_t)t * BH;
                a.dy_stride_b = H;
                a.dy_relu = 0;
            }
            a.first = (t == T - 1);
            a.dhinit = (t == T - 1 && dhn) ? dhn + (size_t)l * BH : nullptr;
            a.dcinit = (t == T - 1 && dcn) ? dcn + (size_t)l * BH : nullptr;
            a.B = B; a.H = H;
            HALO_TRY(launch_step_bwd(a, st));
        }
        // parameter gradients over all frames at once
        const float *in;
        int in_dim;
        if (l == 0) { in = x; in_dim = in0; }
        else {
            const LayerBufs pb = layer_bufs(reserve, l - 1, T, B, H);
            in = p_drop > 0.f ? pb.ydrop : pb.h + BH;
            in_dim = H;
        }
        const bool need_din = (l > 0) || dx;
        float *din_out = l > 0 ? din : dx;
        const DropoutCfg ddrop = make_dropout(l > 0 ? p_drop : 0.f, seed, HALO_STREAM_LSTM_LAYER0 + (uint32_t)(l > 0 ? l - 1 : 0),
                                              offset, offset_dev);
        if (halo_math_mode() == HALO_MATH_BF16X3 && in_dim >= 64) {
            // every operand is split/tiled once; transposes are absorbed by the prep pass
            HALO_TRY(halo_prep_tiles(lb.gates, 4 * H, T * B, 4 * H, 1, img_gT, st));     // dG^T  [4H][TB]
            HALO_TRY(halo_prep_tiles(lb.h, H, T * B, H, 1, img_hT, st));                 // h_prev^T [H][TB]
            HALO_TRY(halo_gemm_bf16x3_tiled(img_gT, img_hT, 4 * H, H, T * B, dw_hh[l], H, nullptr, nullptr, 0, nullptr, st));
            HALO_TRY(halo_prep_tiles(in, in_dim, T * B, in_dim, 1, img_inT, st));        // in^T [in][TB]
            HALO_TRY(halo_gemm_bf16x3_tiled(img_gT, img_inT, 4 * H, in_dim, T * B, dw_ih[l], in_dim, nullptr, nullptr, 0,
                                            nullptr, st));
            if (need_din) {
                HALO_TRY(halo_prep_tiles(lb.gates, T * B, 4 * H, 4 * H, 0, img_g, st));           // dG [TB][4H]
                HALO_TRY(halo_prep_tiles(w_ih[l], in_dim, 4 * H, in_dim, 1, img_wT, st));         // W_ih^T [in][4H]
                HALO_TRY(halo_gemm_bf16x3_tiled(img_g, img_wT, T * B, in_dim, 4 * H, din_out, in_dim, nullptr, nullptr, 0,
                                                &ddrop, st));
            }
        } else {
            // dW_hh[4H,H] = dG[T*B,4H]^T * h_{t-1}[T*B,H]   (h buffer rows 0..T-1 are the previous states)
            HALO_TRY(halo_gemm_f32(0, 0, 4 * H, H, T * B, lb.gates, 4 * H, lb.h, H, dw_hh[l], H, nullptr, nullptr, 0, 0.f,
                                   0, 0, 0, nullptr, stream));
            // dW_ih[4H,in] = dG^T * in
            HALO_TRY(halo_gemm_f32(0, 0, 4 * H, in_dim, T * B, lb.gates, 4 * H, in, in_dim, dw_ih[l], in_dim, nullptr,
                                   nullptr, 0, 0.f, 0, 0, 0, nullptr, stream));
            // gradient w.r.t. the layer's input; for l > 0 with layer l-1's dropout mask folded in
            if (need_din)
                HALO_TRY(halo_gemm_f32(1, 0, T * B, in_dim, 4 * H, lb.gates, 4 * H, w_ih[l], in_dim, din_out, in_dim,
                                       nullptr, nullptr, 0, l > 0 ? p_drop : 0.f, seed,
                                       HALO_STREAM_LSTM_LAYER0 + (uint32_t)(l > 0 ? l - 1 : 0), offset, offset_dev, stream));
        }
        HALO_TRY(halo_colsum(lb.gates, T * B, 4 * H, 4 * H, db_ih[l], stream));
        HALO_TRY(copy_d2d(db_hh[l], db_ih[l], (size_t)4 * H, st));
    }
    return HALO_OK;
}

}  // extern "C"
