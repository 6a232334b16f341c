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
#include <stdlib.h>
#include <string.h>
#include "halo_common.h"
#include "halo_internal.h"
#include "tiled_image.h"
#include "lstm_persist.h"

namespace {

enum YMode { Y_NONE = 0, Y_PLAIN = 1, Y_RELU = 2, Y_DROPOUT = 3 };

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// ---- packed operand images --------------------------------------------------------------------
// F32 format : k-block = 16,  block = [64 lanes][4 f32]            = 1 KiB, feeds 4 x mfma_f32_16x16x4_f32
// X3  format : k-block = 32,  block = [hi|lo][64 lanes][8 bf16]    = 2 KiB, feeds 3 x mfma_f32_16x16x32_bf16
//              (x ~ hi + lo, product = lo*hi' + hi*lo' + hi*hi', fp32 accumulate: see gemm_bf16x3.hip)
// Lane l of a block holds row (l & 15) of the 16-row tile and the k-group (l >> 4).
template <bool X3>
struct Packed {
    static constexpr int KB = X3 ? 32 : 16;            // k per block
    static constexpr int BLOCK_BYTES = X3 ? 2048 : 1024;
    // store one element (tile row i, absolute k) of tile `tile` whose K extent is nk
    __device__ static __forceinline__ void store(void *img, int tile, int nk, int i, int k, float v) {
        const int blk = k / KB, kk = k % KB;
        char *base = (char *)img + ((long)tile * (nk / KB) + blk) * BLOCK_BYTES;
        if (X3) {
            const int lane = (kk >> 3) * 16 + i, e = kk & 7;
            const __bf16 hi = (__bf16)v;
            const __bf16 lo = (__bf16)(v - (float)hi);
            reinterpret_cast<__bf16 *>(base)[lane * 8 + e] = hi;
            reinterpret_cast<__bf16 *>(base + 1024)[lane * 8 + e] = lo;
        } else {
            const int lane = (kk >> 2) * 16 + i, e = kk & 3;
            reinterpret_cast<float *>(base)[lane * 4 + e] = v;
        }
    }
};

constexpr int CHUNK = 4;    // F32: k-blocks per register stage
constexpr int CHUNK3 = 2;   // X3:  k-blocks per register stage (4 x 16 B per lane and block)

__device__ __forceinline__ void load_chunk(f32x4 (&a)[CHUNK], f32x4 (&w)[CHUNK], const char *ap, const char *bp, int blk0) {
#pragma unroll
    for (int i = 0; i < CHUNK; ++i) {
        a[i] = *reinterpret_cast<const f32x4 *>(ap + (long)(blk0 + i) * 1024);
        w[i] = *reinterpret_cast<const f32x4 *>(bp + (long)(blk0 + i) * 1024);
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

struct Frag3 {
    bf16x8 ah, al, wh, wl;
};

// ONE: single-pass bf16 (HALO_MATH_BF16): only the hi halves of the packed blocks are fetched and multiplied
template <bool ONE>
__device__ __forceinline__ void load_chunk3(Frag3 (&f)[CHUNK3], const char *ap, const char *bp, int blk0) {
#pragma unroll
    for (int i = 0; i < CHUNK3; ++i) {
        const char *a = ap + (long)(blk0 + i) * 2048, *w = bp + (long)(blk0 + i) * 2048;
        f[i].ah = *reinterpret_cast<const bf16x8 *>(a);
        f[i].wh = *reinterpret_cast<const bf16x8 *>(w);
        if (!ONE) {
            f[i].al = *reinterpret_cast<const bf16x8 *>(a + 1024);
            f[i].wl = *reinterpret_cast<const bf16x8 *>(w + 1024);
        }
    }
}

template <bool ONE>
__device__ __forceinline__ void mma_chunk3(const Frag3 (&f)[CHUNK3], f32x4 &acc0, f32x4 &acc1) {
    if (!ONE) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[0].al, f[0].wh, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[1].al, f[1].wh, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[0].ah, f[0].wl, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[1].ah, f[1].wl, acc1, 0, 0, 0);
    }
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[0].ah, f[0].wh, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[1].ah, f[1].wh, acc1, 0, 0, 0);
}

// sum over k-blocks [0, nblk) of A-block x B-block; ap/bp point at this lane's 16 bytes of block 0
template <bool X3, bool ONE = false>
__device__ __forceinline__ f32x4 packed_dot(const char *ap, const char *bp, int nblk) {
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    if (X3) {
        if (nblk % (2 * CHUNK3) == 0) {
            Frag3 f0[CHUNK3], f1[CHUNK3];
            load_chunk3<ONE>(f0, ap, bp, 0);
            for (int c = 0; c < nblk; c += 2 * CHUNK3) {
                load_chunk3<ONE>(f1, ap, bp, c + CHUNK3);
                mma_chunk3<ONE>(f0, acc0, acc1);
                load_chunk3<ONE>(f0, ap, bp, min(c + 2 * CHUNK3, nblk - CHUNK3));   // last pass: harmless re-read
                mma_chunk3<ONE>(f1, acc0, acc1);
            }
        } else {
            for (int blk = 0; blk < nblk; ++blk) {
                const char *a = ap + (long)blk * 2048, *w = bp + (long)blk * 2048;
                const bf16x8 ah = *reinterpret_cast<const bf16x8 *>(a), wh = *reinterpret_cast<const bf16x8 *>(w);
                if (!ONE) {
                    const bf16x8 al = *reinterpret_cast<const bf16x8 *>(a + 1024), wl = *reinterpret_cast<const bf16x8 *>(w + 1024);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, wh, acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wl, acc0, 0, 0, 0);
                }
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wh, acc0, 0, 0, 0);
            }
        }
    } else {
        if (nblk % (2 * CHUNK) == 0) {
            // two register stages, branch-free so that hipcc can keep the next stage's loads in flight
            // behind a counted vmcnt while the MFMAs of the current stage run
            f32x4 a0[CHUNK], w0[CHUNK], a1[CHUNK], w1[CHUNK];
            load_chunk(a0, w0, ap, bp, 0);
            for (int c = 0; c < nblk; c += 2 * CHUNK) {
                load_chunk(a1, w1, ap, bp, c + CHUNK);
                mma_chunk(a0, w0, acc0, acc1);
                load_chunk(a0, w0, ap, bp, min(c + 2 * CHUNK, nblk - CHUNK));
                mma_chunk(a1, w1, acc0, acc1);
            }
        } else {
            for (int blk = 0; blk < nblk; ++blk) {
                const f32x4 a = *reinterpret_cast<const f32x4 *>(ap + (long)blk * 1024);
                const f32x4 w = *reinterpret_cast<const f32x4 *>(bp + (long)blk * 1024);
#pragma unroll
                for (int m = 0; m < 4; ++m) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], w[m], acc0, 0, 0, 0);
            }
        }
    }
    return acc0 + acc1;
}

// ---- forward variant with the h_{t-1} tile staged once in LDS ---------------------------------
// The four gate waves of a k-slice read the same A blocks; staging the 16 x H tile (64 KiB at H=1024)
// with global_load_lds and reading fragments with ds_read_b128 takes three quarters of the A traffic
// off the vector-memory path, which is what bounds this kernel (512 KiB per CU per step otherwise).
typedef __attribute__((address_space(3))) const char lds_cchar;

constexpr int WCH = 4;   // W k-blocks per register stage in the LDS-A variant

struct ATileStage {      // what to copy into LDS before the first fragment read
    const char *src;     // this batch tile's packed blocks (global, contiguous)
    char *dst;           // LDS
    int pieces;          // 1 KiB pieces
    int wave, nwaves, lane;
    int hi_only;         // single-pass bf16: skip the lo half (odd 1 KiB pieces) of every 2 KiB block
    __device__ __forceinline__ void issue() const {
        for (int pc = wave; pc < pieces; pc += nwaves) {
            if (hi_only && (pc & 1)) continue;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (long)pc * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void *)(dst + pc * 1024), 16, 0, 0);
        }
        asm volatile("" ::: "memory");        // nothing below may be issued ahead of the LDS-DMA requests
    }
    // wait for the tile while the NEWER loads (the first W stage, N of them per lane) stay in flight:
    // vmcnt retires in issue order, so "all but the N youngest" covers exactly the LDS-DMA requests
    template <int N>
    __device__ __forceinline__ void wait_keeping() const {
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
    }
};

template <bool X3, bool ONE = false>
__device__ __forceinline__ f32x4 dot_lds_a(const ATileStage &stage, lds_cchar *a_lds, const char *bp, int nblk) {
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    if (X3) {
        bf16x8 wh0[WCH], wl0[WCH], wh1[WCH], wl1[WCH];
        auto loadw = [&](bf16x8 (&wh)[WCH], bf16x8 (&wl)[WCH], int blk0) {
#pragma unroll
            for (int i = 0; i < WCH; ++i) {
                wh[i] = *reinterpret_cast<const bf16x8 *>(bp + (long)(blk0 + i) * 2048);
                if (!ONE) wl[i] = *reinterpret_cast<const bf16x8 *>(bp + (long)(blk0 + i) * 2048 + 1024);
            }
        };
        auto mma = [&](const bf16x8 (&wh)[WCH], const bf16x8 (&wl)[WCH], int blk0) {
#pragma unroll
            for (int i = 0; i < WCH; ++i) {
                typedef __attribute__((address_space(3))) const bf16x8 lds_frag;
                const bf16x8 ah = *reinterpret_cast<lds_frag *>(a_lds + (blk0 + i) * 2048);
                f32x4 &acc = (i & 1) ? acc1 : acc0;
                if (!ONE) {
                    const bf16x8 al = *reinterpret_cast<lds_frag *>(a_lds + (blk0 + i) * 2048 + 1024);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, wh[i], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wl[i], acc, 0, 0, 0);
                }
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wh[i], acc, 0, 0, 0);
            }
        };
        stage.issue();
        loadw(wh0, wl0, 0);
        asm volatile("" ::: "memory");
        stage.wait_keeping<(ONE ? 1 : 2) * WCH>();
        for (int c = 0; c < nblk; c += 2 * WCH) {
            loadw(wh1, wl1, c + WCH);
            mma(wh0, wl0, c);
            loadw(wh0, wl0, min(c + 2 * WCH, nblk - WCH));
            mma(wh1, wl1, c + WCH);
        }
    } else {
        f32x4 w0[WCH], w1[WCH];
        auto loadw = [&](f32x4 (&w)[WCH], int blk0) {
#pragma unroll
            for (int i = 0; i < WCH; ++i) w[i] = *reinterpret_cast<const f32x4 *>(bp + (long)(blk0 + i) * 1024);
        };
        auto mma = [&](const f32x4 (&w)[WCH], int blk0) {
#pragma unroll
            for (int i = 0; i < WCH; ++i) {
                typedef __attribute__((address_space(3))) const f32x4 lds_frag;
                const f32x4 a = *reinterpret_cast<lds_frag *>(a_lds + (blk0 + i) * 1024);
                f32x4 &acc = (i & 1) ? acc1 : acc0;
#pragma unroll
                for (int m = 0; m < 4; ++m) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], w[i][m], acc, 0, 0, 0);
            }
        };
        stage.issue();
        loadw(w0, 0);
        asm volatile("" ::: "memory");
        stage.wait_keeping<WCH>();
        for (int c = 0; c < nblk; c += 2 * WCH) {
            loadw(w1, c + WCH);
            mma(w0, c);
            loadw(w0, min(c + 2 * WCH, nblk - WCH));
            mma(w1, c + WCH);
        }
    }
    return acc0 + acc1;
}

struct StepFwdArgs {
    const void *hp_prev;  // packed h_{t-1}  [BT/16][H/KB] blocks
    const float *cprev;   // [B,H]
    const void *wp;       // packed W_hh     [H/16][4][H/KB] blocks
    float *gates;         // [B,4H] in: x W_ih^T + b ; out: activated i,f,g,o
    float *hout;          // [B,H] row-major h_t
    void *hp_out;         // packed h_t
    float *cout;          // [B,H]
    float *y;             // optional second output of h (strided)
    long y_stride_b;
    int y_mode;
    uint64_t drop_base;   // flat index of (t, b=0, j=0) in the time-major dropout tensor
    DropoutCfg drop;
    int B, H;
};

template <int KS, bool X3, bool ALDS, bool ONE = false>
__global__ __launch_bounds__(256 * KS) void lstm_step_fwd_kernel(const StepFwdArgs p) {
    constexpr int NW = 4 * KS;
    using PK = Packed<X3>;
    extern __shared__ __attribute__((aligned(16))) char dyn_lds[];    // [red: NW*256 f32][A tile: (H/KB) blocks]
    float (*red)[256] = reinterpret_cast<float (*)[256]>(dyn_lds);
    const int jt = blockIdx.x, bt = blockIdx.y;
    const int j0 = jt * 16, b0 = bt * 16;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int gate = wave & 3, ks = wave >> 2;
    const int H = p.H, nkb = H / PK::KB;
    const int nblk = nkb / KS;

    // the cell update's own operands are requested first so that they arrive under the MFMA loop
    const int i = threadIdx.x >> 4, j = threadIdx.x & 15;
    const int b = b0 + i;
    const bool cell = threadIdx.x < 256 && b < p.B;
    float gin[4] = {0.f, 0.f, 0.f, 0.f}, cprev = 0.f;
    if (cell) {
#pragma unroll
        for (int g = 0; g < 4; ++g) gin[g] = p.gates[(long)b * 4 * H + (long)g * H + j0 + j];
        cprev = p.cprev[(long)b * H + j0 + j];
    }

    const char *bp = (const char *)p.wp + (((long)jt * 4 + gate) * nkb + ks * nblk) * PK::BLOCK_BYTES + lane * 16;
    f32x4 acc;
    if (ALDS) {
        // stage this batch tile's packed h_{t-1} (nkb blocks, contiguous) into LDS, 1 KiB per wave instruction
        char *a_tile = dyn_lds + NW * 256 * sizeof(float);
        const char *a_src = (const char *)p.hp_prev + (long)bt * nkb * PK::BLOCK_BYTES;
        const ATileStage stage = {a_src, a_tile, nkb * PK::BLOCK_BYTES / 1024, __builtin_amdgcn_readfirstlane(wave), NW, lane, ONE};
        lds_cchar *a_lds = (lds_cchar *)(a_tile + (long)ks * nblk * PK::BLOCK_BYTES + lane * 16);
        acc = dot_lds_a<X3, ONE>(stage, a_lds, bp, nblk);
    } else {
        const char *ap = (const char *)p.hp_prev + ((long)bt * nkb + ks * nblk) * PK::BLOCK_BYTES + lane * 16;
        acc = packed_dot<X3, ONE>(ap, bp, nblk);
    }
    // D layout: col = lane&15 (hidden unit), row = 4*(lane>>4) + reg (batch)
    {
        const int r = lane & 15, q = lane >> 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) red[wave][(4 * q + e) * 16 + r] = acc[e];
    }
    __syncthreads();

    if (threadIdx.x < 256) {
        float h = 0.f;
        if (cell) {
            float pre[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < KS; ++k) s += red[k * 4 + g][threadIdx.x];
                pre[g] = s + gin[g];
            }
            const float ig = sigmoidf_(pre[0]), fg = sigmoidf_(pre[1]), gg = tanhf(pre[2]), og = sigmoidf_(pre[3]);
            const long e = (long)b * H + j0 + j;
            const float c = fg * cprev + ig * gg;
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
        PK::store(p.hp_out, bt, H, i, j0 + j, h);      // packed copy for the next step (rows >= B: zeros)
    }
}

struct StepBwdArgs {
    const void *dgp_next;  // packed gate gradients of step t+1 [BT/16][4H/KB] blocks, or NULL at t = T-1
    const void *wpT;       // packed W_hh^T  [H/16][4H/KB] blocks
    float *gates;          // [B,4H] in: activated gates of step t ; out: gradients w.r.t. pre-activations
    void *dgp_out;         // packed copy of the gate gradients written here
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

template <int NW, bool X3, bool ONE = false>
__global__ __launch_bounds__(64 * NW) void lstm_step_bwd_kernel(const StepBwdArgs p) {
    using PK = Packed<X3>;
    __shared__ float red[NW][256];
    const int jt = blockIdx.x, bt = blockIdx.y;
    const int j0 = jt * 16, b0 = bt * 16;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int H = p.H, K = 4 * H, nkb4 = K / PK::KB;

    const int i = threadIdx.x >> 4, j = threadIdx.x & 15;
    const int b = b0 + i;
    const bool cell = threadIdx.x < 256 && b < p.B;
    const long e = (long)b * H + j0 + j;
    float gv[4] = {0.f, 0.f, 0.f, 0.f}, c = 0.f, cprev = 0.f, dyv = 0.f, dcin = 0.f, dh0 = 0.f;
    if (cell) {     // requested before the MFMA loop, consumed after it
#pragma unroll
        for (int g = 0; g < 4; ++g) gv[g] = p.gates[(long)b * K + (long)g * H + j0 + j];
        c = p.c[e];
        cprev = p.cprev[e];
        if (p.dy) dyv = p.dy[(long)b * p.dy_stride_b + j0 + j];
        dcin = p.first ? (p.dcinit ? p.dcinit[e] : 0.f) : p.dc[e];
        if (p.dhinit) dh0 = p.dhinit[e];
    }

    if (p.dgp_next) {
        const int nblk = nkb4 / NW;
        const char *ap = (const char *)p.dgp_next + ((long)bt * nkb4 + wave * nblk) * PK::BLOCK_BYTES + lane * 16;
        const char *bp = (const char *)p.wpT + ((long)jt * nkb4 + wave * nblk) * PK::BLOCK_BYTES + lane * 16;
        const f32x4 acc = packed_dot<X3, ONE>(ap, bp, nblk);
        const int r = lane & 15, q = lane >> 4;
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) red[wave][(4 * q + e2) * 16 + r] = acc[e2];
    }
    __syncthreads();

    if (threadIdx.x < 256) {
        float dg[4] = {0.f, 0.f, 0.f, 0.f};
        if (cell) {
            float dh = dh0;
            if (p.dgp_next) {
#pragma unroll
                for (int k = 0; k < NW; ++k) dh += red[k][threadIdx.x];
            }
            const float ig = gv[0], fg = gv[1], gg = gv[2], og = gv[3];
            const float tc = tanhf(c);
            if (p.dy) {
                float d = dyv;
                if (p.dy_relu && !(og * tc > 0.f)) d = 0.f;
                dh += d;
            }
            float dcc = dcin + dh * og * (1.f - tc * tc);
            const float d_o = dh * tc;
            const float d_i = dcc * gg, d_f = dcc * cprev, d_g = dcc * ig;
            p.dc[e] = dcc * fg;
            dg[0] = d_i * ig * (1.f - ig);
            dg[1] = d_f * fg * (1.f - fg);
            dg[2] = d_g * (1.f - gg * gg);
            dg[3] = d_o * og * (1.f - og);
            float *gp = p.gates + (long)b * K + j0 + j;
            gp[0] = dg[0]; gp[H] = dg[1]; gp[2 * (long)H] = dg[2]; gp[3 * (long)H] = dg[3];
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) PK::store(p.dgp_out, bt, K, i, g * H + j0 + j, dg[g]);
    }
}

// ---- pack kernels: one thread per (block, lane) ------------------------------------------------
// logical operand X[row][k]; element (row, k) of tile `tile` (16 rows) goes through Packed<X3>::store's map.
// wp  : tile = (jt*4 + g), row r -> W[g*H + jt*16 + r][k]            (k over H)
// wpT : tile = jt,         row r -> W[k][jt*16 + r]                  (k over 4H)
// rows: tile = bt,         row r -> x[bt*16 + r][k] (0 beyond B)     (k over W)
// MODE 2 also initialises the layer's row-major state rows: h_rm <- src (or 0), c_rm <- csrc (or 0)
template <bool X3, int MODE, bool HI_ONLY = false>   // MODE 0: wp, 1: wpT, 2: rows; HI_ONLY: the lo halves are not written (single-pass bf16 readers)
__device__ __forceinline__ void pack_unit(long u, const float *__restrict__ src, void *__restrict__ dst, int H, int B, int Kdim,
                                          float *__restrict__ h_rm, const float *__restrict__ csrc, float *__restrict__ c_rm) {
    using PK = Packed<X3>;
    constexpr int EPL = X3 ? 8 : 4;            // elements per lane and block
    const int nkb = Kdim / PK::KB;
    const int lane = (int)(u & 63);
    long blk = u >> 6;
    const int kb = (int)(blk % nkb);
    const int tile = (int)(blk / nkb);
    const int r = lane & 15, k0 = kb * PK::KB + (lane >> 4) * EPL;
    float x[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        float v = 0.f;
        if (MODE == 0) {
            const int g = tile & 3, jt = tile >> 2;
            v = src[((long)g * H + jt * 16 + r) * H + k0 + e];
        } else if (MODE == 1) {
            v = src[(long)(k0 + e) * H + tile * 16 + r];
        } else {
            const int bb = tile * 16 + r;
            if (src && bb < B) v = src[(long)bb * Kdim + k0 + e];
            if (bb < B) {
                if (h_rm) h_rm[(long)bb * Kdim + k0 + e] = v;
                if (c_rm) c_rm[(long)bb * Kdim + k0 + e] = csrc ? csrc[(long)bb * Kdim + k0 + e] : 0.f;
            }
        }
        x[e] = v;
    }
    char *base = (char *)dst + blk * PK::BLOCK_BYTES;
    if (X3) {
        bf16x8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const __bf16 hh = (__bf16)x[e];
            hi[e] = hh;
            lo[e] = (__bf16)(x[e] - (float)hh);
        }
        *reinterpret_cast<bf16x8 *>(base + lane * 16) = hi;
        if (!HI_ONLY) *reinterpret_cast<bf16x8 *>(base + 1024 + lane * 16) = lo;
    } else {
        f32x4 v4;
#pragma unroll
        for (int e = 0; e < 4; ++e) v4[e] = x[e];
        *reinterpret_cast<f32x4 *>(base + lane * 16) = v4;
    }
}

template <bool X3, int MODE>
__global__ __launch_bounds__(256) void pack_kernel(const float *__restrict__ src, void *__restrict__ dst, int H, int B,
                                                   int Kdim, long units, float *__restrict__ h_rm,
                                                   const float *__restrict__ csrc, float *__restrict__ c_rm) {
    for (long u = blockIdx.x * 256L + threadIdx.x; u < units; u += (long)gridDim.x * 256)
        pack_unit<X3, MODE>(u, src, dst, H, B, Kdim, h_rm, csrc, c_rm);
}

// Everything a persistent recurrence needs ahead of its launch, in ONE launch: the packed image of W_hh (WMODE 0, forward) or
// W_hh^T (WMODE 1, backward), for the forward also the packed + row-major initial state (pack MODE 2), and the zeroed epoch words.
struct PrologueArgs {
    const float *w; void *wdst; long w_units; int wK;
    const float *h0; void *hp0; float *h_rm; const float *c0; float *c_rm; long s_units;
    unsigned *zero; long zero_units;      // 16-byte units
    unsigned *zero2; long zero2_units;    // a second region to clear (the padded last row tile of an operand image), may be empty
    int H, B;
};
template <int WMODE>
__global__ __launch_bounds__(256) void persist_prologue_kernel(const PrologueArgs a) {
    const long total = a.w_units + a.s_units + a.zero_units + a.zero2_units;
    for (long u = blockIdx.x * 256L + threadIdx.x; u < total; u += (long)gridDim.x * 256) {
        if (u < a.w_units) pack_unit<true, WMODE>(u, a.w, a.wdst, a.H, a.B, a.wK, nullptr, nullptr, nullptr);
        else if (u < a.w_units + a.s_units) pack_unit<true, 2>(u - a.w_units, a.h0, a.hp0, a.H, a.B, a.H, a.h_rm, a.c0, a.c_rm);
        else if (u < a.w_units + a.s_units + a.zero_units) reinterpret_cast<uint4 *>(a.zero)[u - a.w_units - a.s_units] = make_uint4(0u, 0u, 0u, 0u);
        else reinterpret_cast<uint4 *>(a.zero2)[u - a.w_units - a.s_units - a.zero_units] = make_uint4(0u, 0u, 0u, 0u);
    }
}

// The same for the two-layer persistent launch (lstm_persist2.hip): three weight images (W_hh0, W_hh1, W_ih1; transposed for the
// backward), both layers' initial states, the epoch words.  That launch runs in single-pass bf16 and reads the hi halves only: the lo
// halves of the packed blocks are not written (a third of the prologue's bytes).
// GEMM operand images (tiled_image.h) of up to four fp32 matrices, written by `blocks` workgroups of a packing launch: layer lo's input
// and its W_ih, the operands of the input projection (that product's own operand launch is gone), and the transposes of both, which the
// backward's weight-gradient and input-gradient products want (its operand launch is gone too).  tr: the source is [K][R].
struct ImageJobs {
    struct { const float *src; int R, K, ld, KT, tr, first; char *img; } job[4];
    int n, blocks, with_lo;
};
// block b of the jobs; tile: 32 x 129 floats of LDS (transposed sources only)
__device__ __forceinline__ void image_jobs_block(const ImageJobs &a, int b, float *tile) {
    int j = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (i < a.n && b >= a.job[i].first) j = i;
    const int local = b - a.job[j].first, kt = local % a.job[j].KT, rt = local / a.job[j].KT;
    if (a.job[j].tr)
        halo_img::prep_transposed_block(a.job[j].src, a.job[j].R, a.job[j].K, a.job[j].ld, a.job[j].img, a.job[j].KT, a.with_lo, kt, rt,
                                        reinterpret_cast<float (*)[halo_img::TR + 1]>(tile));
    else
        halo_img::prep_rowmajor_block(a.job[j].src, a.job[j].R, a.job[j].K, a.job[j].ld, a.job[j].img, a.job[j].KT, a.with_lo, kt, rt);
}
inline void image_jobs_add(ImageJobs &a, const float *src, int R, int K, int ld, int tr, char *img) {
    const int KT = (K + 31) / 32;
    a.job[a.n++] = {src, R, K, ld, KT, tr, a.blocks, img};
    a.blocks += ((R + 127) / 128) * KT;
}

struct Prologue2Args {
    const float *w[3]; void *wdst[3]; long w_units; int wK;         // w_units per matrix
    const float *h0[2]; void *hp0[2]; float *h_rm[2]; const float *c0[2]; float *c_rm[2]; long s_units;   // per layer (forward only)
    unsigned *zero; long zero_units;
    unsigned *zero2; long zero2_units;
    int H, B;
    ImageJobs img;          // the first img.blocks workgroups of persist2_prologue_kernel (pack_pair: behind its tile blocks)
};
template <int WMODE>
__global__ __launch_bounds__(256) void persist2_prologue_kernel(const Prologue2Args a) {
    if ((int)blockIdx.x < a.img.blocks) {
        __shared__ float tile[32 * (halo_img::TR + 1)];
        image_jobs_block(a.img, blockIdx.x, tile);
        return;
    }
    const long nw = 3 * a.w_units, ns = 2 * a.s_units;
    const long total = nw + ns + a.zero_units + a.zero2_units;
    for (long u = (blockIdx.x - a.img.blocks) * 256L + threadIdx.x; u < total; u += (long)(gridDim.x - a.img.blocks) * 256) {
        if (u < nw) {
            const int m = (int)(u / a.w_units);
            pack_unit<true, WMODE, true>(u - m * a.w_units, a.w[m], a.wdst[m], a.H, a.B, a.wK, nullptr, nullptr, nullptr);
        } else if (u < nw + ns) {
            const int l = (int)((u - nw) / a.s_units);
            pack_unit<true, 2, true>(u - nw - l * a.s_units, a.h0[l], a.hp0[l], a.H, a.B, a.H, a.h_rm[l], a.c0[l], a.c_rm[l]);
        } else if (u < nw + ns + a.zero_units) {
            reinterpret_cast<uint4 *>(a.zero)[u - nw - ns] = make_uint4(0u, 0u, 0u, 0u);
        } else {
            reinterpret_cast<uint4 *>(a.zero2)[u - nw - ns - a.zero_units] = make_uint4(0u, 0u, 0u, 0u);
        }
    }
}

// The two-layer launches' six weight images -- W_hh0, W_hh1, W_ih1 packed for the forward (pack MODE 0) AND transposed for the backward
// (MODE 1), hi halves -- from ONE read of the three matrices at the head of the forward: a wave takes a 32 x 32 tile of W [4H][H], writes
// its two forward blocks (16 rows x 32 k each) straight from the registers it loaded, turns the tile through LDS and writes the two
// transposed blocks (32 k x 16 columns each).  The backward of the same step then packs nothing (halo_lstm_bwd finds the images in the
// reserve).  The states and the epoch words ride along in the tail blocks of the same launch.
struct PackPairArgs {
    const float *w[3];
    char *fwd[3], *tr[3];
    int H;
    int tile_blocks;          // workgroups that pack tiles (4 tiles each); the rest run the prologue's unit loop
    Prologue2Args rest;       // w_units = 0: states + zeroed words
    // zcount regions of zunits 16-byte units each, zstride bytes apart, in each of the two zbase buffers: column block 0 (the zero
    // initial state) of the two h_prev^T operand images the forward launch fills from block 1 on
    char *zbase[2];
    long zstride, zunits;
    int zcount;
};
__global__ __launch_bounds__(256) void persist2_pack_pair_kernel(const PackPairArgs a) {
    __shared__ float tile[4][32][33];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if ((int)blockIdx.x >= a.tile_blocks && (int)blockIdx.x < a.tile_blocks + a.rest.img.blocks) {
        image_jobs_block(a.rest.img, blockIdx.x - a.tile_blocks, &tile[0][0][0]);
        return;
    }
    if ((int)blockIdx.x >= a.tile_blocks) {
        const Prologue2Args &r = a.rest;
        const long ns = 2 * r.s_units, nz = 2L * a.zcount * a.zunits, total = ns + r.zero_units + r.zero2_units + nz;
        const int first = a.tile_blocks + a.rest.img.blocks;
        for (long u = (blockIdx.x - first) * 256L + threadIdx.x; u < total; u += (long)(gridDim.x - first) * 256) {
            if (u >= ns + r.zero_units + r.zero2_units) {
                const long v = u - (ns + r.zero_units + r.zero2_units);
                const long per = (long)a.zcount * a.zunits;
                const int which = (int)(v / per);
                const long w = v % per;
                reinterpret_cast<uint4 *>(a.zbase[which] + (w / a.zunits) * a.zstride)[w % a.zunits] = make_uint4(0u, 0u, 0u, 0u);
            } else if (u < ns) {
                const int l = (int)(u / r.s_units);
                pack_unit<true, 2, true>(u - l * r.s_units, r.h0[l], r.hp0[l], r.H, r.B, r.H, r.h_rm[l], r.c0[l], r.c_rm[l]);
            } else if (u < ns + r.zero_units) {
                reinterpret_cast<uint4 *>(r.zero)[u - ns] = make_uint4(0u, 0u, 0u, 0u);
            } else {
                reinterpret_cast<uint4 *>(r.zero2)[u - ns - r.zero_units] = make_uint4(0u, 0u, 0u, 0u);
            }
        }
        return;
    }
    const int H = a.H, nkb = H / 32, nkb4 = 4 * H / 32;
    const int tiles_c = H / 32, tiles_per = (4 * H / 32) * tiles_c;
    const long t = (long)blockIdx.x * 4 + wave;                  // tile index over the three matrices (tile_blocks * 4 == 3 * tiles_per)
    const int m = (int)(t / tiles_per), tt = (int)(t % tiles_per);
    const int R0 = (tt / tiles_c) * 32, C0 = (tt % tiles_c) * 32;
    const float *w = a.w[m];
    const int i = lane & 15, kg = lane >> 4;
    const int g = R0 / H, jt = (R0 % H) / 16;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const float *src = w + (long)(R0 + half * 16 + i) * H + C0 + kg * 8;
        const f32x4 v0 = *reinterpret_cast<const f32x4 *>(src), v1 = *reinterpret_cast<const f32x4 *>(src + 4);
        bf16x8 hi;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            hi[e] = (__bf16)v0[e]; hi[4 + e] = (__bf16)v1[e];
            tile[wave][half * 16 + i][kg * 8 + e] = v0[e];
            tile[wave][half * 16 + i][kg * 8 + 4 + e] = v1[e];
        }
        *reinterpret_cast<bf16x8 *>(a.fwd[m] + (((long)(jt + half) * 4 + g) * nkb + C0 / 32) * 2048 + lane * 16) = hi;
    }
    __builtin_amdgcn_wave_barrier();                             // the tile is this wave's own: LDS program order within the wave suffices
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        bf16x8 hi;
#pragma unroll
        for (int e = 0; e < 8; ++e) hi[e] = (__bf16)tile[wave][kg * 8 + e][c * 16 + i];
        *reinterpret_cast<bf16x8 *>(a.tr[m] + ((long)(C0 / 16 + c) * nkb4 + R0 / 32) * 2048 + lane * 16) = hi;
    }
}

// =================================================================================================
// Layer-diagonal ("wavefront") fusion.  Step t of layer l only needs step t of layer l-1 and step
// t-1 of layer l, so ONE launch runs step d-l of every layer l (grid.z = layer): a 2-layer, T=21
// forward is 22 launches instead of 42, and each launch moves 2-3x the bytes for the same fixed
// launch/prologue/epilogue cost.  For l > 0 the input projection is folded into the step as extra
// contraction depth (A = [packed y_{l-1,t} ; packed h_{l,t-1}], B = packed [W_ih ; W_hh]); in the
// backward the gradient arriving from the layer above is the second half of an 8H-deep contraction
// (A = [dG_l[t+1] ; dG_{l+1}[t]], B = [W_hh_l^T ; W_ih_{l+1}^T]) with the dropout mask applied to
// that half only.  bf16x3 packed format only; requires H % 64 == 0 (else the per-layer path runs).
// =================================================================================================
constexpr int MAXL = 4;

struct DiagFwdLayer {
    const void *hp_prev;   // packed h_{l,t-1}
    const void *xp;        // packed y_{l-1,t} (l > 0), else NULL
    const void *wp;        // l = 0: packed W_hh ; l > 0: packed [W_ih ; W_hh] (K = 2H)
    float *gates;          // [B,4H]  l = 0 in: x W_ih^T + b, out: activated ; l > 0: out only
    const float *b_ih, *b_hh;   // l > 0
    const float *cprev;
    float *hout, *cout;
    void *hp_out;
    float *y;              // row-major second output (dropped y for the next layer / relu features), may be NULL
    long y_stride_b;
    int y_mode;
    void *yp_out;          // packed copy of y for the next layer's A operand, may be NULL
    uint64_t drop_base;
    DropoutCfg drop;
    int active;
};
struct DiagFwdArgs {
    DiagFwdLayer layer[MAXL];
    int B, H;
};

__global__ __launch_bounds__(512) void lstm_diag_fwd_kernel(const DiagFwdArgs args) {
    using PK = Packed<true>;
    constexpr int KS = 2, NW = 8;
    const DiagFwdLayer &p = args.layer[blockIdx.z];
    if (!p.active) return;
    extern __shared__ __attribute__((aligned(16))) char dyn_lds[];    // [red: 8*256 f32][A tile (layer 0)]
    float (*red)[256] = reinterpret_cast<float (*)[256]>(dyn_lds);
    const int jt = blockIdx.x, bt = blockIdx.y;
    const int j0 = jt * 16, b0 = bt * 16;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int gate = wave & 3, ks = wave >> 2;
    const int H = args.H, nkb = H / 32;

    const int i = threadIdx.x >> 4, j = threadIdx.x & 15;
    const int b = b0 + i;
    const bool cell = threadIdx.x < 256 && b < args.B;
    float gin[4] = {0.f, 0.f, 0.f, 0.f}, cprev = 0.f;
    if (cell) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
            gin[g] = p.xp ? p.b_ih[g * H + j0 + j] + p.b_hh[g * H + j0 + j] : p.gates[(long)b * 4 * H + (long)g * H + j0 + j];
        cprev = p.cprev[(long)b * H + j0 + j];
    }
    f32x4 acc;
    if (p.xp) {
        // K = 2H: k-slice 0 contracts the lower layer's output, k-slice 1 the recurrent state
        const char *a_img = (const char *)(ks == 0 ? p.xp : p.hp_prev);
        const char *ap = a_img + (long)bt * nkb * PK::BLOCK_BYTES + lane * 16;
        const char *bp = (const char *)p.wp + (((long)jt * 4 + gate) * 2 * nkb + (long)ks * nkb) * PK::BLOCK_BYTES + lane * 16;
        acc = packed_dot<true>(ap, bp, nkb);
    } else {
        const int nblk = nkb / KS;
        const char *bp = (const char *)p.wp + (((long)jt * 4 + gate) * nkb + ks * nblk) * PK::BLOCK_BYTES + lane * 16;
        if (nblk % (2 * WCH) == 0) {
            char *a_tile = dyn_lds + NW * 256 * sizeof(float);
            const char *a_src = (const char *)p.hp_prev + (long)bt * nkb * PK::BLOCK_BYTES;
            const ATileStage stage = {a_src, a_tile, nkb * PK::BLOCK_BYTES / 1024, __builtin_amdgcn_readfirstlane(wave), NW, lane};
            lds_cchar *a_lds = (lds_cchar *)(a_tile + (long)ks * nblk * PK::BLOCK_BYTES + lane * 16);
            acc = dot_lds_a<true>(stage, a_lds, bp, nblk);
        } else {
            const char *ap = (const char *)p.hp_prev + ((long)bt * nkb + ks * nblk) * PK::BLOCK_BYTES + lane * 16;
            acc = packed_dot<true>(ap, bp, nblk);
        }
    }
    {
        const int r = lane & 15, q = lane >> 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) red[wave][(4 * q + e) * 16 + r] = acc[e];
    }
    __syncthreads();

    if (threadIdx.x < 256) {
        float h = 0.f, yv = 0.f;
        if (cell) {
            float pre[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) pre[g] = red[g][threadIdx.x] + red[4 + g][threadIdx.x] + gin[g];
            const float ig = sigmoidf_(pre[0]), fg = sigmoidf_(pre[1]), gg = tanhf(pre[2]), og = sigmoidf_(pre[3]);
            const long e = (long)b * H + j0 + j;
            const float c = fg * cprev + ig * gg;
            h = og * tanhf(c);
            float *gp = p.gates + (long)b * 4 * H + j0 + j;
            gp[0] = ig; gp[H] = fg; gp[2 * (long)H] = gg; gp[3 * (long)H] = og;
            p.cout[e] = c;
            p.hout[e] = h;
            yv = h;
            if (p.y_mode == Y_RELU) yv = fmaxf(h, 0.f);
            else if (p.y_mode == Y_DROPOUT) yv = h * dropout_mult(p.drop, p.drop_base + (uint64_t)e);
            if (p.y) p.y[(long)b * p.y_stride_b + j0 + j] = yv;
        }
        PK::store(p.hp_out, bt, H, i, j0 + j, h);
        if (p.yp_out) PK::store(p.yp_out, bt, H, i, j0 + j, yv);
    }
}

struct DiagBwdLayer {
    const void *dgp_next;     // packed dG_l[t+1], NULL at t = T-1
    const void *dgp_above;    // packed dG_{l+1}[t], NULL for the top layer
    const void *wpT_hh;       // packed W_hh_l^T        [H/16][4H/32] blocks
    const void *wpT_ih_above; // packed W_ih_{l+1}^T    [H/16][4H/32] blocks
    float *gates;
    void *dgp_out;
    const float *c, *cprev;
    float *dc;
    const float *dy;          // top layer: gradient w.r.t. the (relu'd) output, strided; may be NULL
    long dy_stride_b;
    int dy_relu;
    const float *dhinit, *dcinit;
    int first;
    uint64_t drop_base;       // flat index of (t, 0, 0) in this layer's [T,B,H] output (its dropout mask)
    DropoutCfg drop;
    int active;
};
struct DiagBwdArgs {
    DiagBwdLayer layer[MAXL];
    int B, H;
};

__global__ __launch_bounds__(1024) void lstm_diag_bwd_kernel(const DiagBwdArgs args) {
    using PK = Packed<true>;
    constexpr int NW = 16;
    const DiagBwdLayer &p = args.layer[blockIdx.z];
    if (!p.active) return;
    __shared__ float red[NW][256];
    const int jt = blockIdx.x, bt = blockIdx.y;
    const int j0 = jt * 16, b0 = bt * 16;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int H = args.H, K = 4 * H, nkb4 = K / 32;

    const int i = threadIdx.x >> 4, j = threadIdx.x & 15;
    const int b = b0 + i;
    const bool cell = threadIdx.x < 256 && b < args.B;
    const long e = (long)b * H + j0 + j;
    float gv[4] = {0.f, 0.f, 0.f, 0.f}, c = 0.f, cprev = 0.f, dyv = 0.f, dcin = 0.f, dh0 = 0.f;
    if (cell) {
#pragma unroll
        for (int g = 0; g < 4; ++g) gv[g] = p.gates[(long)b * K + (long)g * H + j0 + j];
        c = p.c[e];
        cprev = p.cprev[e];
        if (p.dy) dyv = p.dy[(long)b * p.dy_stride_b + j0 + j];
        dcin = p.first ? (p.dcinit ? p.dcinit[e] : 0.f) : p.dc[e];
        if (p.dhinit) dh0 = p.dhinit[e];
    }
    // waves 0..7 (or all 16 for the top layer): recurrent half; waves 8..15: the half from the layer above
    const bool two = p.dgp_above != nullptr;
    const int seg = two ? (wave >> 3) : 0;
    const int wseg = two ? (wave & 7) : wave;
    // H % 64 == 0 makes nkb4 = H/8 a multiple of 8: a segment is cut over 8 waves, or over 16 when it divides
    const int nw_seg = (!two && nkb4 % 16 == 0) ? 16 : 8;
    const int nblk = nkb4 / nw_seg;
    const void *a_img = seg ? p.dgp_above : p.dgp_next;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (a_img && wseg < nw_seg) {
        const char *ap = (const char *)a_img + ((long)bt * nkb4 + wseg * nblk) * PK::BLOCK_BYTES + lane * 16;
        const char *bp = (const char *)(seg ? p.wpT_ih_above : p.wpT_hh) + ((long)jt * nkb4 + wseg * nblk) * PK::BLOCK_BYTES + lane * 16;
        acc = packed_dot<true>(ap, bp, nblk);
    }
    {
        const int r = lane & 15, q = lane >> 4;
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) red[wave][(4 * q + e2) * 16 + r] = acc[e2];
    }
    __syncthreads();

    if (threadIdx.x < 256) {
        float dg[4] = {0.f, 0.f, 0.f, 0.f};
        if (cell) {
            float rec = 0.f, above = 0.f;
            if (two) {
#pragma unroll
                for (int k = 0; k < 8; ++k) { rec += red[k][threadIdx.x]; above += red[8 + k][threadIdx.x]; }
                above *= dropout_mult(p.drop, p.drop_base + (uint64_t)e);   // this layer's own output mask
            } else {
#pragma unroll
                for (int k = 0; k < NW; ++k) rec += red[k][threadIdx.x];
            }
            float dh = dh0 + rec + above;
            const float ig = gv[0], fg = gv[1], gg = gv[2], og = gv[3];
            const float tc = tanhf(c);
            if (p.dy) {
                float d = dyv;
                if (p.dy_relu && !(og * tc > 0.f)) d = 0.f;
                dh += d;
            }
            float dcc = dcin + dh * og * (1.f - tc * tc);
            const float d_o = dh * tc;
            const float d_i = dcc * gg, d_f = dcc * cprev, d_g = dcc * ig;
            p.dc[e] = dcc * fg;
            dg[0] = d_i * ig * (1.f - ig);
            dg[1] = d_f * fg * (1.f - fg);
            dg[2] = d_g * (1.f - gg * gg);
            dg[3] = d_o * og * (1.f - og);
            float *gp = p.gates + (long)b * K + j0 + j;
            gp[0] = dg[0]; gp[H] = dg[1]; gp[2 * (long)H] = dg[2]; gp[3 * (long)H] = dg[3];
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) PK::store(p.dgp_out, bt, K, i, g * H + j0 + j, dg[g]);
    }
}

// packed [W_ih ; W_hh] for the fused forward of a layer l > 0: tile (jt*4+g), k over 2H
__global__ __launch_bounds__(256) void pack_cat_kernel(const float *__restrict__ w_ih, const float *__restrict__ w_hh,
                                                       void *__restrict__ dst, int H, long units) {
    const int nkb = 2 * H / 32;
    for (long u = blockIdx.x * 256L + threadIdx.x; u < units; u += (long)gridDim.x * 256) {
        const int lane = (int)(u & 63);
        long blk = u >> 6;
        const int kb = (int)(blk % nkb);
        const int tile = (int)(blk / nkb);
        const int g = tile & 3, jt = tile >> 2;
        const int r = lane & 15, k0 = kb * 32 + (lane >> 4) * 8;
        const float *src = k0 < H ? w_ih + ((long)g * H + jt * 16 + r) * H + k0 : w_hh + ((long)g * H + jt * 16 + r) * H + (k0 - H);
        bf16x8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float xv = src[e];
            const __bf16 hh = (__bf16)xv;
            hi[e] = hh;
            lo[e] = (__bf16)(xv - (float)hh);
        }
        char *base = (char *)dst + blk * 2048;
        *reinterpret_cast<bf16x8 *>(base + lane * 16) = hi;
        *reinterpret_cast<bf16x8 *>(base + 1024 + lane * 16) = lo;
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

// HALO_MATH_BF16 keeps the split packed images and skips their lo halves in the step kernels (launch_step_*)
inline bool use_x3(int H) { return halo_math_mode() != HALO_MATH_F32 && H % 32 == 0; }

// waves = (gate, k-slice): the slice count must divide the number of k-blocks
inline int pick_fwd_ks(int H, bool x3) {
    const int nkb = H / (x3 ? 32 : 16);
    static const int cap = getenv("HALO_LSTM_KS") ? atoi(getenv("HALO_LSTM_KS")) : 2;   // 8 waves measured best at H=1024
    return (nkb % 4 == 0 && cap >= 4) ? 4 : (nkb % 2 == 0 && cap >= 2) ? 2 : 1;
}
inline int pick_bwd_nw(int H, bool x3) {
    const int nkb4 = 4 * H / (x3 ? 32 : 16);
    return (nkb4 % 16 == 0) ? 16 : (nkb4 % 8 == 0) ? 8 : 4;
}

template <int KS, bool X3, bool ONE>
int launch_step_fwd_ks(const StepFwdArgs &a, hipStream_t st) {
    dim3 grid(a.H / 16, (a.B + 15) / 16);
    const int nkb = a.H / (X3 ? 32 : 16);
    const size_t red_bytes = (size_t)4 * KS * 256 * sizeof(float);
    const size_t tile_bytes = (size_t)nkb * (X3 ? 2048 : 1024);
    // LDS-staged A tile when the per-wave block count suits its 2-stage W pipeline and the tile fits
    static const bool want_alds = !(getenv("HALO_LSTM_ALDS") && atoi(getenv("HALO_LSTM_ALDS")) == 0);
    const bool alds = want_alds && ((nkb / KS) % (2 * WCH) == 0) && red_bytes + tile_bytes <= 128 * 1024;
    if (alds) {
        static bool attr = false;
        if (!attr) {
            if (hipFuncSetAttribute((const void *)lstm_step_fwd_kernel<KS, X3, true, ONE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    128 * 1024) != hipSuccess)
                return HALO_ELAUNCH;
            attr = true;
        }
        hipLaunchKernelGGL((lstm_step_fwd_kernel<KS, X3, true, ONE>), grid, dim3(256 * KS), red_bytes + tile_bytes, st, a);
    } else {
        hipLaunchKernelGGL((lstm_step_fwd_kernel<KS, X3, false, ONE>), grid, dim3(256 * KS), red_bytes, st, a);
    }
    return halo_launch_status();
}

template <bool X3, bool ONE>
int launch_step_fwd_t(const StepFwdArgs &a, hipStream_t st) {
    switch (pick_fwd_ks(a.H, X3)) {
        case 4: return launch_step_fwd_ks<4, X3, ONE>(a, st);
        case 2: return launch_step_fwd_ks<2, X3, ONE>(a, st);
        default: return launch_step_fwd_ks<1, X3, ONE>(a, st);
    }
}
// HALO_MATH_BF16: the split (hi|lo) packed images are kept, the kernels fetch and multiply only their hi halves
inline bool one_pass() { return halo_math_mode() == HALO_MATH_BF16; }
int launch_step_fwd(const StepFwdArgs &a, bool x3, hipStream_t st) {
    if (x3 && one_pass()) return launch_step_fwd_t<true, true>(a, st);
    return x3 ? launch_step_fwd_t<true, false>(a, st) : launch_step_fwd_t<false, false>(a, st);
}

template <bool X3, bool ONE>
int launch_step_bwd_t(const StepBwdArgs &a, hipStream_t st) {
    dim3 grid(a.H / 16, (a.B + 15) / 16);
    switch (pick_bwd_nw(a.H, X3)) {
        case 16: hipLaunchKernelGGL((lstm_step_bwd_kernel<16, X3, ONE>), grid, dim3(1024), 0, st, a); break;
        case 8: hipLaunchKernelGGL((lstm_step_bwd_kernel<8, X3, ONE>), grid, dim3(512), 0, st, a); break;
        default: hipLaunchKernelGGL((lstm_step_bwd_kernel<4, X3, ONE>), grid, dim3(256), 0, st, a); break;
    }
    return halo_launch_status();
}
int launch_step_bwd(const StepBwdArgs &a, bool x3, hipStream_t st) {
    if (x3 && one_pass()) return launch_step_bwd_t<true, true>(a, st);
    return x3 ? launch_step_bwd_t<true, false>(a, st) : launch_step_bwd_t<false, false>(a, st);
}

// MODE 0: W_hh for the forward, 1: W_hh^T for the backward, 2: row-major [B,W] rows (initial state)
template <int MODE>
int launch_pack(const float *src, void *dst, int H, int B, int Kdim, int tiles, bool x3, hipStream_t st,
                float *h_rm = nullptr, const float *csrc = nullptr, float *c_rm = nullptr) {
    const long units = (long)tiles * (Kdim / (x3 ? 32 : 16)) * 64;
    long g = (units + 255) / 256;
    if (g > 2048) g = 2048;
    if (x3) hipLaunchKernelGGL((pack_kernel<true, MODE>), dim3((unsigned)g), dim3(256), 0, st, src, dst, H, B, Kdim, units, h_rm, csrc, c_rm);
    else hipLaunchKernelGGL((pack_kernel<false, MODE>), dim3((unsigned)g), dim3(256), 0, st, src, dst, H, B, Kdim, units, h_rm, csrc, c_rm);
    return halo_launch_status();
}

inline unsigned pack_grid(size_t units) {
    size_t g = (units + 255) / 256;
    return (unsigned)(g > 2048 ? 2048 : (g < 1 ? 1 : g));
}

// ---- measurement hook (halo_lstm_chain_events): two events recorded on the launch stream right around a layer's recurrent
// chain, so a caller can time the chain without the batched GEMMs / operand preparation of the same call ----
struct ChainInfo { int launches; const char *kernel; };
ChainInfo g_chain_info[2] = {{0, ""}, {0, ""}};      // [0] forward, [1] backward: what the last chain ran as
inline void chain_begin(hipStream_t st) { if (halo_ctx_cur().chain_ev0) (void)hipEventRecord(halo_ctx_cur().chain_ev0, st); }
inline void chain_end(hipStream_t st, int dir, int launches, const char *kernel) {
    if (halo_ctx_cur().chain_ev1) (void)hipEventRecord(halo_ctx_cur().chain_ev1, st);
    g_chain_info[dir].launches = launches;
    g_chain_info[dir].kernel = kernel;
}

#define HALO_TRY(expr)            \
    do {                          \
        int rc_ = (expr);         \
        if (rc_ != HALO_OK) return rc_; \
    } while (0)

inline int copy_d2d(float *dst, const float *src, size_t n, hipStream_t st) {
    return hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, st) == hipSuccess ? HALO_OK : HALO_ELAUNCH;
}

// ---- layer-diagonal fused path: eligibility, extra buffers, drivers ------------------------------
inline bool fused_ok(int H, int L) {
    return halo_lstm_fusion() && L >= 2 && L <= MAXL && halo_math_mode() != HALO_MATH_F32 && H % 64 == 0;
}
// appended to the reserve: packed forward weights of every layer (all layers are live at once) and the
// packed y_{l,t} images of the non-top layers
inline size_t fused_fwd_extra_floats(int T, int B, int H, int L) {
    return (size_t)4 * H * H + (size_t)(L - 1) * 8 * H * H + (size_t)(L - 1) * T * bt16(B) * H;
}
inline float *fused_wpk(float *extra, int l, int H) {
    return l == 0 ? extra : extra + (size_t)4 * H * H + (size_t)(l - 1) * 8 * H * H;
}
inline float *fused_yp(float *extra, int l, int T, int B, int H, int L) {
    return extra + (size_t)4 * H * H + (size_t)(L - 1) * 8 * H * H + (size_t)l * T * bt16(B) * H;
}
// appended to the backward workspace, per layer: packed W_hh^T, packed W_ih^T, dc carry, two packed dG images
inline size_t fused_bwd_layer_floats(int B, int H) { return (size_t)8 * H * H + (size_t)B * H + 2 * bt16(B) * 4 * H; }

int lstm_fwd_fused(const float *x, const float *const *w_ih, const float *const *w_hh, const float *const *b_ih,
                   const float *const *b_hh, const float *h0, const float *c0, float *y, long y_stride_t, long y_stride_b,
                   int y_relu, float *hn, float *cn, float *reserve, float *extra, char *img_in, char *img_w, int T, int B,
                   int in0, int H, int L, float p_drop, uint64_t seed, uint32_t offset, const uint32_t *offset_dev,
                   hipStream_t st) {
    const size_t BH = (size_t)B * H, PH = bt16(B) * H;
    const int nkb = H / 32;
    // operand preparation for every layer, then layer 0's batched input projection
    for (int l = 0; l < L; ++l) {
        const LayerBufs lb = layer_bufs(reserve, l, T, B, H);
        if (l == 0) {
            HALO_TRY(launch_pack<0>(w_hh[0], fused_wpk(extra, 0, H), H, B, H, (H / 16) * 4, true, st));
        } else {
            const long units = (long)(H / 16) * 4 * (2 * nkb) * 64;
            long g = (units + 255) / 256;
            if (g > 2048) g = 2048;
            hipLaunchKernelGGL(pack_cat_kernel, dim3((unsigned)g), dim3(256), 0, st, w_ih[l], w_hh[l], (void *)fused_wpk(extra, l, H), H, units);
            HALO_TRY(halo_launch_status());
        }
        const float *h0l = h0 ? h0 + (size_t)l * BH : nullptr;
        HALO_TRY(launch_pack<2>(h0l, lb.hp, H, B, H, (B + 15) / 16, true, st, lb.h, c0 ? c0 + (size_t)l * BH : nullptr, lb.c));
    }
    {
        const LayerBufs lb = layer_bufs(reserve, 0, T, B, H);
        if (in0 >= 64) {
            HALO_TRY(halo_prep_tiles(x, T * B, in0, in0, 0, img_in, st));
            HALO_TRY(halo_prep_tiles(w_ih[0], 4 * H, in0, in0, 0, img_w, st));
            HALO_TRY(halo_gemm_bf16x3_tiled(img_in, img_w, T * B, 4 * H, in0, lb.gates, 4 * H, b_ih[0], b_hh[0], 0, nullptr, st));
        } else {
            HALO_TRY(halo_gemm_f32(1, 1, T * B, 4 * H, in0, x, in0, w_ih[0], in0, lb.gates, 4 * H, b_ih[0], b_hh[0], 0, 0.f, 0, 0,
                                   0, nullptr, (halo_stream_t)st));
        }
    }
    const bool alds = (nkb / 2) % (2 * WCH) == 0;
    const size_t lds_bytes = (size_t)8 * 256 * sizeof(float) + (alds ? (size_t)nkb * 2048 : 0);
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void *)lstm_diag_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess)
            return HALO_ELAUNCH;
        attr = true;
    }
    for (int d = 0; d < T + L - 1; ++d) {
        DiagFwdArgs a;
        a.B = B; a.H = H;
        for (int l = 0; l < MAXL; ++l) {
            DiagFwdLayer &q = a.layer[l];
            const int t = d - l;
            q.active = l < L && t >= 0 && t < T;
            if (!q.active) continue;
            const LayerBufs lb = layer_bufs(reserve, l, T, B, H);
            const bool last = (l == L - 1);
            const bool drop_out = !last && p_drop > 0.f;
            q.hp_prev = lb.hp + (size_t)t * PH;
            q.xp = l > 0 ? fused_yp(extra, l - 1, T, B, H, L) + (size_t)t * PH : nullptr;
            q.wp = fused_wpk(extra, l, H);
            q.gates = lb.gates + (size_t)t * B * 4 * H;
            q.b_ih = b_ih[l]; q.b_hh = b_hh[l];
            q.cprev = lb.c + (size_t)t * BH;
            q.hout = lb.h + (size_t)(t + 1) * BH;
            q.cout = lb.c + (size_t)(t + 1) * BH;
            q.hp_out = lb.hp + (size_t)(t + 1) * PH;
            q.drop = make_dropout(drop_out ? p_drop : 0.f, seed, HALO_STREAM_LSTM_LAYER0 + (uint32_t)l, offset, offset_dev);
            q.drop_base = (uint64_t)t * BH;
            if (last) {
                q.y = y ? y + (long)t * y_stride_t : nullptr;
                q.y_stride_b = y_stride_b;
                q.y_mode = y_relu ? Y_RELU : Y_PLAIN;
                q.yp_out = nullptr;
            } else {
                q.y = drop_out ? lb.ydrop + (size_t)t * BH : nullptr;
                q.y_stride_b = H;
                q.y_mode = drop_out ? Y_DROPOUT : Y_PLAIN;
                q.yp_out = fused_yp(extra, l, T, B, H, L) + (size_t)t * PH;
            }
        }
        hipLaunchKernelGGL(lstm_diag_fwd_kernel, dim3(H / 16, (B + 15) / 16, L), dim3(512), lds_bytes, st, a);
        HALO_TRY(halo_launch_status());
    }
    for (int l = 0; l < L; ++l) {
        const LayerBufs lb = layer_bufs(reserve, l, T, B, H);
        if (hn) HALO_TRY(copy_d2d(hn + (size_t)l * BH, lb.h + (size_t)T * BH, BH, st));
        if (cn) HALO_TRY(copy_d2d(cn + (size_t)l * BH, lb.c + (size_t)T * BH, BH, st));
    }
    return HALO_OK;
}

// the backward step chain of ALL layers, fused along layer diagonals; leaves dG of every layer in its gates buffer
int lstm_bwd_fused_chain(const float *const *w_ih, const float *const *w_hh, const float *dy, long y_stride_t,
                         long y_stride_b, int y_relu, const float *dhn, const float *dcn, float *reserve, float *extra,
                         int T, int B, int H, int L, float p_drop, uint64_t seed, uint32_t offset,
                         const uint32_t *offset_dev, hipStream_t st) {
    const size_t BH = (size_t)B * H, PG = bt16(B) * 4 * H, per = fused_bwd_layer_floats(B, H);
    for (int l = 0; l < L; ++l) {
        float *base = extra + (size_t)l * per;
        HALO_TRY(launch_pack<1>(w_hh[l], base, H, B, 4 * H, H / 16, true, st));
        if (l > 0) HALO_TRY(launch_pack<1>(w_ih[l], base + (size_t)4 * H * H, H, B, 4 * H, H / 16, true, st));
    }
    for (int d = 0; d < T + L - 1; ++d) {
        DiagBwdArgs a;
        a.B = B; a.H = H;
        for (int l = 0; l < MAXL; ++l) {
            DiagBwdLayer &q = a.layer[l];
            const int t = l < L ? T - 1 - (d - (L - 1 - l)) : -1;
            q.active = l < L && t >= 0 && t < T;
            if (!q.active) continue;
            const LayerBufs lb = layer_bufs(reserve, l, T, B, H);
            float *base = extra + (size_t)l * per;
            float *dcarry = base + (size_t)8 * H * H;
            float *dgp = dcarry + BH;
            const bool top = (l == L - 1);
            q.dgp_next = (t == T - 1) ? nullptr : dgp + (size_t)((t + 1) & 1) * PG;
            q.dgp_out = dgp + (size_t)(t & 1) * PG;
            q.wpT_hh = base;
            if (!top) {
                float *above = extra + (size_t)(l + 1) * per;
                q.dgp_above = above + (size_t)8 * H * H + BH + (size_t)(t & 1) * PG;
                q.wpT_ih_above = above + (size_t)4 * H * H;
            } else {
                q.dgp_above = nullptr; q.wpT_ih_above = nullptr;
            }
            q.gates = lb.gates + (size_t)t * B * 4 * H;
            q.c = lb.c + (size_t)(t + 1) * BH;
            q.cprev = lb.c + (size_t)t * BH;
            q.dc = dcarry;
            q.dy = (top && dy) ? dy + (long)t * y_stride_t : nullptr;
            q.dy_stride_b = y_stride_b;
            q.dy_relu = y_relu;
            q.first = (t == T - 1);
            q.dhinit = (t == T - 1 && dhn) ? dhn + (size_t)l * BH : nullptr;
            q.dcinit = (t == T - 1 && dcn) ? dcn + (size_t)l * BH : nullptr;
            q.drop = make_dropout(top ? 0.f : p_drop, seed, HALO_STREAM_LSTM_LAYER0 + (uint32_t)l, offset, offset_dev);
            q.drop_base = (uint64_t)t * BH;
        }
        hipLaunchKernelGGL(lstm_diag_bwd_kernel, dim3(H / 16, (B + 15) / 16, L), dim3(1024), 0, st, a);
        HALO_TRY(halo_launch_status());
    }
    return HALO_OK;
}

// reserve = [ ... as laid out above ... | epoch words of the persistent recurrence (PERSIST_FLAG_BYTES, 256-byte aligned) ]
inline size_t reserve_flags_offset(int T, int B, int in0, int H, int L) {
    const int kin = in0 > H ? in0 : H;
    const size_t n = ((size_t)4 * H * H + (size_t)L * layer_floats(T, B, H)) * sizeof(float) +
                     halo_tiled_image_bytes(T * B, kin) + halo_tiled_image_bytes(4 * H, kin) +
                     (L >= 2 ? fused_fwd_extra_floats(T, B, H, L) * sizeof(float) : 0);
    return (n + 255) & ~(size_t)255;
}
// workspace of the backward = [ W_hh^T | dc carry | din | packed gate-gradient images | tiled images | fused extra | epoch words ]
inline int bwd_dg_images(int T) { return T > 2 ? T : 2; }     // the persistent recurrence writes one image per time step
inline size_t bwd_flags_offset(int T, int B, int in0, int H, int L) {
    const int kin = in0 > H ? in0 : H;
    const size_t n = ((size_t)H * 4 * H + (size_t)B * H + (size_t)T * B * H + (size_t)bwd_dg_images(T) * bt16(B) * 4 * H) * sizeof(float) +
                     halo_tiled_image_bytes(4 * H, T * B) + halo_tiled_image_bytes(T * B, 4 * H) +
                     2 * halo_tiled_image_bytes(kin, T * B) + halo_tiled_image_bytes(kin, 4 * H) +
                     (L >= 2 ? (size_t)L * fused_bwd_layer_floats(B, H) * sizeof(float) : 0);
    return (n + 255) & ~(size_t)255;
}

// behind the epoch words of a 2-layer reserve: what the two-layer forward leaves for the backward of the same step -- the three transposed
// weight images, and the operand images of the weight-gradient products [h0_prev^T | in0^T slot] [h1_prev^T | dropout(h0)^T]
// ... and, last, the epoch words of the two-layer BACKWARD launch, zeroed by the forward's packing launch so that the backward of the same
// step starts without a prologue launch of its own
inline size_t reserve_p2_images_bytes(int T, int B, int in0, int H) {
    const int kin = in0 > H ? in0 : H;
    return (size_t)3 * 4 * H * H * sizeof(float) + 3 * halo_tiled_image_bytes(H, T * B) + halo_tiled_image_bytes(kin, T * B);
}
inline size_t reserve_p2_bytes(int T, int B, int in0, int H) {
    return reserve_p2_images_bytes(T, B, in0, H) + PERSIST_FLAG_BYTES + halo_tiled_image_bytes(in0 > H ? in0 : H, 4 * H);
}
inline unsigned *reserve_bwd_flags(float *reserve, int T, int B, int in0, int H, int L) {
    return (unsigned *)((char *)reserve + reserve_flags_offset(T, B, in0, H, L) + PERSIST_FLAG_BYTES + reserve_p2_images_bytes(T, B, in0, H));
}
// ... and behind them the image of W_ih^T of the pair's lower layer (rows: its input width), written by the forward's packing launch
inline char *reserve_p2_wT(float *reserve, int T, int B, int in0, int H, int L) {
    return (char *)reserve_bwd_flags(reserve, T, B, in0, H, L) + PERSIST_FLAG_BYTES;
}
inline char *reserve_p2_images(float *reserve, int T, int B, int in0, int H, int L) {
    return (char *)reserve + reserve_flags_offset(T, B, in0, H, L) + PERSIST_FLAG_BYTES + (size_t)3 * 4 * H * H * sizeof(float);
}

inline bool persist_emit_enabled() {
    int &v = halo_ctx_cur().persist_emit;
    if (v < 0) { const char *e = getenv("HALO_PERSIST_EMIT"); v = e ? (atoi(e) != 0) : 1; }
    return v != 0;
}

// HALO_LSTM_KEEP_DG=1: the persistent backward launches store their fp32 gate gradients back into the gates buffers even where no
// launch reads them any more (the chain emits the operand images and bias partials itself) -- the behaviour before round 4's last day
inline bool keep_dg_switch() {
    static const bool on = getenv("HALO_LSTM_KEEP_DG") && getenv("HALO_LSTM_KEEP_DG")[0] == '1';
    return on;
}

inline bool pair_dw_enabled() {
    static const bool on = !(getenv("HALO_LSTM_PAIR_DW") && atoi(getenv("HALO_LSTM_PAIR_DW")) == 0);
    return on;
}

// What follows a layer's backward chain: every operand image of its gradient GEMMs in ONE launch (dG [TB][4H] for the input
// gradient and dG^T [4H][TB] for both weight gradients unless the chain wrote them itself: ``emit``; W_ih^T, h_prev^T, in^T; the
// bias gradients from the chain's partial rows), then the input gradient (feeds layer l-1's recurrence or the caller's dx) and
// the parameter gradients over all frames at once.
int lstm_bwd_layer_tail(const float *x, const float *const *w_ih, float *reserve, int l, int in0, int T, int B, int H, int L, float p_drop,
                        uint64_t seed, uint32_t offset, const uint32_t *offset_dev, bool fused, bool emit, char *img_g, char *img_gT,
                        char *img_hT, char *img_inT, char *img_wT, const float *bias_part, float *din, float *dx, float *const *dw_ih,
                        float *const *dw_hh, float *const *db_ih, float *const *db_hh, bool need_din, hipStream_t st) {
    const size_t BH = (size_t)B * H;
    const LayerBufs lb = layer_bufs(reserve, l, T, B, H);
    hipStream_t side = st;
    const float *in;
    int in_dim;
    if (l == 0) { in = x; in_dim = in0; }
    else {
        const LayerBufs pb = layer_bufs(reserve, l - 1, T, B, H);
        in = p_drop > 0.f ? pb.ydrop : pb.h + BH;
        in_dim = H;
    }
    (void)fused;
    float *din_out = l > 0 ? din : dx;
    const DropoutCfg ddrop = make_dropout(l > 0 ? p_drop : 0.f, seed, HALO_STREAM_LSTM_LAYER0 + (uint32_t)(l > 0 ? l - 1 : 0),
                                          offset, offset_dev);
    const bool tiled = halo_math_mode() != HALO_MATH_F32 && in_dim >= 64;
    int rc = HALO_OK;
    if (tiled) {
        HaloPrepJob jobs[5];
        int nj = 0;
        if (need_din) {
            if (!emit) jobs[nj++] = {2, lb.gates, T * B, 4 * H, 4 * H, img_g, img_gT};              // (emit: the chain wrote both)
            jobs[nj++] = {1, w_ih[l], in_dim, 4 * H, in_dim, img_wT, nullptr};                      // W_ih^T [in][4H]
        } else if (!emit) {
            jobs[nj++] = {1, lb.gates, 4 * H, T * B, 4 * H, img_gT, nullptr};                       // dG^T [4H][TB]
        }
        if (emit)       // the bias gradients: the chain left one partial row per batch tile, summed here beside the other jobs
            jobs[nj++] = {3, bias_part, (B + 15) / 16, 4 * H, 4 * H, db_ih[l], db_hh[l]};
        jobs[nj++] = {1, lb.h, H, T * B, H, img_hT, nullptr};                                       // h_prev^T [H][TB]
        jobs[nj++] = {1, in, in_dim, T * B, in_dim, img_inT, nullptr};                              // in^T [in][TB]
        HALO_TRY(halo_prep_jobs(jobs, nj, st));
    }
    // (1) the gradient w.r.t. this layer's input feeds layer l-1's recurrence
    if (need_din) {
        if (tiled) {
            HALO_TRY(halo_gemm_bf16x3_tiled(img_g, img_wT, T * B, in_dim, 4 * H, din_out, in_dim, nullptr, nullptr, 0,
                                            &ddrop, st));
        } else {
            HALO_TRY(halo_gemm_f32(1, 0, T * B, in_dim, 4 * H, lb.gates, 4 * H, w_ih[l], in_dim, din_out, in_dim,
                                   nullptr, nullptr, 0, l > 0 ? p_drop : 0.f, seed,
                                   HALO_STREAM_LSTM_LAYER0 + (uint32_t)(l > 0 ? l - 1 : 0), offset, offset_dev, (halo_stream_t)st));
        }
    }
    // (2) this layer's parameter gradients
    if (tiled && H % 128 == 0 && img_inT == img_hT + halo_tiled_image_bytes(H, T * B) && pair_dw_enabled()) {
        // dW_hh | dW_ih = dG^T x (h_prev^T stacked on in^T)^T: ONE launch over the two images, which lie one after the other
        // (512 tiles at H = 1024 instead of 2 x 256: one launch's fill and drain instead of two)
        HaloG256Problem gp = {};
        gp.A = img_gT; gp.B = img_hT; gp.M = 4 * H; gp.N = H + in_dim; gp.K = T * B;
        gp.C = dw_hh[l]; gp.ldc = H; gp.n_split = H; gp.C2 = dw_ih[l]; gp.ldc2 = in_dim; gp.kslices = 1;
        // (on 256 x 256 tiles -- gemm256.hip, one workgroup per CU -- when its rounds of 256 tiles are at least 55 % full)
        const long t256 = (long)((4 * H + 255) / 256) * ((H + in_dim + 255) / 256);
        if (halo_gemm256_enabled() && halo_gemm256_fits(gp) && 100 * t256 >= 55 * ((t256 + 255) / 256) * 256) rc = halo_gemm256_launch(&gp, 1, side);
        else
        rc = halo_gemm_bf16x3_tiled_nsplit(img_gT, img_hT, 4 * H, H + in_dim, T * B, dw_hh[l], H, H, dw_ih[l], in_dim, side);
    } else if (tiled) {
        rc = halo_gemm_bf16x3_tiled(img_gT, img_hT, 4 * H, H, T * B, dw_hh[l], H, nullptr, nullptr, 0, nullptr, side);
        if (!rc) rc = halo_gemm_bf16x3_tiled(img_gT, img_inT, 4 * H, in_dim, T * B, dw_ih[l], in_dim, nullptr, nullptr, 0,
                                             nullptr, side);
    } else {
        // dW_hh[4H,H] = dG[T*B,4H]^T * h_{t-1}[T*B,H]   (h buffer rows 0..T-1 are the previous states)
        rc = halo_gemm_f32(0, 0, 4 * H, H, T * B, lb.gates, 4 * H, lb.h, H, dw_hh[l], H, nullptr, nullptr, 0, 0.f, 0, 0,
                           0, nullptr, (halo_stream_t)side);
        // dW_ih[4H,in] = dG^T * in
        if (!rc) rc = halo_gemm_f32(0, 0, 4 * H, in_dim, T * B, lb.gates, 4 * H, in, in_dim, dw_ih[l], in_dim, nullptr,
                                    nullptr, 0, 0.f, 0, 0, 0, nullptr, (halo_stream_t)side);
    }
    if (!rc && !(emit && tiled)) rc = halo_colsum2(lb.gates, T * B, 4 * H, 4 * H, db_ih[l], db_hh[l], side);
    return rc;
}


// ---- both layers in one persistent launch (lstm_persist2.hip; bf16 arithmetic, L = 2) ------------------------------------------
// layer 0's input projection (one batched GEMM), one prologue launch, ONE launch for the 2 T time steps of the stack
int lstm_fwd_persist2(const float *in, int in_dim, int lo, const float *const *w_ih, const float *const *w_hh, const float *const *b_ih,
                      const float *const *b_hh, const float *h0, const float *c0, float *y, long y_stride_t, long y_stride_b,
                      int y_relu, float *hn, float *cn, float *reserve, float *extra, char *img_in, char *img_w, int T, int B,
                      int in0, int H, int L, float p_drop, uint64_t seed, uint32_t offset, const uint32_t *offset_dev, hipStream_t st) {
    // layers lo and lo + 1 = L - 1 (the top pair of the stack; the layers below ran one launch each): ``in`` [T*B][in_dim] is layer lo's input
    const int hi = lo + 1;
    const size_t BH = (size_t)B * H;
    const LayerBufs l0 = layer_bufs(reserve, lo, T, B, H), l1 = layer_bufs(reserve, hi, T, B, H);
    w_ih += lo; w_hh += lo; b_ih += lo; b_hh += lo;                   // index 0 / 1 below = layer lo / lo + 1
    if (h0) h0 += (size_t)lo * BH;
    if (c0) c0 += (size_t)lo * BH;
    HaloCtx &ctx = halo_ctx_cur();
    // the packing / prologue launch below also writes this product's two operand images: it runs first, the product behind it
    const bool images_by_pack = in_dim >= 64;
    if (in_dim < 64) {
        HALO_TRY(halo_gemm_f32(1, 1, T * B, 4 * H, in_dim, in, in_dim, w_ih[0], in_dim, l0.gates, 4 * H, b_ih[0], b_hh[0], 0, 0.f, 0, 0, 0,
                               nullptr, (halo_stream_t)st));
    }
    float *wp0 = reserve, *wp1 = fused_wpk(extra, hi, H), *wpi = wp1 + (size_t)4 * H * H;
    unsigned *flags = (unsigned *)((char *)reserve + reserve_flags_offset(T, B, in0, H, L));
    Prologue2Args pa;
    pa.img.n = 0; pa.img.blocks = 0; pa.img.with_lo = halo_math_mode() != HALO_MATH_BF16;
    pa.w[0] = w_hh[0]; pa.wdst[0] = wp0; pa.w[1] = w_hh[1]; pa.wdst[1] = wp1; pa.w[2] = w_ih[1]; pa.wdst[2] = wpi;
    pa.w_units = (long)(H / 16) * 4 * (H / 32) * 64; pa.wK = H;
    for (int l = 0; l < 2; ++l) {
        const LayerBufs lb = l ? l1 : l0;
        pa.h0[l] = h0 ? h0 + (size_t)l * BH : nullptr; pa.hp0[l] = lb.hp; pa.h_rm[l] = lb.h;
        pa.c0[l] = c0 ? c0 + (size_t)l * BH : nullptr; pa.c_rm[l] = lb.c;
    }
    pa.s_units = (long)((B + 15) / 16) * (H / 32) * 64;
    pa.zero = flags; pa.zero_units = (long)(PERSIST_FLAG_BYTES / 16);
    pa.zero2 = nullptr; pa.zero2_units = 0;
    pa.H = H; pa.B = B;
    ctx.packT_reserve = nullptr;
    ctx.emitT_reserve = nullptr;
    char *emit_hT0 = nullptr, *emit_hT1 = nullptr, *emit_xT1 = nullptr;
    if (ctx.lstm_expect_backward && H % 32 == 0) {
        // a backward will follow (halo_set_lstm_expect_backward): its three transposed images come out of the same read of the weights
        float *wT = (float *)((char *)reserve + reserve_flags_offset(T, B, in0, H, L) + PERSIST_FLAG_BYTES);
        PackPairArgs pp;
        for (int m = 0; m < 3; ++m) { pp.w[m] = pa.w[m]; pp.fwd[m] = (char *)pa.wdst[m]; pp.tr[m] = (char *)(wT + (size_t)m * 4 * H * H); }
        pp.H = H;
        pp.tile_blocks = 3 * (4 * H / 32) * (H / 32) / 4;
        pp.rest = pa; pp.rest.w_units = 0;
        pp.rest.zero2 = reserve_bwd_flags(reserve, T, B, in0, H, L); pp.rest.zero2_units = (long)(PERSIST_FLAG_BYTES / 16);
        pp.zbase[0] = pp.zbase[1] = nullptr; pp.zstride = 0; pp.zunits = 0; pp.zcount = 0;
        ctx.emitT_reserve = nullptr;
        const bool emitT = persist_emit_enabled() && !h0 && B % 32 == 0 && H % 128 == 0 && T * B >= 32;
        if (emitT) {
            // the forward launch writes h_prev^T of both layers and dropout(h0)^T as GEMM operand images; their column block 0 (zero state) here
            char *imgs = reserve_p2_images(reserve, T, B, in0, H, L);
            const size_t hb = halo_tiled_image_bytes(H, T * B), ib = halo_tiled_image_bytes(in0 > H ? in0 : H, T * B);
            emit_hT0 = imgs; emit_hT1 = imgs + hb + ib; emit_xT1 = emit_hT1 + hb;
            const long KT = (T * B + 31) / 32;
            pp.zbase[0] = emit_hT0; pp.zbase[1] = emit_hT1;
            pp.zstride = KT * 16384; pp.zunits = (long)(B / 32) * 16384 / 16; pp.zcount = H / 128;
            ctx.emitT_reserve = reserve;
        }
        ctx.fwdT_reserve = nullptr;
        if (images_by_pack) {
            ImageJobs &ij = pp.rest.img;
            image_jobs_add(ij, in, T * B, in_dim, in_dim, 0, img_in);
            image_jobs_add(ij, w_ih[0], 4 * H, in_dim, in_dim, 0, img_w);
            if (emitT) {        // ... and their transposes for the backward: in^T beside the h_prev^T images, W_ih^T behind the backward's words
                image_jobs_add(ij, in, in_dim, T * B, in_dim, 1, emit_hT0 + halo_tiled_image_bytes(H, T * B));
                image_jobs_add(ij, w_ih[0], in_dim, 4 * H, in_dim, 1, reserve_p2_wT(reserve, T, B, in0, H, L));
                ctx.fwdT_reserve = reserve; ctx.fwdT_src[0] = in; ctx.fwdT_src[1] = w_ih[0];
            }
        }
        const unsigned rest_blocks = pack_grid((size_t)(2 * pa.s_units + pa.zero_units + pp.rest.zero2_units + 2L * pp.zcount * pp.zunits));
        hipLaunchKernelGGL(persist2_pack_pair_kernel, dim3((unsigned)(pp.tile_blocks + pp.rest.img.blocks) + rest_blocks), dim3(256), 0, st, pp);
        HALO_TRY(halo_launch_status());
        if (images_by_pack)
            HALO_TRY(halo_gemm_bf16x3_tiled(img_in, img_w, T * B, 4 * H, in_dim, l0.gates, 4 * H, b_ih[0], b_hh[0], 0, nullptr, st));
        ctx.packT_reserve = reserve;
        ctx.bwdflags_reserve = reserve;
        for (int m = 0; m < 3; ++m) ctx.packT_w[m] = pa.w[m];
    } else {
        // forward only.  Weights the caller declared unchanged (halo_set_lstm_weights_stamp) keep the images the previous call packed into
        // this reserve: the launch then only packs the initial states and zeroes the epoch words
        const int dims[5] = {T, B, in0, H, L};
        const bool keep = ctx.lstm_weights_stamp != 0 && lo == 0 && ctx.packF_stamp == ctx.lstm_weights_stamp && ctx.packF_reserve == reserve &&
                          ctx.packF_w[0] == pa.w[0] && ctx.packF_w[1] == pa.w[1] && ctx.packF_w[2] == pa.w[2] &&
                          !memcmp(ctx.packF_dims, dims, sizeof(dims));
        if (keep) pa.w_units = 0;
        if (images_by_pack) {
            image_jobs_add(pa.img, in, T * B, in_dim, in_dim, 0, img_in);
            image_jobs_add(pa.img, w_ih[0], 4 * H, in_dim, in_dim, 0, img_w);
        }
        hipLaunchKernelGGL(persist2_prologue_kernel<0>, dim3((unsigned)pa.img.blocks + pack_grid((size_t)(3 * pa.w_units + 2 * pa.s_units + pa.zero_units))),
                           dim3(256), 0, st, pa);
        HALO_TRY(halo_launch_status());
        if (images_by_pack)
            HALO_TRY(halo_gemm_bf16x3_tiled(img_in, img_w, T * B, 4 * H, in_dim, l0.gates, 4 * H, b_ih[0], b_hh[0], 0, nullptr, st));
        ctx.packF_stamp = lo == 0 ? ctx.lstm_weights_stamp : 0;
        ctx.packF_reserve = reserve;
        for (int m = 0; m < 3; ++m) ctx.packF_w[m] = pa.w[m];
        memcpy(ctx.packF_dims, dims, sizeof(dims));
    }
    Persist2Fwd a;
    a.wp0 = (const char *)wp0; a.wp1 = (const char *)wp1; a.wpi = (const char *)wpi;
    a.hp0 = (char *)l0.hp; a.hp1 = (char *)l1.hp;
    a.xp = p_drop > 0.f ? (char *)fused_yp(extra, lo, T, B, H, L) : nullptr;
    a.gates0 = l0.gates; a.h0 = l0.h; a.c0 = l0.c;
    a.gates1 = l1.gates; a.h1 = l1.h; a.c1 = l1.c;
    a.b_ih1 = b_ih[1]; a.b_hh1 = b_hh[1];
    a.ydrop = p_drop > 0.f ? l0.ydrop : nullptr;
    a.y = y; a.y_stride_t = y_stride_t; a.y_stride_b = y_stride_b; a.y_mode = y ? (y_relu ? 2 : 1) : 0;
    a.drop = make_dropout(p_drop, seed, HALO_STREAM_LSTM_LAYER0 + (uint32_t)lo, offset, offset_dev);
    a.img_hT0 = emit_hT0; a.img_hT1 = emit_hT1; a.img_xT1 = emit_xT1;
    a.flags = flags;
    a.stamps = halo_lstm_persist_stamp_buffer();       // diagnostic: [blocks][T + 2][16] here
    a.T = T; a.B = B; a.H = H;
    chain_begin(st);
    HALO_TRY(halo_lstm_persist2_fwd(a, st));
    chain_end(st, 0, 1, "lstm_persist2_fwd_kernel");
    for (int l = 0; l < 2; ++l) {
        const LayerBufs lb = l ? l1 : l0;
        if (hn) HALO_TRY(copy_d2d(hn + (size_t)(lo + l) * BH, lb.h + (size_t)T * BH, BH, st));
        if (cn) HALO_TRY(copy_d2d(cn + (size_t)(lo + l) * BH, lb.c + (size_t)T * BH, BH, st));
    }
    return HALO_OK;
}


// the two-layer persistent backward (lstm_persist2.hip) keeps both layers' gate-gradient images alive at once: appended to the
// workspace are layer 1's bias partials [ceil(B/16)][4H], its packed images [T] and its dG^T GEMM operand image
inline size_t bwd_p2_offset(int T, int B, int in0, int H, int L) {
    const size_t n = bwd_flags_offset(T, B, in0, H, L) + PERSIST_FLAG_BYTES + (size_t)((B + 15) / 16) * 4 * H * sizeof(float);
    return (n + 255) & ~(size_t)255;
}
inline size_t bwd_p2_bytes(int T, int B, int H) {
    return (((size_t)((B + 15) / 16) * 4 * H * sizeof(float) + 255) & ~(size_t)255) + (size_t)T * bt16(B) * 4 * H * sizeof(float) +
           halo_tiled_image_bytes(4 * H, T * B);
}

}  // namespace

extern "C" {

int halo_lstm_chain_events(void *ev_begin, void *ev_end) {
    halo_ctx_cur().chain_ev0 = (hipEvent_t)ev_begin;
    halo_ctx_cur().chain_ev1 = (hipEvent_t)ev_end;
    return HALO_OK;
}

int halo_lstm_chain_info(int backward, int *launches, char *kernel, int kernel_len) {
    const ChainInfo &ci = g_chain_info[backward ? 1 : 0];
    if (launches) *launches = ci.launches;
    if (kernel && kernel_len > 0) {
        strncpy(kernel, ci.kernel, (size_t)kernel_len - 1);
        kernel[kernel_len - 1] = 0;
    }
    return HALO_OK;
}

size_t halo_lstm_reserve_bytes(int T, int B, int in0, int H, int L) {
    if (T <= 0 || B <= 0 || in0 <= 0 || H <= 0 || L <= 0) return 0;
    // L >= 2: behind the epoch words, what the two-layer forward (the stack's top pair) leaves for the backward of the same step
    return reserve_flags_offset(T, B, in0, H, L) + PERSIST_FLAG_BYTES + (L >= 2 ? reserve_p2_bytes(T, B, in0, H) : 0);
}

int halo_set_lstm_persistent(int on) {
    halo_lstm_persist_enable(on);
    return HALO_OK;
}

int halo_set_lstm_persistent_images(int on) {
    halo_ctx_cur().persist_emit = on ? 1 : 0;
    return HALO_OK;
}


int halo_lstm_persistent_eligible(int B, int H) { return halo_lstm_persist_ok(B, H) && use_x3(H) ? 1 : 0; }

int halo_set_lstm_persistent2(int on) {
    halo_lstm_persist2_enable(on);
    return HALO_OK;
}
int halo_set_lstm_bwd_mid_event(void *event) {
    halo_ctx_cur().bwd_mid_event = (hipEvent_t)event;
    halo_ctx_cur().bwd_mid_recorded = 0;
    return HALO_OK;
}
int halo_lstm_bwd_mid_event_recorded(void) { return halo_ctx_cur().bwd_mid_recorded; }
int halo_set_lstm_interleave(int on) {
    halo_lstm_interleave_enable(on);
    return HALO_OK;
}
int halo_set_lstm_weights_stamp(uint64_t stamp) {
    halo_ctx_cur().lstm_weights_stamp = stamp;
    return HALO_OK;
}

int halo_set_grad_sumsq(float *partials, int capacity) {
    HaloCtx &ctx = halo_ctx_cur();
    if (partials && capacity <= 0) return HALO_EINVAL;
    ctx.grad_sumsq = partials; ctx.grad_sumsq_cap = partials ? capacity : 0; ctx.grad_sumsq_n = 0; ctx.grad_sumsq_cover = 0;
    return HALO_OK;
}
int halo_grad_sumsq_state(int *count, unsigned *covered) {
    const HaloCtx &ctx = halo_ctx_cur();
    if (count) *count = ctx.grad_sumsq_n;
    if (covered) *covered = ctx.grad_sumsq_cover;
    return HALO_OK;
}
int halo_set_lstm_dx_slabs(int n) {
    if (n < 1 || n > 64) return HALO_EINVAL;
    halo_ctx_cur().lstm_dx_slabs = n;
    return HALO_OK;
}
int halo_lstm_dx_slabs_left(void) { return halo_ctx_cur().lstm_dx_slabs_left; }
int halo_set_lstm_expect_backward(int on) {
    halo_ctx_cur().lstm_expect_backward = on ? 1 : 0;
    return HALO_OK;
}
int halo_lstm_persistent2_eligible(int T, int B, int H, int L) {      /* L >= 2: the stack's top two layers */ return !fused_ok(H, L) && use_x3(H) && halo_lstm_persist2_ok(T, B, H, L) ? 1 : 0; }

size_t halo_lstm_status_offset(int backward, int T, int B, int in0, int H, int L) {
    if (T <= 0 || B <= 0 || in0 <= 0 || H <= 0 || L <= 0) return 0;
    return backward ? bwd_flags_offset(T, B, in0, H, L) : reserve_flags_offset(T, B, in0, H, L);
}

size_t halo_lstm_bwd_workspace_bytes(int T, int B, int in0, int H, int L) {
    if (T <= 0 || B <= 0 || H <= 0 || L <= 0) return 0;
    if (L >= 2) return bwd_p2_offset(T, B, in0, H, L) + bwd_p2_bytes(T, B, H);
    // packed W_hh^T [H,4H] + dc carry [B,H] + gradient w.r.t. a layer's input [T,B,H] + packed gate-gradient images
    // [BT16, 4H] (one per time step) + tiled images for the batched gradient GEMMs + epoch words
    // ... + the batch-tile partials of a layer's bias gradient [ceil(B/16)][4H] (written by the persistent backward)
    return bwd_flags_offset(T, B, in0, H, L) + PERSIST_FLAG_BYTES + (size_t)((B + 15) / 16) * 4 * H * sizeof(float);
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
    // what an earlier forward left for "its" backward (transposed weight images, operand images) is forgotten with every new forward,
    // whichever path it takes: the bookkeeping then always describes the most recent forward, and a recycled reserve address cannot match
    halo_ctx_cur().packT_reserve = nullptr;
    halo_ctx_cur().emitT_reserve = nullptr;
    float *wp = reserve;
    const bool x3 = use_x3(H);
    const int kin = in0 > H ? in0 : H;
    char *img_in = (char *)(reserve + (size_t)4 * H * H + (size_t)L * layer_floats(T, B, H));
    char *img_w = img_in + halo_tiled_image_bytes(T * B, kin);
    for (int l = 0; l < L; ++l) HALO_CHECK_ARG(w_ih[l] && w_hh[l] && b_ih[l] && b_hh[l]);
    // bf16: the top two layers of the stack run as ONE persistent launch (lstm_persist2.hip); the layers below them one launch each
    const bool pair = !fused_ok(H, L) && x3 && halo_lstm_persist2_ok(T, B, H, L);
    const int n_single = pair ? L - 2 : L;
    if (fused_ok(H, L)) {
        float *extra = (float *)(img_w + halo_tiled_image_bytes(4 * H, kin));
        return lstm_fwd_fused(x, w_ih, w_hh, b_ih, b_hh, h0, c0, y, y_stride_t, y_stride_b, y_relu, hn, cn, reserve, extra,
                              img_in, img_w, T, B, in0, H, L, p_drop, seed, offset, offset_dev, st);
    }
    for (int l = 0; l < L; ++l) {
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
        if (pair && l == n_single) {
            float *extra = (float *)(img_w + halo_tiled_image_bytes(4 * H, kin));
            return lstm_fwd_persist2(in, in_dim, l, w_ih, w_hh, b_ih, b_hh, h0, c0, y, y_stride_t, y_stride_b, y_relu, hn, cn, reserve, extra,
                                     img_in, img_w, T, B, in0, H, L, p_drop, seed, offset, offset_dev, st);
        }
        // gates[T*B, 4H] = in[T*B, in_dim] * W_ih^T + b_ih + b_hh
        if (halo_math_mode() != HALO_MATH_F32 && in_dim >= 64) {
            const HaloPrepJob jobs[2] = {{0, in, T * B, in_dim, in_dim, img_in, nullptr}, {0, w_ih[l], 4 * H, in_dim, in_dim, img_w, nullptr}};
            HALO_TRY(halo_prep_jobs(jobs, 2, st));                    // both operand images in one launch
            HALO_TRY(halo_gemm_bf16x3_tiled(img_in, img_w, T * B, 4 * H, in_dim, lb.gates, 4 * H, b_ih[l], b_hh[l], 0,
                                            nullptr, st));
        } else {
            HALO_TRY(halo_gemm_f32(1, 1, T * B, 4 * H, in_dim, in, in_dim, w_ih[l], in_dim, lb.gates, 4 * H, b_ih[l],
                                   b_hh[l], 0, 0.f, 0, 0, 0, nullptr, stream));
        }
        const float *h0l = h0 ? h0 + (size_t)l * BH : nullptr;
        const bool persist = x3 && halo_lstm_persist_ok(B, H) && halo_lstm_persist_fits(T, B, H);
        unsigned *flags = (unsigned *)((char *)reserve + reserve_flags_offset(T, B, in0, H, L));
        if (persist) {
            // one launch: packed W_hh, packed + row-major initial state, zeroed epoch words
            PrologueArgs pa;
            pa.w = w_hh[l]; pa.wdst = wp; pa.w_units = (long)(H / 16) * 4 * (H / 32) * 64; pa.wK = H;
            pa.h0 = h0l; pa.hp0 = lb.hp; pa.h_rm = lb.h; pa.c0 = c0 ? c0 + (size_t)l * BH : nullptr; pa.c_rm = lb.c;
            pa.s_units = (long)((B + 15) / 16) * (H / 32) * 64;
            pa.zero = flags; pa.zero_units = (long)(PERSIST_FLAG_BYTES / 16);
            pa.zero2 = nullptr; pa.zero2_units = 0;
            pa.H = H; pa.B = B;
            hipLaunchKernelGGL(persist_prologue_kernel<0>, dim3(pack_grid((size_t)(pa.w_units + pa.s_units + pa.zero_units))), dim3(256), 0, st, pa);
            HALO_TRY(halo_launch_status());
        } else {
            HALO_TRY(launch_pack<0>(w_hh[l], wp, H, B, H, (H / 16) * 4, x3, st));
            // one launch: packed h_{-1}, row-major h_{-1} and c_{-1} (zeros when no initial state is given)
            HALO_TRY(launch_pack<2>(h0l, lb.hp, H, B, H, (B + 15) / 16, x3, st, lb.h, c0 ? c0 + (size_t)l * BH : nullptr, lb.c));
        }
        const DropoutCfg dc = make_dropout(drop_out ? p_drop : 0.f, seed, HALO_STREAM_LSTM_LAYER0 + (uint32_t)l, offset, offset_dev);
        if (persist) {
            // ONE launch for all T steps: W_hh stays in registers, h_t is handed between workgroups (lstm_persist.hip)
            PersistFwd a;
            a.wp = (const char *)wp;
            a.hp = (char *)lb.hp;
            a.gates = lb.gates; a.h = lb.h; a.c = lb.c;
            a.T = T; a.B = B; a.H = H;
            a.drop = dc;
            if (last && y) { a.y = y; a.y_stride_t = y_stride_t; a.y_stride_b = y_stride_b; a.y_mode = y_relu ? Y_RELU : Y_PLAIN; }
            else if (drop_out) { a.y = lb.ydrop; a.y_stride_t = (long)BH; a.y_stride_b = H; a.y_mode = Y_DROPOUT; }
            else { a.y = nullptr; a.y_stride_t = 0; a.y_stride_b = 0; a.y_mode = Y_NONE; }
            a.flags = flags;
            a.stamps = halo_lstm_persist_stamp_buffer();
            chain_begin(st);
            HALO_TRY(halo_lstm_persist_fwd(a, st));
            chain_end(st, 0, 1, "lstm_persist_fwd_kernel");
            if (hn) HALO_TRY(copy_d2d(hn + (size_t)l * BH, lb.h + (size_t)T * BH, BH, st));
            if (cn) HALO_TRY(copy_d2d(cn + (size_t)l * BH, lb.c + (size_t)T * BH, BH, st));
            continue;
        }
        chain_begin(st);
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
            HALO_TRY(launch_step_fwd(a, x3, st));
        }
        chain_end(st, 0, T, "lstm_step_fwd_kernel");
        if (hn) HALO_TRY(copy_d2d(hn + (size_t)l * BH, lb.h + (size_t)T * BH, BH, st));
        if (cn) HALO_TRY(copy_d2d(cn + (size_t)l * BH, lb.c + (size_t)T * BH, BH, st));
    }
    return HALO_OK;
}

int halo_lstm_bwd(const float *x, const float *const *w_ih, const float *const *w_hh, const float *dy,
                  long y_stride_t, long y_stride_b, int y_relu, const float *dhn, const float *dcn, float *reserve,
                  float *workspace, float *dx, float *const *dw_ih, float *const *dw_hh, float *const *db_ih,
                  float *const *db_hh, int T, int B, int in0, int H, int L, int layer_begin, int layer_end,
                  float p_drop, uint64_t seed, uint32_t offset, const uint32_t *offset_dev, halo_stream_t stream) {
    HALO_CHECK_ARG(x && w_ih && w_hh && reserve && workspace && dw_ih && dw_hh && db_ih && db_hh);
    HALO_CHECK_ARG(0 <= layer_begin && layer_begin < layer_end && layer_end <= L);
    HALO_CHECK_ARG(layer_end < L || dy || dhn || dcn);
    HALO_CHECK_ARG(T > 0 && B > 0 && in0 > 0 && H > 0 && L > 0);
    if (H % 16 != 0) return HALO_ENOTSUP;
    halo_ctx_cur().lstm_dx_slabs_left = 1;
    hipStream_t st = (hipStream_t)stream;
    const size_t BH = (size_t)B * H, PG = bt16(B) * 4 * H;
    const bool x3 = use_x3(H);
    // The weight-gradient GEMMs run on the call's own stream.  (Forking them onto a side stream beside the next layer's step chain
    // was measured slower -- the chain is fetch-bound and any co-runner stretches it, DESIGN.md section 3 -- and shared the
    // operand-image workspace across layers without a per-layer join; the option is gone.)
    hipStream_t side = st;
    float *wpT = workspace;
    float *dcarry = wpT + (size_t)H * 4 * H;
    float *din = dcarry + BH;          // [T,B,H] gradient w.r.t. the current layer's input
    float *dgp = din + (size_t)T * BH; // packed gate-gradient images: ping-pong for the step launches, one per step otherwise
    const int kin = in0 > H ? in0 : H;
    char *img_gT = (char *)(dgp + (size_t)bwd_dg_images(T) * PG);
    char *img_g = img_gT + halo_tiled_image_bytes(4 * H, T * B);
    char *img_hT = img_g + halo_tiled_image_bytes(T * B, 4 * H);
    char *img_inT = img_hT + halo_tiled_image_bytes(kin, T * B);
    char *img_wT = img_inT + halo_tiled_image_bytes(kin, T * B);
    // layer-diagonal fused step chain: all layers at once, on the call that contains the top layer
    const bool fused = fused_ok(H, L);
    if (fused && layer_end == L) {
        float *extra = (float *)(img_wT + halo_tiled_image_bytes(kin, 4 * H));
        HALO_TRY(lstm_bwd_fused_chain(w_ih, w_hh, dy, y_stride_t, y_stride_b, y_relu, dhn, dcn, reserve, extra, T, B, H, L,
                                      p_drop, seed, offset, offset_dev, st));
    }
    int top_end = layer_end;          // layers [layer_begin, top_end) are left for the per-layer loop below
    if (!fused && x3 && L >= 2 && layer_begin <= L - 2 && layer_end == L && halo_lstm_persist2_ok(T, B, H, L)) {
        // the stack's top two layers' chains in ONE persistent launch (lstm_persist2.hip), the upper layer's input gradient formed inside it
        const int lo = L - 2, hi = L - 1;
        for (int l = lo; l <= hi; ++l) HALO_CHECK_ARG(w_ih[l] && w_hh[l] && dw_ih[l] && dw_hh[l] && db_ih[l] && db_hh[l]);
        const LayerBufs l0 = layer_bufs(reserve, lo, T, B, H), l1 = layer_bufs(reserve, hi, T, B, H);
        const float *in_lo = lo == 0 ? x : (p_drop > 0.f ? layer_bufs(reserve, lo - 1, T, B, H).ydrop : layer_bufs(reserve, lo - 1, T, B, H).h + BH);
        const int in_lo_dim = lo == 0 ? in0 : H;
        float *extra = (float *)(img_wT + halo_tiled_image_bytes(kin, 4 * H));
        float *wpT0 = wpT, *wpT1 = extra, *wpTi = extra + (size_t)4 * H * H, *dcarry1 = extra + (size_t)8 * H * H;      // (or the reserve's: below)
        char *flag_base = (char *)workspace + bwd_flags_offset(T, B, in0, H, L);
        float *bias_part0 = (float *)(flag_base + PERSIST_FLAG_BYTES);
        char *p2 = (char *)workspace + bwd_p2_offset(T, B, in0, H, L);
        float *bias_part1 = (float *)p2;
        float *dgp1 = (float *)(p2 + (((size_t)((B + 15) / 16) * 4 * H * sizeof(float) + 255) & ~(size_t)255));
        char *img_gT1 = (char *)(dgp1 + (size_t)T * PG);
        const bool can_emit = persist_emit_enabled() && B % 32 == 0;
        const bool emit0 = can_emit && in_lo_dim >= 64, emit1 = can_emit;
        const bool need_din = lo > 0 || dx != nullptr;         // the gradient w.r.t. layer lo's input: feeds layer lo - 1, or the caller
        float *din_out = lo > 0 ? din : dx;
        // the forward of this step may have left the three transposed images in the reserve (one read of the weights for all six)
        HaloCtx &ctx = halo_ctx_cur();
        const bool have_T = ctx.packT_reserve == reserve && ctx.packT_w[0] == w_hh[lo] && ctx.packT_w[1] == w_hh[hi] && ctx.packT_w[2] == w_ih[hi];
        // ... and have zeroed this launch's epoch words there: nothing is left for a prologue launch (the padding rows of the gate-gradient
        // row image need no zeroing: they only reach output rows past T*B, which no product stores)
        const bool clean = have_T && ctx.bwdflags_reserve == reserve;
        ctx.bwdflags_reserve = nullptr;                        // one backward per forward finds them clean
        if (clean) flag_base = (char *)reserve_bwd_flags(reserve, T, B, in0, H, L);
        if (have_T) {
            float *wT = (float *)((char *)reserve + reserve_flags_offset(T, B, in0, H, L) + PERSIST_FLAG_BYTES);
            wpT0 = wT; wpT1 = wT + (size_t)4 * H * H; wpTi = wT + (size_t)8 * H * H;
        }
        Prologue2Args pa;
        pa.img.n = 0; pa.img.blocks = 0; pa.img.with_lo = 0;
        pa.w[0] = w_hh[lo]; pa.wdst[0] = wpT0; pa.w[1] = w_hh[hi]; pa.wdst[1] = wpT1; pa.w[2] = w_ih[hi]; pa.wdst[2] = wpTi;
        pa.w_units = have_T ? 0 : (long)(H / 16) * (4 * H / 32) * 64; pa.wK = 4 * H;
        for (int l = 0; l < 2; ++l) { pa.h0[l] = nullptr; pa.hp0[l] = nullptr; pa.h_rm[l] = nullptr; pa.c0[l] = nullptr; pa.c_rm[l] = nullptr; }
        pa.s_units = 0;
        pa.zero = (unsigned *)flag_base; pa.zero_units = (long)(PERSIST_FLAG_BYTES / 16);
        pa.zero2 = nullptr; pa.zero2_units = 0;
        if (emit0 && need_din && (T * B) % 128 != 0) {   // the padded last row tile of the lower layer's dG row image
            const long tile_bytes = (long)(4 * H / 32) * 16384;
            pa.zero2 = (unsigned *)(img_g + (long)((T * B) / 128) * tile_bytes); pa.zero2_units = tile_bytes / 16;
        }
        pa.H = H; pa.B = B;
        if (!clean) {
            hipLaunchKernelGGL(persist2_prologue_kernel<1>, dim3(pack_grid((size_t)(3 * pa.w_units + pa.zero_units + pa.zero2_units))), dim3(256), 0,
                               st, pa);
            HALO_TRY(halo_launch_status());
        }
        Persist2Bwd a;
        a.wpT0 = (const char *)wpT0; a.wpT1 = (const char *)wpT1; a.wpTi = (const char *)wpTi;
        a.dgp0 = (char *)dgp; a.dgp1 = (char *)dgp1;
        a.gates0 = l0.gates; a.gates1 = l1.gates; a.c0 = l0.c; a.c1 = l1.c;
        a.dc0 = dcarry; a.dc1 = dcarry1;
        a.dy = dy; a.dy_stride_t = y_stride_t; a.dy_stride_b = y_stride_b; a.dy_relu = y_relu;
        a.dhinit0 = dhn ? dhn + (size_t)lo * BH : nullptr; a.dcinit0 = dcn ? dcn + (size_t)lo * BH : nullptr;
        a.dhinit1 = dhn ? dhn + (size_t)hi * BH : nullptr; a.dcinit1 = dcn ? dcn + (size_t)hi * BH : nullptr;
        a.drop = make_dropout(p_drop, seed, HALO_STREAM_LSTM_LAYER0 + (uint32_t)lo, offset, offset_dev);
        a.flags = (unsigned *)flag_base;
        a.abort_word = (unsigned *)((char *)workspace + bwd_flags_offset(T, B, in0, H, L));      // (== flags unless the words live in the reserve)
        a.stamps = halo_lstm_persist_stamp_buffer();       // diagnostic: [blocks][T + 1][16] here
        a.img_rows0 = emit0 && need_din ? img_g : nullptr;
        a.img_cols0 = emit0 ? img_gT : nullptr;
        a.img_cols1 = emit1 ? img_gT1 : nullptr;
        a.bias_part0 = emit0 ? bias_part0 : nullptr;
        a.bias_part1 = emit1 ? bias_part1 : nullptr;
        // a layer whose operand images and bias partials the chain writes itself has no reader left for its fp32 gate gradients
        a.skip_dg0 = !keep_dg_switch() && emit0 && (!need_din || a.img_rows0);
        a.skip_dg1 = !keep_dg_switch() && emit1;
        a.T = T; a.B = B; a.H = H;
        chain_begin(st);
        HALO_TRY(halo_lstm_persist2_bwd(a, st));
        chain_end(st, 1, 1, "lstm_persist2_bwd_kernel");
        if (ctx.emitT_reserve == reserve && emit0 && emit1 && pair_dw_enabled()) {
            // the forward launch of this step left h_prev^T (both layers) and dropout(h_lo)^T as operand images in the reserve: ONE operand
            // launch for what is left (the bias sums, W_ih_lo^T, in^T), then the two paired weight-gradient launches and the input gradient
            char *imgs = reserve_p2_images(reserve, T, B, in0, H, L);
            const size_t hb = halo_tiled_image_bytes(H, T * B), ib = halo_tiled_image_bytes(kin, T * B);
            char *hT0 = imgs, *inT0 = imgs + hb, *hT1 = imgs + hb + ib;
            HaloPrepJob jobs[4];
            int nj = 0;
            // the bias gradients' batch-tile partials: summed here, or queued for a later launch's tail blocks (small_jobs.h)
            // (queued, they also leave the squared-norm partials of what they write when the caller collects those: halo_set_grad_sumsq)
            for (int l = 1; l >= 0; --l) {
                HaloSmallJob j = {};
                j.kind = 2; j.n = (B + 15) / 16; j.len = 4 * H; j.a = l ? bias_part1 : bias_part0; j.o1 = db_ih[l ? hi : lo]; j.o2 = db_hh[l ? hi : lo];
                const int jb = halo_small_job_blocks(j);
                const bool collect = ctx.grad_sumsq && ctx.defer_small_jobs && ctx.grad_sumsq_n + jb <= ctx.grad_sumsq_cap;
                if (collect) j.ss = ctx.grad_sumsq + ctx.grad_sumsq_n;
                if (halo_defer_small_job(j)) {
                    if (collect) { ctx.grad_sumsq_n += jb; ctx.grad_sumsq_cover |= l ? 4u : 8u; }
                } else jobs[nj++] = {3, j.a, (B + 15) / 16, 4 * H, 4 * H, j.o1, j.o2};
            }
            // W_ih_lo^T and in^T: the forward's packing launch of this step may have written them (same reserve, same weights: have_T)
            const bool have_fwdT = have_T && ctx.fwdT_reserve == reserve && ctx.fwdT_src[0] == in_lo && ctx.fwdT_src[1] == w_ih[lo];
            if (have_fwdT) img_wT = reserve_p2_wT(reserve, T, B, in0, H, L);
            else {
                if (need_din) jobs[nj++] = {1, w_ih[lo], in_lo_dim, 4 * H, in_lo_dim, img_wT, nullptr};
                jobs[nj++] = {1, in_lo, in_lo_dim, T * B, in_lo_dim, inT0, nullptr};
            }
            HALO_TRY(halo_prep_jobs(jobs, nj, st));
            // the upper layer's two weight gradients; the K-slices of the caller's input gradient ride in the same launch where it has room
            int carried = 0;
            const bool dx_slices = need_din && lo == 0 && ctx.lstm_dx_slabs > 1;
            // (and each launch leaves the squared-norm partials of the gradients it stores when the caller collects them)
            auto sumsq_slot = [&](int tiles) { return ctx.grad_sumsq && ctx.grad_sumsq_n + tiles <= ctx.grad_sumsq_cap ? ctx.grad_sumsq + ctx.grad_sumsq_n : nullptr; };
            int parts = 0;
            // both layers' weight gradients (and the input gradient's K-slices) as ONE launch of 256 x 256 tiles (gemm256.hip): 208 + 48
            // workgroups at H = 1024, one per CU, where the 128 x 128-tile kernel below runs two launches of 512 + 288 (+ 88 carried)
            HaloG256Problem gp[3] = {};
            gp[0].A = img_gT1; gp[0].B = hT1; gp[0].M = 4 * H; gp[0].N = 2 * H; gp[0].K = T * B;
            gp[0].C = dw_hh[hi]; gp[0].ldc = H; gp[0].n_split = H; gp[0].C2 = dw_ih[hi]; gp[0].ldc2 = H; gp[0].kslices = 1;
            gp[1].A = img_gT; gp[1].B = hT0; gp[1].M = 4 * H; gp[1].N = H + in_lo_dim; gp[1].K = T * B;
            gp[1].C = dw_hh[lo]; gp[1].ldc = H; gp[1].n_split = H; gp[1].C2 = dw_ih[lo]; gp[1].ldc2 = in_lo_dim; gp[1].kslices = 1;
            gp[2].A = img_g; gp[2].B = img_wT; gp[2].M = T * B; gp[2].N = in_lo_dim; gp[2].K = 4 * H;
            gp[2].C = din_out; gp[2].ldc = in_lo_dim; gp[2].n_split = in_lo_dim; gp[2].kslices = ctx.lstm_dx_slabs; gp[2].slab_stride = (long)T * B * in_lo_dim;
            const int ngp = dx_slices ? 3 : 2;
            bool all256 = halo_gemm256_enabled() != 0;
            for (int i = 0; i < ngp && all256; ++i) all256 = halo_gemm256_fits(gp[i]) != 0;
            if (all256) {
                const int t0 = ((4 * H + 255) / 256) * ((2 * H + 255) / 256), t1 = ((4 * H + 255) / 256) * ((H + in_lo_dim + 255) / 256);
                float *slot2 = sumsq_slot(t0 + t1);
                gp[0].sumsq = slot2; gp[1].sumsq = slot2 ? slot2 + t0 : nullptr;
                if (ctx.bwd_mid_event) {
                    // a data-parallel caller starts reducing the TOP layer's gradients behind this event (halo_set_lstm_bwd_mid_event): its
                    // product goes first, alone (128 tiles, ~23 us), the lower layer's and the input gradient's behind it -- 3 us more
                    // compute than the one launch, and the exchange starts 20 us earlier
                    HALO_TRY(halo_gemm256_launch(gp, 1, st));
                    if (hipEventRecord(ctx.bwd_mid_event, st) == hipSuccess) ++ctx.bwd_mid_recorded;
                    HALO_TRY(halo_gemm256_launch(gp + 1, ngp - 1, st));
                } else {
                    HALO_TRY(halo_gemm256_launch(gp, ngp, st));
                }
                if (slot2) { ctx.grad_sumsq_n += gp[0].tiles + gp[1].tiles; ctx.grad_sumsq_cover |= 3u; }
                if (dx_slices) ctx.lstm_dx_slabs_left = gp[2].kslices;
                else if (need_din) {
                    const DropoutCfg ddrop = make_dropout(lo > 0 ? p_drop : 0.f, seed, HALO_STREAM_LSTM_LAYER0 + (uint32_t)(lo > 0 ? lo - 1 : 0), offset,
                                                          offset_dev);
                    HALO_TRY(halo_gemm_bf16x3_tiled(img_g, img_wT, T * B, in_lo_dim, 4 * H, din_out, in_lo_dim, nullptr, nullptr, 0, &ddrop, st));
                }
            } else {
            float *slot = sumsq_slot(((4 * H + 63) / 64) * ((2 * H + 127) / 128));       // (room for the most workgroups the launch may use: half-height tiles)
            HALO_TRY(halo_gemm_bf16x3_tiled_nsplit_carry(img_gT1, hT1, 4 * H, 2 * H, T * B, dw_hh[hi], H, H, dw_ih[hi], H,
                                                         dx_slices ? img_g : nullptr, img_wT, T * B, in_lo_dim, 4 * H, din_out, ctx.lstm_dx_slabs,
                                                         &carried, slot, slot ? &parts : nullptr, st));
            if (slot) { ctx.grad_sumsq_n += parts; ctx.grad_sumsq_cover |= 1u; }
            // the top layer's weight gradients are final behind this point of the stream: a data-parallel caller starts their exchange here
            if (ctx.bwd_mid_event && hipEventRecord(ctx.bwd_mid_event, st) == hipSuccess) ++ctx.bwd_mid_recorded;
            if (carried) ctx.lstm_dx_slabs_left = carried;
            else
            if (need_din) {     // (masked by the dropout of the layer below's output, which this gradient flows into)
                const DropoutCfg ddrop = make_dropout(lo > 0 ? p_drop : 0.f, seed, HALO_STREAM_LSTM_LAYER0 + (uint32_t)(lo > 0 ? lo - 1 : 0), offset,
                                                      offset_dev);
                if (lo == 0 && ctx.lstm_dx_slabs > 1) {
                    // the caller's buffer has room for K-slices and its consumer adds them as it reads (halo_set_lstm_dx_slabs): no reduce launch
                    HALO_TRY(halo_gemm_bf16x3_tiled_slices(img_g, img_wT, T * B, in_lo_dim, 4 * H, din_out, ctx.lstm_dx_slabs, &ctx.lstm_dx_slabs_left, st));
                } else
                HALO_TRY(halo_gemm_bf16x3_tiled(img_g, img_wT, T * B, in_lo_dim, 4 * H, din_out, in_lo_dim, nullptr, nullptr, 0, &ddrop, st));
            }
            slot = sumsq_slot(((4 * H + 63) / 64) * ((H + in_lo_dim + 127) / 128));
            parts = 0;
            HALO_TRY(halo_gemm_bf16x3_tiled_nsplit_carry(img_gT, hT0, 4 * H, H + in_lo_dim, T * B, dw_hh[lo], H, H, dw_ih[lo], in_lo_dim, nullptr,
                                                         nullptr, 0, 0, 0, nullptr, 0, nullptr, slot, slot ? &parts : nullptr, st));
            if (slot) { ctx.grad_sumsq_n += parts; ctx.grad_sumsq_cover |= 2u; }
            }
        } else {
            HALO_TRY(lstm_bwd_layer_tail(x, w_ih, reserve, hi, in0, T, B, H, L, p_drop, seed, offset, offset_dev, false, emit1, img_g,
                                         emit1 ? img_gT1 : img_gT, img_hT, img_inT, img_wT, bias_part1, din, dx, dw_ih, dw_hh, db_ih, db_hh, false, st));
            if (ctx.bwd_mid_event && hipEventRecord(ctx.bwd_mid_event, st) == hipSuccess) ++ctx.bwd_mid_recorded;
            HALO_TRY(lstm_bwd_layer_tail(x, w_ih, reserve, lo, in0, T, B, H, L, p_drop, seed, offset, offset_dev, false, emit0, img_g, img_gT, img_hT,
                                         img_inT, img_wT, bias_part0, din, dx, dw_ih, dw_hh, db_ih, db_hh, need_din, st));
        }
        top_end = lo;                 // the layers below (if any) follow, one launch each, fed by din
        if (top_end <= layer_begin) return HALO_OK;
    }
    for (int l = top_end - 1; l >= layer_begin; --l) {
        HALO_CHECK_ARG(w_ih[l] && w_hh[l] && dw_ih[l] && dw_hh[l] && db_ih[l] && db_hh[l]);
        const LayerBufs lb = layer_bufs(reserve, l, T, B, H);
        const bool last = (l == L - 1);
        const bool persist = !fused && x3 && halo_lstm_persist_ok(B, H) && halo_lstm_persist_fits(T, B, H);
        // the chain writes the gate gradients' GEMM operand images itself (three-pass images: hi and lo parts) where its tiles map onto
        // whole 16-byte chunks of them: B % 32 == 0; what the operand-image launch below then no longer reads is the 4H-wide fp32 dG
        const int in_dim_l = l == 0 ? in0 : H;
        const bool need_din_l = (l > 0 && !fused) || (l == 0 && dx);
        const bool emit = persist && persist_emit_enabled() && halo_math_mode() != HALO_MATH_F32 && B % 32 == 0 && in_dim_l >= 64;
        if (persist) {
            PrologueArgs pa;                                  // packed W_hh^T and zeroed epoch words in one launch
            pa.w = w_hh[l]; pa.wdst = wpT; pa.w_units = (long)(H / 16) * (4 * H / 32) * 64; pa.wK = 4 * H;
            pa.h0 = nullptr; pa.hp0 = nullptr; pa.h_rm = nullptr; pa.c0 = nullptr; pa.c_rm = nullptr; pa.s_units = 0;
            pa.zero = (unsigned *)((char *)workspace + bwd_flags_offset(T, B, in0, H, L)); pa.zero_units = (long)(PERSIST_FLAG_BYTES / 16);
            pa.zero2 = nullptr; pa.zero2_units = 0;
            if (emit && need_din_l && (T * B) % 128 != 0) {   // the padded last row tile of dG's row image: cleared whole, the chain fills its real rows
                const long tile_bytes = (long)(4 * H / 32) * 16384;
                pa.zero2 = (unsigned *)(img_g + (long)((T * B) / 128) * tile_bytes); pa.zero2_units = tile_bytes / 16;
            }
            pa.H = H; pa.B = B;
            hipLaunchKernelGGL(persist_prologue_kernel<1>, dim3(pack_grid((size_t)(pa.w_units + pa.zero_units + pa.zero2_units))), dim3(256), 0, st, pa);
            HALO_TRY(halo_launch_status());
        } else if (!fused) {
            HALO_TRY(launch_pack<1>(w_hh[l], wpT, H, B, 4 * H, H / 16, x3, st));
        }
        if (persist) {
            PersistBwd a;
            a.wpT = (const char *)wpT;
            a.dgp = (char *)dgp;
            a.gates = lb.gates; a.c = lb.c; a.dc = dcarry;
            if (last) { a.dy = dy; a.dy_stride_t = y_stride_t; a.dy_stride_b = y_stride_b; a.dy_relu = y_relu; }
            else { a.dy = din; a.dy_stride_t = (long)BH; a.dy_stride_b = H; a.dy_relu = 0; }
            a.dhinit = dhn ? dhn + (size_t)l * BH : nullptr;
            a.dcinit = dcn ? dcn + (size_t)l * BH : nullptr;
            a.flags = (unsigned *)((char *)workspace + bwd_flags_offset(T, B, in0, H, L));
            a.stamps = nullptr;
            a.img_rows = emit && need_din_l ? img_g : nullptr;
            a.img_cols = emit ? img_gT : nullptr;
            a.bias_part = emit ? (float *)((char *)workspace + bwd_flags_offset(T, B, in0, H, L) + PERSIST_FLAG_BYTES) : nullptr;
            a.skip_dg = emit && (!need_din_l || a.img_rows) && !keep_dg_switch();      // (as the two-layer launch: no reader left)
            a.T = T; a.B = B; a.H = H;
            chain_begin(st);
            HALO_TRY(halo_lstm_persist_bwd(a, st));
            chain_end(st, 1, 1, "lstm_persist_bwd_kernel");
        }
        if (!fused && !persist) chain_begin(st);
        for (int t = T - 1; t >= 0 && !fused && !persist; --t) {
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
            HALO_TRY(launch_step_bwd(a, x3, st));
        }
        if (!fused && !persist) chain_end(st, 1, T, "lstm_step_bwd_kernel");
        // parameter gradients over all frames at once
        HALO_TRY(lstm_bwd_layer_tail(x, w_ih, reserve, l, in0, T, B, H, L, p_drop, seed, offset, offset_dev, fused, emit, img_g, img_gT, img_hT,
                                     img_inT, img_wT, emit ? (const float *)((char *)workspace + bwd_flags_offset(T, B, in0, H, L) + PERSIST_FLAG_BYTES) : nullptr,
                                     din, dx, dw_ih, dw_hh, db_ih, db_hh, (l > 0 && !fused) || (l == 0 && dx), st));
    }
    return HALO_OK;
}

}  // extern "C"
