// The weight-resident persistent LSTM recurrence of lstm_persist.hip with 32 batch rows per workgroup (single-pass bf16 arithmetic only):
// workgroup (jt, bt) owns hidden units [16 jt, +16) of batch rows [32 bt, +32) as TWO 16-row sub-tiles that share the register-resident
// weights, so (H/16) x ceil(B/32) workgroups cover a batch of up to 128 at H = 1024 with one workgroup per CU (the 16-row kernels stop at 64).
// In bf16 the weights take 64 of a wave's registers, which leaves room for the second sub-tile's fragments and accumulators; the split-bf16
// kernels have no such room (256 of 256 registers in the backward), so `bf16x3` batches beyond 64 keep the step-launch chain.
// Same protocol, image layouts and outputs as lstm_persist.hip: the exchange group is the 32-row tile (its H/16 workgroups), the packed
// images stay indexed by 16-row batch tiles (2 bt, 2 bt + 1), one wave instruction publishes both sub-tiles' pieces.
#include <stdlib.h>
#include "halo_common.h"
#include "halo_internal.h"
#include "lstm_persist.h"
#include "lstm_persist_dev.h"

namespace {

// ================================================================================================================
// forward.  KBW: k-blocks (32 deep) per wave = H / 256.
// ================================================================================================================
template <int KBW>
__global__ __launch_bounds__(512, 2) void lstm_persist_fwd32_kernel(const PersistFwd p) {
    __shared__ float red[2][NWAVE][4][256];              // partial gate sums of the 8 K-slices, per sub-tile
    __shared__ __attribute__((aligned(16))) float hbuf[2][16][16];
    __shared__ int s_abort;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = p.H, B = p.B, T = p.T;
    const int NJ = H / 16, NBT16 = (B + 15) / 16, NBT32 = (B + 31) / 32, nkb = H / 32;
    int jt, bt;
    map_block(blockIdx.x, gridDim.x, NJ, NBT32, jt, bt);
    const int j0 = jt * 16;
    const bool has1 = 2 * bt + 1 < NBT16;                // the second sub-tile exists (B = 48: the last workgroup row has one)

    bf16x8 wh[4][KBW];                                   // gates 0..3, k-blocks [wave*KBW, +KBW), hi halves
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int i = 0; i < KBW; ++i)
            wh[g][i] = *reinterpret_cast<const bf16x8 *>(p.wp + (((long)jt * 4 + g) * nkb + wave * KBW + i) * 2048 + lane * 16);

    const int ci = (tid & 255) >> 4, cj = tid & 15;      // cell threads: tid < 256 -> (batch row, hidden unit) of BOTH sub-tiles
    const int BH = B * H;
    int brow[2], e0[2];
    bool cell[2];
    float cst[2] = {0.f, 0.f}, gin[2][4];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        brow[s] = bt * 32 + s * 16 + ci;
        cell[s] = tid < 256 && brow[s] < B;
        e0[s] = brow[s] * H + j0 + cj;
#pragma unroll
        for (int g = 0; g < 4; ++g) gin[s][g] = cell[s] ? p.gates[brow[s] * 4 * H + g * H + j0 + cj] : 0.f;
        if (cell[s]) cst[s] = p.c[e0[s]];
    }
    const __amdgpu_buffer_rsrc_t hp_rsrc = make_rsrc(p.hp);
    const int my_replica = (xcc_id() + p.replica_shift) % PERSIST_REPLICAS;
    const unsigned *grp_flags = p.flags + my_replica * PERSIST_REPLICA_WORDS + PERSIST_FLAG_HEADER + bt * NJ;
    if (tid == 0) s_abort = 0;
    for (int t = 0; t < T; ++t) {
        bool ok = true;
        if (t > 0 && wave == 5) ok = poll_group(grp_flags, 0, NJ, (unsigned)t, lane, p.nap);
        if (!ok && lane == 0) {
            s_abort = 1;
            raise_abort(p.flags, p.status);
        }
        lds_barrier();                                                             // (A)
        if (s_abort) return;
        // this wave's fragments of both sub-tiles of image t (= h_{t-1})
        bf16x8 ah[2][KBW];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int img = ((t * NBT16 + 2 * bt + (s && has1 ? 1 : 0)) * nkb + wave * KBW) * 2048;   // (a missing sub-tile re-reads the first: unused)
#pragma unroll
            for (int i = 0; i < KBW; ++i) ah[s][i] = load_sc1_u(hp_rsrc, lane * 16, img + i * 2048);
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc[2][4];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[s][g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < KBW; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int s = 0; s < 2; ++s) acc[s][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[s][i], wh[g][i], acc[s][g], 0, 0, 0);
        {
            const int r = lane & 15, q = lane >> 4;
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) red[s][wave][g][(4 * q + e) * 16 + r] = acc[s][g][e];
        }
        lds_barrier();                                                             // (B)
        float gt[2][4], hv[2] = {0.f, 0.f};
        if (tid < 256) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int g = 0; g < 4; ++g) gt[s][g] = 0.f;
                if (cell[s]) {
                    float pre[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float sum = 0.f;
#pragma unroll
                        for (int k = 0; k < NWAVE; ++k) sum += red[s][k][g][tid];
                        pre[g] = sum + gin[s][g];
                    }
                    gt[s][0] = fast_sigmoid(pre[0]); gt[s][1] = fast_sigmoid(pre[1]); gt[s][2] = fast_tanh(pre[2]); gt[s][3] = fast_sigmoid(pre[3]);
                    cst[s] = gt[s][1] * cst[s] + gt[s][0] * gt[s][2];
                    hv[s] = gt[s][3] * fast_tanh(cst[s]);
                }
                hbuf[s][ci][cj] = hv[s];                 // rows >= B: zeros
            }
        }
        lds_barrier();                                                             // (C)
        if (wave == 4) {
            // lanes 0-31 publish sub-tile 0's hi piece of h_t, lanes 32-63 sub-tile 1's (image t + 1, batch tiles 2 bt and 2 bt + 1): this
            // tile is k-groups 2 (jt & 1), 2 (jt & 1) + 1 of k-block jt / 2
            const int s = lane >> 5, kg = (lane >> 4) & 1, row = lane & 15;
            bf16x8 hi;
#pragma unroll
            for (int e = 0; e < 8; ++e) hi[e] = (__bf16)hbuf[s][row][kg * 8 + e];
            const int dst = (((t + 1) * NBT16 + 2 * bt + s) * nkb + (jt >> 1)) * 2048 + (((jt & 1) * 2 + kg) * 16 + row) * 16;
            if (s == 0 || has1) store_sc1(hp_rsrc, dst, hi);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if ((int)blockIdx.x != p.mute) publish_epoch(p.flags, bt * NJ + jt, (unsigned)(t + 1), lane);
        }
        if (tid < 256) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                if (!cell[s]) continue;
                float *gp = p.gates + (t * B + brow[s]) * 4 * H + j0 + cj;
                gp[0] = gt[s][0]; gp[H] = gt[s][1]; gp[2 * H] = gt[s][2]; gp[3 * H] = gt[s][3];
                p.c[(t + 1) * BH + e0[s]] = cst[s];
                p.h[(t + 1) * BH + e0[s]] = hv[s];
                if (p.y_mode != Y_NONE) {
                    float v = hv[s];
                    if (p.y_mode == Y_RELU) v = fmaxf(v, 0.f);
                    else if (p.y_mode == Y_DROPOUT) v = v * dropout_mult(p.drop, (uint64_t)((long)t * BH + e0[s]));
                    p.y[(long)t * p.y_stride_t + (long)brow[s] * p.y_stride_b + j0 + cj] = v;
                }
                if (t + 1 < T) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) gin[s][g] = p.gates[((t + 1) * B + brow[s]) * 4 * H + g * H + j0 + cj];
                }
            }
        }
    }
}

// ================================================================================================================
// backward.  KC: chunks of 4 k-blocks per wave = (4H / 32 / 8) / 4 = H / 256.
// ================================================================================================================
template <int KC>
__global__ __launch_bounds__(512, 2) void lstm_persist_bwd32_kernel(const PersistBwd p) {
    constexpr int KBW = 4 * KC, CH = 4, NCH = 2 * KC, NBUF = 3;      // chunk c: sub-tile c / KC, k-blocks 4 (c % KC) ..; three register buffers
    __shared__ float red[2][NWAVE][256];
    __shared__ __attribute__((aligned(16))) float dgbuf[2][4][16][16];
    __shared__ int s_abort;
    __shared__ unsigned s_published;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = p.H, B = p.B, T = p.T, K = 4 * H;
    const int NJ = H / 16, NBT16 = (B + 15) / 16, NBT32 = (B + 31) / 32, nkb4 = K / 32;
    int jt, bt;
    map_block(blockIdx.x, gridDim.x, NJ, NBT32, jt, bt);
    const int j0 = jt * 16;
    const bool has1 = 2 * bt + 1 < NBT16;

    bf16x8 wh[KBW];                                      // columns j0..j0+15 of W_hh^T, k-blocks [wave*KBW, +KBW) of the 4H-deep contraction
#pragma unroll
    for (int i = 0; i < KBW; ++i) wh[i] = *reinterpret_cast<const bf16x8 *>(p.wpT + ((long)jt * nkb4 + wave * KBW + i) * 2048 + lane * 16);

    const int ci = (tid & 255) >> 4, cj = tid & 15;
    const int BH = B * H;
    int brow[2], e0[2];
    bool cell[2];
    float gv[2][4], cc[2] = {0.f, 0.f}, cprev[2] = {0.f, 0.f}, dyv[2] = {0.f, 0.f}, dcarry[2] = {0.f, 0.f}, dh0[2] = {0.f, 0.f}, bsum[2][4];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        brow[s] = bt * 32 + s * 16 + ci;
        cell[s] = tid < 256 && brow[s] < B;
        e0[s] = brow[s] * H + j0 + cj;
#pragma unroll
        for (int g = 0; g < 4; ++g) { gv[s][g] = 0.f; bsum[s][g] = 0.f; }
        if (cell[s]) {
            const int t = T - 1;
#pragma unroll
            for (int g = 0; g < 4; ++g) gv[s][g] = p.gates[(t * B + brow[s]) * K + g * H + j0 + cj];
            cc[s] = p.c[(t + 1) * BH + e0[s]];
            cprev[s] = p.c[t * BH + e0[s]];
            if (p.dy) dyv[s] = p.dy[(long)t * p.dy_stride_t + (long)brow[s] * p.dy_stride_b + j0 + cj];
            if (p.dcinit) dcarry[s] = p.dcinit[e0[s]];
            if (p.dhinit) dh0[s] = p.dhinit[e0[s]];
        }
    }
    const __amdgpu_buffer_rsrc_t dg_rsrc = make_rsrc(p.dgp);
    const int my_replica = (xcc_id() + p.replica_shift) % PERSIST_REPLICAS;
    const unsigned *grp_flags = p.flags + my_replica * PERSIST_REPLICA_WORDS + PERSIST_FLAG_HEADER + bt * NJ;
    if (tid == 0) { s_abort = 0; s_published = 0; }

    for (int s = 0; s < T; ++s) {
        const int t = T - 1 - s;
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        bool ok = true;
        if (s > 0 && wave == 7) ok = poll_group(grp_flags, 0, NJ, (unsigned)s, lane, p.nap);
        if (!ok && lane == 0) {
            s_abort = 1;
            raise_abort(p.flags, p.status);
        }
        lds_barrier();                                                             // (A)
        if (s_abort) return;
        if (s > 0) {
            // image t+1 (gate gradients of the step done before) of both sub-tiles, this wave's K-eighth
            const int img0 = (((t + 1) * NBT16 + 2 * bt) * nkb4 + wave * KBW) * 2048;
            const int img1 = has1 ? img0 + nkb4 * 2048 : img0;
            bf16x8 ah[NBUF][CH];
            auto loadc = [&](int buf, int c) {
                const int base = (c < KC ? img0 : img1) + (c % KC) * CH * 2048;
#pragma unroll
                for (int i = 0; i < CH; ++i) ah[buf][i] = load_sc1_u(dg_rsrc, lane * 16, base + i * 2048);
            };
#pragma unroll
            for (int c = 0; c < NBUF - 1 && c < NCH; ++c) loadc(c, c);
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                if (c + NBUF - 1 < NCH) loadc((c + NBUF - 1) % NBUF, c + NBUF - 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < CH; ++i)
                    acc[c < KC ? 0 : 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[c % NBUF][i], wh[(c % KC) * CH + i], acc[c < KC ? 0 : 1], 0, 0, 0);
            }
        }
        {
            const int r = lane & 15, q = lane >> 4;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) red[u][wave][(4 * q + e) * 16 + r] = acc[u][e];
        }
        lds_barrier();                                                             // (B)
        float dg[2][4];
        if (tid < 256) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
#pragma unroll
                for (int g = 0; g < 4; ++g) dg[u][g] = 0.f;
                if (cell[u]) {
                    float dh = s == 0 ? dh0[u] : 0.f;
                    if (s > 0) {
#pragma unroll
                        for (int k = 0; k < NWAVE; ++k) dh += red[u][k][tid];
                    }
                    const float ig = gv[u][0], fg = gv[u][1], gg = gv[u][2], og = gv[u][3];
                    const float tc = fast_tanh(cc[u]);
                    if (p.dy) {
                        float d = dyv[u];
                        if (p.dy_relu && !(og * tc > 0.f)) d = 0.f;
                        dh += d;
                    }
                    const float dcc = dcarry[u] + dh * og * (1.f - tc * tc);
                    const float d_o = dh * tc;
                    const float d_i = dcc * gg, d_f = dcc * cprev[u], d_g = dcc * ig;
                    dcarry[u] = dcc * fg;
                    dg[u][0] = d_i * ig * (1.f - ig);
                    dg[u][1] = d_f * fg * (1.f - fg);
                    dg[u][2] = d_g * (1.f - gg * gg);
                    dg[u][3] = d_o * og * (1.f - og);
#pragma unroll
                    for (int g = 0; g < 4; ++g) bsum[u][g] += dg[u][g];
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) dgbuf[u][g][ci][cj] = dg[u][g];
            }
        }
        lds_barrier();                                                             // (C)
        if (tid >= 256) {
            // wave 4 + g packs gate g: lanes 0-31 sub-tile 0's hi piece of dG_t, lanes 32-63 sub-tile 1's.  Gate g's columns j0..j0+15 are
            // k-groups 2 (jt & 1), 2 (jt & 1) + 1 of k-block g H/32 + jt/2
            const int g = wave - 4, u = lane >> 5, kg = (lane >> 4) & 1, row = lane & 15;
            const bool on = u == 0 || has1;
            bf16x8 hi;
#pragma unroll
            for (int e = 0; e < 8; ++e) hi[e] = (__bf16)dgbuf[u][g][row][kg * 8 + e];
            const int dst = ((t * NBT16 + 2 * bt + u) * nkb4 + g * (H / 32) + (jt >> 1)) * 2048 + (((jt & 1) * 2 + kg) * 16 + row) * 16;
            if (on) store_sc1(dg_rsrc, dst, hi);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            unsigned old = 0;
            if (lane == 0) old = atomicAdd(&s_published, 1u);
            old = __builtin_amdgcn_readfirstlane(old);
            if (old == 4u * (unsigned)s + 3u) publish_epoch(p.flags, bt * NJ + jt, (unsigned)(s + 1), lane);
            // ---- off the hand-off path: the tile in the two GEMM operand images (hi parts) ----
            if (on && p.img_rows) {
                const int grow = t * B + bt * 32 + u * 16 + row, kcol = g * H + j0 + kg * 8;
                const long blk = ((long)(grow >> 7) * nkb4 + (kcol >> 5)) * 2;
                const int r = grow & 127, c = (kcol & 31) >> 3;
                *reinterpret_cast<bf16x8 *>(p.img_rows + blk * 8192 + r * 64 + ((c ^ ((r >> 2) & 3)) << 4)) = hi;
            }
            if (on && p.img_cols) {
                bf16x8 hit;
#pragma unroll
                for (int e = 0; e < 8; ++e) hit[e] = (__bf16)dgbuf[u][g][kg * 8 + e][row];
                const int grow = g * H + j0 + row, kcol = t * B + bt * 32 + u * 16 + kg * 8;
                const int KT = (T * B + 31) >> 5;
                const long blk = ((long)(grow >> 7) * KT + (kcol >> 5)) * 2;
                const int r = grow & 127, c = (kcol & 31) >> 3;
                *reinterpret_cast<bf16x8 *>(p.img_cols + blk * 8192 + r * 64 + ((c ^ ((r >> 2) & 3)) << 4)) = hit;
            }
        }
        if (tid < 256) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (!cell[u]) continue;
                float *gp = p.gates + (t * B + brow[u]) * K + j0 + cj;
                if (!p.skip_dg) { gp[0] = dg[u][0]; gp[H] = dg[u][1]; gp[2 * H] = dg[u][2]; gp[3 * H] = dg[u][3]; }
                if (t > 0) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) gv[u][g] = p.gates[((t - 1) * B + brow[u]) * K + g * H + j0 + cj];
                    cc[u] = cprev[u];
                    cprev[u] = p.c[(t - 1) * BH + e0[u]];
                    if (p.dy) dyv[u] = p.dy[(long)(t - 1) * p.dy_stride_t + (long)brow[u] * p.dy_stride_b + j0 + cj];
                }
            }
        }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
        if (cell[u] && p.dc) p.dc[e0[u]] = dcarry[u];
    if (p.bias_part) {          // per 16-row batch tile: the sum over its rows (fixed order), one value per (gate, hidden unit)
        lds_barrier();
        if (tid < 256) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int g = 0; g < 4; ++g) dgbuf[u][g][ci][cj] = bsum[u][g];     // rows >= B hold zeros
        }
        lds_barrier();
        if (tid < 128) {
            const int u = tid >> 6, g = (tid >> 4) & 3, j = tid & 15;
            if (u == 0 || has1) {
                float sum = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) sum += dgbuf[u][g][r][j];
                p.bias_part[(long)(2 * bt + u) * K + (long)g * H + j0 + j] = sum;
            }
        }
    }
}

constexpr size_t FORCE_ONE_PER_CU_LDS32 = 48 * 1024;  // with the static arrays (up to 66 KiB): more than half a CU's LDS

template <typename K, typename A>
int launch32(K kernel, A a, int blocks, hipStream_t st) {
    static const int shift = getenv("HALO_PERSIST_REPLICA_SHIFT") ? atoi(getenv("HALO_PERSIST_REPLICA_SHIFT")) : 3;
    static const int nap = getenv("HALO_PERSIST_NAP") ? atoi(getenv("HALO_PERSIST_NAP")) : 2;
    a.poll_mode = 0; a.replica_shift = shift; a.nap = nap;
    a.status = halo_ctx_cur().status;
    if (hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FORCE_ONE_PER_CU_LDS32) != hipSuccess)
        return HALO_ELAUNCH;
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(512), FORCE_ONE_PER_CU_LDS32, st, a);
    return halo_launch_status();
}

}  // namespace

int halo_lstm_persist_fwd32(const PersistFwd &a0, hipStream_t st) {
    PersistFwd a = a0;
    a.mute = halo_ctx_cur().mute_block;
    const int blocks = (a.H / 16) * ((a.B + 31) / 32);
    switch (a.H / 256) {
        case 1: return launch32(lstm_persist_fwd32_kernel<1>, a, blocks, st);
        case 2: return launch32(lstm_persist_fwd32_kernel<2>, a, blocks, st);
        case 3: return launch32(lstm_persist_fwd32_kernel<3>, a, blocks, st);
        case 4: return launch32(lstm_persist_fwd32_kernel<4>, a, blocks, st);
        case 5: return launch32(lstm_persist_fwd32_kernel<5>, a, blocks, st);
        case 6: return launch32(lstm_persist_fwd32_kernel<6>, a, blocks, st);
        default: return HALO_ENOTSUP;
    }
}

int halo_lstm_persist_bwd32(const PersistBwd &a, hipStream_t st) {
    const int blocks = (a.H / 16) * ((a.B + 31) / 32);
    switch (a.H / 256) {
        case 1: return launch32(lstm_persist_bwd32_kernel<1>, a, blocks, st);
        case 2: return launch32(lstm_persist_bwd32_kernel<2>, a, blocks, st);
        case 3: return launch32(lstm_persist_bwd32_kernel<3>, a, blocks, st);
        case 4: return launch32(lstm_persist_bwd32_kernel<4>, a, blocks, st);
        case 5: return launch32(lstm_persist_bwd32_kernel<5>, a, blocks, st);
        case 6: return launch32(lstm_persist_bwd32_kernel<6>, a, blocks, st);
        default: return HALO_ENOTSUP;
    }
}
