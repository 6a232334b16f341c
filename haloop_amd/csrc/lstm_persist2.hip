// Both layers of a 2-layer LSTM stack in ONE weight-resident persistent launch per direction (gfx950; SURVEY.md K3 and section 7,
// "layer-wavefront pipelining: layer 1 step t overlaps layer 0 step t+1"; ha/rnn.py:11,25 is the 2-layer nn.LSTM this replaces).
//
// Why.  lstm_persist.hip runs one layer per launch: 21 dependent steps per layer and direction, each ending in a group-wide
// hand-off that costs more than the step's arithmetic, so the four chains of a training step are 84 hand-offs
// back to back while the matrix pipes idle.  Layer 1's step t needs layer 0's step t and its own step t-1 only, so here a
// COMBINED step s computes layer 0 at time s and layer 1 at time s-1 in the same workgroup, and the T+1 combined steps pay
// ONE hand-off each: the pieces of h0_s, dropout(h0_s) and h1_{s-1} are published together behind one epoch word.
//
// Decomposition.  Workgroup (jt, bt) owns hidden units [16 jt, +16) of batch rows [16 bt, +16) of BOTH layers, as in
// lstm_persist.hip.  Waves 0-3 are layer 0 (each holds a K-quarter of W_hh0's 64 gate rows in registers), waves 4-7 are layer 1
// (a K-quarter of W_hh1 in registers; the K-quarter of W_ih1 in LDS, one k-block per gate in registers at H=1024 so that the
// 128 KiB tile leaves room for the reduction buffers).  Single-pass bf16 only (HALO_MATH_BF16): three split-bf16 weight tiles
// of 256 KiB each do not fit a CU's 512 KiB register file + 160 KiB LDS; three bf16 tiles of 128 KiB do.
// In the backward the gate gradients of layer 1 at time t feed both layer 1's recurrence (times W_hh1) and the gradient arriving
// at layer 0 (times W_ih1): waves 4-7 multiply the SAME fragments by both matrices (W_hh1^T in registers, W_ih1^T in LDS), so
// the batched input-gradient GEMM between the chains and its operand images disappear too.
//
// Hand-off protocol, epoch words, block placement: lstm_persist.hip / lstm_persist_dev.h, unchanged.
#include <stdlib.h>
#include "halo_common.h"
#include "halo_internal.h"
#include "lstm_persist.h"
#include "lstm_persist_dev.h"

namespace {

// ================================================================================================================
// forward
// ================================================================================================================
// KBQ: k-blocks (32 deep) per wave = H / 128 (a K-quarter).
//
// Combined step s = 0 .. T+1: layer 0 at time s, layer 1 at time s - 2.  Layer 1 lags TWO steps so that its input half,
// dropout(h0_t) W_ih1^T, is not on the hand-off path: the image of dropout(h0_{s-1}) was complete when step s's poll matched, so
// waves 4-7 request its fragments at the start of step s (with the step's own) and multiply them at the END of the step, in the
// shadow of the next hand-off (when every wave would otherwise sit at the barrier); the sums (xacc) seed the accumulators of layer
// 1's step s+1.  On the path a layer-1 wave then does what a layer-0 wave does: KBQ fragment loads and 4 KBQ MFMAs against
// register-resident weights.  Both layers' pieces are packed and published by layer-0 waves (2: h1, 3: h0 and dropout(h0)).
// The two halves run DIFFERENT loop bodies (same three barriers per step): the compiler then allocates registers per half, and
// layer 1's 32 extra fragment registers do not have to coexist with layer 0's packing code.
struct Fwd2Shared {
    float (*red)[4][4][256];          // [layer][K-quarter][gate][batch row * 16 + hidden unit]
    float (*hbuf)[16][16];            // h0_t, dropout(h0_t), h1_t of the tile
    int *s_abort;
    unsigned *s_published;
};

template <int KBQ>
__device__ __forceinline__ void fwd2_layer0_waves(const Persist2Fwd &p, const Fwd2Shared sh, int jt, int bt, int wave, int lane, int u) {
    const int wq = wave & 3;
    const int H = p.H, B = p.B, T = p.T;
    const int NJ = H / 16, NBT = (B + 15) / 16, nkb = H / 32, j0 = jt * 16;
    bf16x8 wr[4][KBQ];                               // K-quarter wq of W_hh0: gates 0..3, k-blocks [wq*KBQ, +KBQ), hi halves
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int i = 0; i < KBQ; ++i)
            wr[g][i] = *reinterpret_cast<const bf16x8 *>(p.wp0 + (((long)jt * 4 + g) * nkb + wq * KBQ + i) * 2048 + lane * 16);
    const int ci = u >> 4, cj = u & 15;              // cell thread: (batch row, hidden unit) of the tile
    const int b = bt * 16 + ci;
    const bool cell = b < B;
    const int BH = B * H;                            // element counts of a layer fit 31 bits (halo_lstm_persist2_ok)
    const int e0 = b * H + j0 + cj;
    float cst = 0.f, gin[4] = {0.f, 0.f, 0.f, 0.f};  // c_{t-1}; the step's pre-activations x W_ih0^T + biases
    // this step's dropout multiplier of h0 (one Philox block per element): drawn in the PREVIOUS step's hand-off window, off the chain
    float dmul = (cell && p.xp) ? dropout_mult(p.drop, (uint64_t)e0) : 1.f;
    if (cell) {
        cst = p.c0[e0];
#pragma unroll
        for (int g = 0; g < 4; ++g) gin[g] = p.gates0[b * 4 * H + g * H + j0 + cj];
    }
    const __amdgpu_buffer_rsrc_t hp0_rsrc = make_rsrc(p.hp0), hp1_rsrc = make_rsrc(p.hp1);
    const __amdgpu_buffer_rsrc_t x_rsrc = p.xp ? make_rsrc(p.xp) : hp0_rsrc;
    const int my_replica = (xcc_id() + p.replica_shift) % PERSIST_REPLICAS;
    const int btl = bt - p.bt0;                      // the epoch words are indexed by the tile's place in this launch
    const unsigned *grp_flags = p.flags + my_replica * PERSIST_REPLICA_WORDS + PERSIST_FLAG_HEADER + btl * NJ;
    for (int s = 0; s <= T + 1; ++s) {
        if (wave == 0) stamp(p.stamps, T + 2, s, 0, lane);
        const bool act0 = s < T, act1 = s >= 2;
        // ---- epoch s: every workgroup of the batch group has published h0_{s-1}, dropout(h0_{s-1}) and h1_{s-3} ----
        bool ok = true;
        if (s > 0 && wave == 1) ok = poll_group(grp_flags, 0, NJ, p.epoch0 + (unsigned)s, lane, p.nap);
        if (!ok && lane == 0) {
            *sh.s_abort = 1;
            raise_abort(p.flags, p.status);
        }
        lds_barrier();                                                             // (A)
        if (*sh.s_abort) return;
        if (wave == 0) stamp(p.stamps, T + 2, s, 1, lane);
        if (act0) {
            const int img = ((s * NBT + bt) * nkb + wq * KBQ) * 2048;              // image s = h0_{s-1}; wave-uniform: the loads' scalar offset
            bf16x8 ah[KBQ];
#pragma unroll
            for (int i = 0; i < KBQ; ++i) ah[i] = load_sc1_u(hp0_rsrc, lane * 16, img + i * 2048);
            __builtin_amdgcn_sched_barrier(0);
            f32x4 acc[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < KBQ; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], wr[g][i], acc[g], 0, 0, 0);
            const int r = lane & 15, q = lane >> 4;      // D layout: col = lane & 15 (hidden unit), row = 4 (lane >> 4) + reg (batch row)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) sh.red[0][wq][g][(4 * q + e) * 16 + r] = acc[g][e];
        }
        lds_barrier();                                                             // (B)
        if (wave == 0) stamp(p.stamps, T + 2, s, 2, lane);
        float ig = 0.f, fg = 0.f, gg = 0.f, og = 0.f, h = 0.f, xv = 0.f;
        if (act0) {
            if (cell) {
                float pre[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float sum = 0.f;
#pragma unroll
                    for (int k = 0; k < 4; ++k) sum += sh.red[0][k][g][u];
                    pre[g] = sum + gin[g];
                }
                h = persist2_fwd_cell(pre, cst, ig, fg, gg, og);
                if (p.xp) xv = h * dmul;
            }
            sh.hbuf[0][ci][cj] = h;                      // rows >= B: zeros
            if (p.xp) sh.hbuf[1][ci][cj] = xv;
        }
        lds_barrier();                                                             // (C)
        if (wave == 0) stamp(p.stamps, T + 2, s, 3, lane);
        if ((wave == 3 && act0) || (wave == 2 && act1)) {
            // wave 3: lanes 0-31 the hi image piece of h0_s (image s+1 of layer 0), lanes 32-63 the piece of dropout(h0_s) (image s of
            // xp); wave 2: lanes 0-31 the piece of h1_{s-2} (image s-1 of layer 1).  The tile is k-groups 2 (jt & 1), 2 (jt & 1) + 1 of
            // k-block jt / 2.
            const int sel = lane >> 5, kg = (lane >> 4) & 1, row = lane & 15;
            const float(*src)[16] = wave == 2 ? sh.hbuf[2] : sh.hbuf[sel];
            bf16x8 hi;
#pragma unroll
            for (int e = 0; e < 8; ++e) hi[e] = (__bf16)src[row][kg * 8 + e];
            const int within = (jt >> 1) * 2048 + (((jt & 1) * 2 + kg) * 16 + row) * 16;
            if (wave == 3) {
                if (sel == 0) store_sc1(hp0_rsrc, (((s + 1) * NBT + bt) * nkb) * 2048 + within, hi);
                else if (p.xp) store_sc1(x_rsrc, ((s * NBT + bt) * nkb) * 2048 + within, hi);
            } else if (sel == 0) {
                store_sc1(hp1_rsrc, (((s - 1) * NBT + bt) * nkb) * 2048 + within, hi);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // the write-through stores have left
            // the last storing wave of the step to get here signals for the workgroup (counter in LDS: Guideline 16); through
            // combined step s layer 0 has stored min(s + 1, T) times, layer 1 max(s - 1, 0) times
            unsigned old = 0;
            if (lane == 0) old = atomicAdd(sh.s_published, 1u);
            old = __builtin_amdgcn_readfirstlane(old);
            const unsigned target = (unsigned)((s + 1 < T ? s + 1 : T) + (s >= 2 ? s - 1 : 0));
            if (old + 1u == target && (int)blockIdx.x != p.mute) publish_epoch(p.flags, btl * NJ + jt, p.epoch0 + (unsigned)(s + 1), lane);
            stamp(p.stamps, T + 2, s, wave == 3 ? 4 : 5, lane);
        } else if (wave == 3 && !act0 && !act1) {
            publish_epoch(p.flags, btl * NJ + jt, p.epoch0 + (unsigned)(s + 1), lane);       // T = 1: neither layer has a step here, the epoch still moves
        }
        if (p.img_hT0 && ((wave == 3 && act0) || (wave == 2 && act1))) {
            // ---- off the hand-off path: the tile, transposed, in the operand images of the weight-gradient products.  Image row = hidden
            //      unit j0 + (lane & 15), one 16-byte chunk = 8 consecutive batch rows.  wave 3: lanes 0-31 h0_s as the state BEFORE step
            //      s + 1 (column block s + 1 of img_hT0), lanes 32-63 dropout(h0_s) = layer 1's input at time s (block s of img_xT1);
            //      wave 2: lanes 0-31 h1_{s-2} as the state before step s - 1 (block s - 1 of img_hT1) ----
            const int sel = lane >> 5, kg = (lane >> 4) & 1, row = lane & 15;
            const float(*src)[16] = wave == 2 ? sh.hbuf[2] : (sel == 1 && p.xp ? sh.hbuf[1] : sh.hbuf[0]);
            const int tb = wave == 2 ? s - 1 : (sel == 0 ? s + 1 : s);           // column block (time index of the image)
            char *img = wave == 2 ? (sel == 0 ? p.img_hT1 : nullptr) : (sel == 0 ? p.img_hT0 : p.img_xT1);
            if (img && tb < T) {
                bf16x8 hit;
#pragma unroll
                for (int e = 0; e < 8; ++e) hit[e] = (__bf16)src[kg * 8 + e][row];
                const int grow = j0 + row, kcol = tb * B + bt * 16 + kg * 8;
                const int KT = (T * B + 31) >> 5;
                const long blk = ((long)(grow >> 7) * KT + (kcol >> 5)) * 2;
                const int r = grow & 127, cc = (kcol & 31) >> 3;
                *reinterpret_cast<bf16x8 *>(img + blk * 8192 + r * 64 + ((cc ^ ((r >> 2) & 3)) << 4)) = hit;
            }
        }
        if (act0 && cell) {
            float *gp = p.gates0 + (s * B + b) * 4 * H + j0 + cj;
            gp[0] = ig; gp[H] = fg; gp[2 * H] = gg; gp[3 * H] = og;
            p.c0[(s + 1) * BH + e0] = cst;
            p.h0[(s + 1) * BH + e0] = h;
            if (p.ydrop) p.ydrop[s * BH + e0] = xv;
            if (p.xp && s + 1 < T) dmul = dropout_mult(p.drop, (uint64_t)((long)(s + 1) * BH + e0));
            if (s + 1 < T) {
#pragma unroll
                for (int g = 0; g < 4; ++g) gin[g] = p.gates0[((s + 1) * B + b) * 4 * H + g * H + j0 + cj];
            }
        }
    }
}

template <int KBQ>
__device__ __forceinline__ void fwd2_layer1_waves(const Persist2Fwd &p, const Fwd2Shared sh, char *wi_lds, int jt, int bt, int wave, int lane,
                                                  int u) {
    // W_ih1's K-quarter is 4 KBQ fragments of 1 KiB per layer-1 wave, kept in LDS; at KBQ = 8 the four quarters (128 KiB) and the
    // reduction buffers (32 KiB) would be the whole 160 KiB, so the first NWREG fragments (gates 0 .. NWREG-1 of k-block 0) stay
    // in registers -- as few as make room for hbuf: the register file is as full as the LDS
    constexpr int NWREG = KBQ >= 8 ? 2 : 0;
    constexpr int NWLDS = 4 * KBQ - NWREG;
    const int wq = wave & 3;
    const int H = p.H, B = p.B, T = p.T;
    const int NBT = (B + 15) / 16, nkb = H / 32, j0 = jt * 16;
    bf16x8 wr[4][KBQ];                               // K-quarter wq of W_hh1
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int i = 0; i < KBQ; ++i)
            wr[g][i] = *reinterpret_cast<const bf16x8 *>(p.wp1 + (((long)jt * 4 + g) * nkb + wq * KBQ + i) * 2048 + lane * 16);
    bf16x8 wir[NWREG > 0 ? NWREG : 1];
    char *my_wi = wi_lds + (long)wq * NWLDS * 1024 + lane * 16;         // + (i * 4 + g - NWREG) * 1024: this lane's 16 bytes
#pragma unroll
    for (int i = 0; i < KBQ; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const bf16x8 v = *reinterpret_cast<const bf16x8 *>(p.wpi + (((long)jt * 4 + g) * nkb + wq * KBQ + i) * 2048 + lane * 16);
            if (i * 4 + g < NWREG) wir[i * 4 + g < NWREG ? i * 4 + g : 0] = v;
            else *reinterpret_cast<bf16x8 *>(my_wi + (i * 4 + g - NWREG) * 1024) = v;           // read back by this lane only
            if (g == 3 && (i & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // sixteen fragments in flight at a time: the registers are half full already
        }
    const int ci = u >> 4, cj = u & 15;
    const int b = bt * 16 + ci;
    const bool cell = b < B;
    const int BH = B * H;
    const int e0 = b * H + j0 + cj;
    float cst = 0.f, bias[4] = {0.f, 0.f, 0.f, 0.f};
    if (cell) {
        cst = p.c1[e0];
#pragma unroll
        for (int g = 0; g < 4; ++g) bias[g] = p.b_ih1[g * H + j0 + cj] + p.b_hh1[g * H + j0 + cj];
    }
    const __amdgpu_buffer_rsrc_t hp1_rsrc = make_rsrc(p.hp1);
    const __amdgpu_buffer_rsrc_t x_rsrc = make_rsrc(p.xp ? p.xp : p.hp0);
    f32x4 xacc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) xacc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s <= T + 1; ++s) {
        if (wave == 4) stamp(p.stamps, T + 2, s, 8, lane);
        const bool act1 = s >= 2, xin = s >= 1 && s <= T;
        const int t = s - 2;
        lds_barrier();                                                             // (A)
        if (*sh.s_abort) return;
        // the input fragments of the NEXT step (time s - 1: image s-1 of xp; without dropout image s of layer 0), complete since this
        // step's poll matched, are requested with the step's own: nothing of them is left for the hand-off window, where loads
        // would sit in front of the epoch stores and polls in the CU's memory queue
        bf16x8 ax[KBQ];
        if (xin) {
            const int ximg = (((p.xp ? s - 1 : s) * NBT + bt) * nkb + wq * KBQ) * 2048;
#pragma unroll
            for (int i = 0; i < KBQ; ++i) ax[i] = load_sc1_u(x_rsrc, lane * 16, ximg + i * 2048);
        }
        if (act1) {
            const int img = ((t * NBT + bt) * nkb + wq * KBQ) * 2048;              // image t = h1_{t-1}
            bf16x8 ah[KBQ];
#pragma unroll
            for (int i = 0; i < KBQ; ++i) ah[i] = load_sc1_u(hp1_rsrc, lane * 16, img + i * 2048);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < KBQ; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) xacc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], wr[g][i], xacc[g], 0, 0, 0);
            const int r = lane & 15, q = lane >> 4;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) sh.red[1][wq][g][(4 * q + e) * 16 + r] = xacc[g][e];
        }
        lds_barrier();                                                             // (B)
        float ig = 0.f, fg = 0.f, gg = 0.f, og = 0.f, h = 0.f;
        if (act1) {
            if (cell) {
                float pre[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float sum = 0.f;
#pragma unroll
                    for (int k = 0; k < 4; ++k) sum += sh.red[1][k][g][u];
                    pre[g] = sum + bias[g];
                }
                h = persist2_fwd_cell(pre, cst, ig, fg, gg, og);
            }
            sh.hbuf[2][ci][cj] = h;
        }
        lds_barrier();                                                             // (C)
        if (act1 && cell) {
            float *gp = p.gates1 + (t * B + b) * 4 * H + j0 + cj;
            gp[0] = ig; gp[H] = fg; gp[2 * H] = gg; gp[3 * H] = og;
            p.c1[(t + 1) * BH + e0] = cst;
            p.h1[(t + 1) * BH + e0] = h;
            if (p.y_mode != 0) p.y[(long)t * p.y_stride_t + (long)b * p.y_stride_b + j0 + cj] = p.y_mode == 2 ? fmaxf(h, 0.f) : h;
        }
        // ---- in the shadow of the hand-off: layer 1's input half for its next step ----
        if (wave == 4) stamp(p.stamps, T + 2, s, 7, lane);
#pragma unroll
        for (int g = 0; g < 4; ++g) xacc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (xin) {
#pragma unroll
            for (int i = 0; i < KBQ; ++i) {
                bf16x8 w[4];
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    w[g] = i * 4 + g < NWREG ? wir[i * 4 + g < NWREG ? i * 4 + g : 0]
                                             : *reinterpret_cast<const bf16x8 *>(my_wi + (i * 4 + g - NWREG) * 1024);
#pragma unroll
                for (int g = 0; g < 4; ++g) xacc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax[i], w[g], xacc[g], 0, 0, 0);
                if (i & 1) __builtin_amdgcn_sched_barrier(0);   // two k-blocks' LDS fragments in flight at a time: the registers are full
            }
        }
        if (wave == 4) stamp(p.stamps, T + 2, s, 6, lane);
    }
}

template <int KBQ>
__global__ __launch_bounds__(512, 2) void lstm_persist2_fwd_kernel(const Persist2Fwd p) {
    __shared__ float red[2][4][4][256];
    __shared__ __attribute__((aligned(16))) float hbuf[3][16][16];
    __shared__ int s_abort;
    __shared__ unsigned s_published;
    extern __shared__ __attribute__((aligned(16))) char wi_lds[];    // [K-quarter][NWLDS] W_ih1 fragments of 1 KiB (k-block major, gate minor)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int jt, bt;
    map_block(blockIdx.x, gridDim.x, p.H / 16, p.nbt, jt, bt);
    bt += p.bt0;
    if (tid == 0) { s_abort = 0; s_published = 0; }
    if (wave == 0) stamp(p.stamps, p.T + 2, 0, 14, lane);            // (diagnostic: launch entry / exit of this workgroup)
    const Fwd2Shared sh = {red, hbuf, &s_abort, &s_published};
    // (the first barrier of either loop orders the two initialisations above before any use)
    if (wave < 4) fwd2_layer0_waves<KBQ>(p, sh, jt, bt, wave, lane, tid & 255);
    else fwd2_layer1_waves<KBQ>(p, sh, wi_lds, jt, bt, wave, lane, tid & 255);
    if (wave == 0) stamp(p.stamps, p.T + 2, 0, 15, lane);
}

// ================================================================================================================
// backward
// ================================================================================================================
// KC: chunks of 4 k-blocks per wave = (4H / 32 / 4) / 4 = H / 128.
//
// Combined step s = 0 .. T: layer 1 at time T-1-s, layer 0 at time T-s.  The gate gradients of layer 1 at time T-s (image T-s of dgp1)
// feed layer 1's recurrence (times W_hh1^T, registers) AND the gradient arriving at layer 0 (times W_ih1^T, LDS) from ONE set of
// fragment loads in waves 4-7; waves 0-3 contract layer 0's own gate gradients (image T-s+1 of dgp0) and, having half the matrix
// work, pack and publish both layers' pieces and write the GEMM operand images.  As in the forward the halves run different loop
// bodies with the same barriers.
struct Bwd2Shared {
    float (*red)[4][256];             // [layer 0 recurrent | layer 1 recurrent | from layer 1 into layer 0][K-quarter]
    float (*dgbuf)[4][16][16];        // [layer][gate][batch row][hidden unit]
    int *s_abort;
    unsigned *s_published;
};

// one layer's cell update of the backward: dh -> gate gradients (and the carried cell gradient); arithmetic in lstm_persist_dev.h
struct Bwd2Cell {
    float gv[4], cc, cprev, dcarry;
    __device__ __forceinline__ void update(float dh, float (&dg)[4]) { dcarry = persist2_bwd_cell(gv, cc, cprev, dcarry, dh, dg); }
};

template <int KC>
__device__ __forceinline__ void bwd2_layer0_waves(const Persist2Bwd &p, const Bwd2Shared sh, int jt, int bt, int wave, int lane, int u) {
    constexpr int KBW = 4 * KC, CH = 4, NCH = KBW / CH, NBUF = 3;     // three buffers of 4 fragments: 12 loads of 1 KiB in flight per wave
    const int wq = wave & 3;
    const int H = p.H, B = p.B, T = p.T, K = 4 * H;
    const int NJ = H / 16, NBT = (B + 15) / 16, nkb4 = K / 32, j0 = jt * 16;
    bf16x8 wr[KBW];                                  // K-quarter wq of W_hh0^T: columns j0..j0+15, k-blocks [wq*KBW, +KBW)
#pragma unroll
    for (int i = 0; i < KBW; ++i) wr[i] = *reinterpret_cast<const bf16x8 *>(p.wpT0 + ((long)jt * nkb4 + wq * KBW + i) * 2048 + lane * 16);
    const int ci = u >> 4, cj = u & 15;
    const int b = bt * 16 + ci;
    const bool cell = b < B;
    const int BH = B * H;
    const int e0 = b * H + j0 + cj;
    Bwd2Cell c;
    c.cc = c.cprev = c.dcarry = 0.f;
    float dh0 = 0.f, bsum[4] = {0.f, 0.f, 0.f, 0.f};
    // layer 0's output mask at the step's time (one Philox block per element): drawn in the previous step's hand-off window
    float dmul = cell ? dropout_mult(p.drop, (uint64_t)((long)(T - 1) * BH + e0)) : 1.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) c.gv[g] = 0.f;
    if (cell) {
        const int t = T - 1;
#pragma unroll
        for (int g = 0; g < 4; ++g) c.gv[g] = p.gates0[(t * B + b) * K + g * H + j0 + cj];
        c.cc = p.c0[(t + 1) * BH + e0];
        c.cprev = p.c0[t * BH + e0];
        if (p.dcinit0) c.dcarry = p.dcinit0[e0];
        if (p.dhinit0) dh0 = p.dhinit0[e0];
    }
    const __amdgpu_buffer_rsrc_t dg0_rsrc = make_rsrc(p.dgp0), dg1_rsrc = make_rsrc(p.dgp1);
    const int my_replica = (xcc_id() + p.replica_shift) % PERSIST_REPLICAS;
    const int btl = bt - p.bt0;                      // the epoch words are indexed by the tile's place in this launch
    const unsigned *grp_flags = p.flags + my_replica * PERSIST_REPLICA_WORDS + PERSIST_FLAG_HEADER + btl * NJ;
    for (int s = 0; s <= T; ++s) {
        if (wave == 0) stamp(p.stamps, T + 1, s, 0, lane);
        const bool act = s >= 1;                     // layer 0 has a cell update in combined step s ...
        const int t = T - s;                          // ... at this time
        bool ok = true;
        if (s > 0 && wave == 1) ok = poll_group(grp_flags, 0, NJ, p.epoch0 + (unsigned)s, lane, p.nap);
        if (!ok && lane == 0) {
            *sh.s_abort = 1;
            raise_abort(p.abort_word, p.status);
        }
        lds_barrier();                                                             // (A)
        if (*sh.s_abort) return;
        if (wave == 0) stamp(p.stamps, T + 1, s, 1, lane);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (s >= 2) {                                 // dG0 of time t + 1 (image T-s+1): layer 0's recurrent term
            const int img = (((T - s + 1) * NBT + bt) * nkb4 + wq * KBW) * 2048;
            bf16x8 ah[NBUF][CH];
            auto loadc = [&](int buf, int cidx) {
#pragma unroll
                for (int i = 0; i < CH; ++i) ah[buf][i] = load_sc1_u(dg0_rsrc, lane * 16, img + (cidx * CH + i) * 2048);
            };
#pragma unroll
            for (int cidx = 0; cidx < NBUF - 1 && cidx < NCH; ++cidx) loadc(cidx, cidx);
#pragma unroll
            for (int cidx = 0; cidx < NCH; ++cidx) {
                if (cidx + NBUF - 1 < NCH) loadc((cidx + NBUF - 1) % NBUF, cidx + NBUF - 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < CH; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[cidx % NBUF][i], wr[cidx * CH + i], acc, 0, 0, 0);
            }
        }
        {
            const int r = lane & 15, q = lane >> 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) sh.red[0][wq][(4 * q + e) * 16 + r] = acc[e];
        }
        lds_barrier();                                                             // (B)
        if (wave == 0) stamp(p.stamps, T + 1, s, 2, lane);
        float dg[4] = {0.f, 0.f, 0.f, 0.f};
        if (act) {
            if (cell) {
                float rec = 0.f, above = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) { rec += sh.red[0][k][u]; above += sh.red[2][k][u]; }
                c.update(persist2_add_masked(s == 1 ? dh0 : rec, above, dmul), dg);      // (dmul: layer 0's own output mask)
#pragma unroll
                for (int g = 0; g < 4; ++g) bsum[g] += dg[g];
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) sh.dgbuf[0][g][ci][cj] = dg[g];
        }
        lds_barrier();                                                             // (C)
        if (wave == 0) stamp(p.stamps, T + 1, s, 3, lane);
        {
            // pack: wave g takes gate g; lanes 0-31 layer 1's piece (time T-1-s, steps 0 .. T-1), lanes 32-63 layer 0's (time T-s,
            // steps 1 .. T).  Gate g's columns j0..j0+15 are k-groups 2 (jt & 1), 2 (jt & 1) + 1 of k-block g H/32 + jt/2.
            const int g = wq, lay = lane < 32 ? 1 : 0, kg = (lane >> 4) & 1, row = lane & 15;
            const bool on = lay ? (s < T) : (s >= 1);
            const int tt = lay ? T - 1 - s : T - s;
            bf16x8 hi;
            if (on) {
#pragma unroll
                for (int e = 0; e < 8; ++e) hi[e] = (__bf16)sh.dgbuf[lay][g][row][kg * 8 + e];
                const int dst = ((tt * NBT + bt) * nkb4 + g * (H / 32) + (jt >> 1)) * 2048 + (((jt & 1) * 2 + kg) * 16 + row) * 16;
                if (lay) store_sc1(dg1_rsrc, dst, hi);
                else store_sc1(dg0_rsrc, dst, hi);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            unsigned old = 0;
            if (lane == 0) old = atomicAdd(sh.s_published, 1u);
            old = __builtin_amdgcn_readfirstlane(old);
            if (old == 4u * (unsigned)s + 3u) publish_epoch(p.flags, btl * NJ + jt, p.epoch0 + (unsigned)(s + 1), lane);
            if (wave == 3) stamp(p.stamps, T + 1, s, 4, lane);
            // ---- off the hand-off path: the tile in the GEMM operand images (hi parts; gemm_bf16x3.hip layout) ----
            if (on) {
                char *img_rows = lay ? nullptr : p.img_rows0;
                char *img_cols = lay ? p.img_cols1 : p.img_cols0;
                if (img_rows) {          // rows tt*B + b, k = g*H + j0 + 8 kg ..
                    const int grow = tt * B + bt * 16 + row, kcol = g * H + j0 + kg * 8;
                    const long blk = ((long)(grow >> 7) * nkb4 + (kcol >> 5)) * 2;
                    const int r = grow & 127, cc = (kcol & 31) >> 3;
                    *reinterpret_cast<bf16x8 *>(img_rows + blk * 8192 + r * 64 + ((cc ^ ((r >> 2) & 3)) << 4)) = hi;
                }
                if (img_cols) {          // rows g*H + j0 + row (hidden unit), k = tt*B + bt*16 + 8 kg ..
                    bf16x8 hit;
#pragma unroll
                    for (int e = 0; e < 8; ++e) hit[e] = (__bf16)sh.dgbuf[lay][g][kg * 8 + e][row];
                    const int grow = g * H + j0 + row, kcol = tt * B + bt * 16 + kg * 8;
                    const int KT = (T * B + 31) >> 5;
                    const long blk = ((long)(grow >> 7) * KT + (kcol >> 5)) * 2;
                    const int r = grow & 127, cc = (kcol & 31) >> 3;
                    *reinterpret_cast<bf16x8 *>(img_cols + blk * 8192 + r * 64 + ((cc ^ ((r >> 2) & 3)) << 4)) = hit;
                }
            }
        }
        if (act && cell) {
            float *gp = p.gates0 + (t * B + b) * K + j0 + cj;
            if (!p.skip_dg0) { gp[0] = dg[0]; gp[H] = dg[1]; gp[2 * H] = dg[2]; gp[3 * H] = dg[3]; }
            if (t > 0) {
                dmul = dropout_mult(p.drop, (uint64_t)((long)(t - 1) * BH + e0));
#pragma unroll
                for (int g = 0; g < 4; ++g) c.gv[g] = p.gates0[((t - 1) * B + b) * K + g * H + j0 + cj];
                c.cc = c.cprev;
                c.cprev = p.c0[(t - 1) * BH + e0];
            }
        }
    }
    if (cell && p.dc0) p.dc0[e0] = c.dcarry;
    lds_barrier();
#pragma unroll
    for (int g = 0; g < 4; ++g) sh.dgbuf[0][g][ci][cj] = bsum[g];              // rows >= B hold zeros
    lds_barrier();
    if (p.bias_part0 && u < 64) {
        const int g = u >> 4, j = u & 15;
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += sh.dgbuf[0][g][r][j];
        p.bias_part0[(long)bt * K + g * H + j0 + j] = sum;
    }
}

template <int KC>
__device__ __forceinline__ void bwd2_layer1_waves(const Persist2Bwd &p, const Bwd2Shared sh, char *wi_lds, int jt, int bt, int wave, int lane,
                                                  int u) {
    constexpr int KBW = 4 * KC, CH = 4, NCH = KBW / CH, NBUF = 3;     // three buffers of 4: 12 loads in flight (the registers also hold LDS fragments)
    const int wq = wave & 3;
    const int H = p.H, B = p.B, T = p.T, K = 4 * H;
    const int NBT = (B + 15) / 16, nkb4 = K / 32, j0 = jt * 16;
    bf16x8 wr[KBW];                                  // K-quarter wq of W_hh1^T
#pragma unroll
    for (int i = 0; i < KBW; ++i) wr[i] = *reinterpret_cast<const bf16x8 *>(p.wpT1 + ((long)jt * nkb4 + wq * KBW + i) * 2048 + lane * 16);
    char *my_wi = wi_lds + (long)wq * KBW * 1024 + lane * 16;                     // K-quarter wq of W_ih1^T: read back by this lane only
#pragma unroll
    for (int i = 0; i < KBW; ++i) {
        *reinterpret_cast<bf16x8 *>(my_wi + i * 1024) =
            *reinterpret_cast<const bf16x8 *>(p.wpTi + ((long)jt * nkb4 + wq * KBW + i) * 2048 + lane * 16);
        if ((i & 15) == 15) __builtin_amdgcn_sched_barrier(0);     // sixteen fragments in flight at a time: W_hh1^T already fills half the registers
    }
    const int ci = u >> 4, cj = u & 15;
    const int b = bt * 16 + ci;
    const bool cell = b < B;
    const int BH = B * H;
    const int e0 = b * H + j0 + cj;
    Bwd2Cell c;
    c.cc = c.cprev = c.dcarry = 0.f;
    float dh0 = 0.f, dyv = 0.f, bsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 4; ++g) c.gv[g] = 0.f;
    if (cell) {
        const int t = T - 1;
#pragma unroll
        for (int g = 0; g < 4; ++g) c.gv[g] = p.gates1[(t * B + b) * K + g * H + j0 + cj];
        c.cc = p.c1[(t + 1) * BH + e0];
        c.cprev = p.c1[t * BH + e0];
        if (p.dy) dyv = p.dy[(long)t * p.dy_stride_t + (long)b * p.dy_stride_b + j0 + cj];
        if (p.dcinit1) c.dcarry = p.dcinit1[e0];
        if (p.dhinit1) dh0 = p.dhinit1[e0];
    }
    const __amdgpu_buffer_rsrc_t dg1_rsrc = make_rsrc(p.dgp1);
    for (int s = 0; s <= T; ++s) {
        const bool act = s < T;                      // layer 1 has a cell update in combined step s ...
        const int t = T - 1 - s;                      // ... at this time
        lds_barrier();                                                             // (A)
        if (*sh.s_abort) return;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
        if (s >= 1) {                                 // dG1 of time T-s: layer 1's recurrent term (acc) and layer 0's incoming gradient (acc2)
            const int img = (((T - s) * NBT + bt) * nkb4 + wq * KBW) * 2048;
            bf16x8 ah[NBUF][CH];
            auto loadc = [&](int buf, int cidx) {
#pragma unroll
                for (int i = 0; i < CH; ++i) ah[buf][i] = load_sc1_u(dg1_rsrc, lane * 16, img + (cidx * CH + i) * 2048);
            };
#pragma unroll
            for (int cidx = 0; cidx < NBUF - 1 && cidx < NCH; ++cidx) loadc(cidx, cidx);
#pragma unroll
            for (int cidx = 0; cidx < NCH; ++cidx) {
                if (cidx + NBUF - 1 < NCH) loadc((cidx + NBUF - 1) % NBUF, cidx + NBUF - 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    const bf16x8 w = *reinterpret_cast<const bf16x8 *>(my_wi + (cidx * CH + i) * 1024);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[cidx % NBUF][i], wr[cidx * CH + i], acc, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[cidx % NBUF][i], w, acc2, 0, 0, 0);
                }
            }
        }
        {
            const int r = lane & 15, q = lane >> 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sh.red[1][wq][(4 * q + e) * 16 + r] = acc[e];
                sh.red[2][wq][(4 * q + e) * 16 + r] = acc2[e];
            }
        }
        lds_barrier();                                                             // (B)
        float dg[4] = {0.f, 0.f, 0.f, 0.f};
        if (act) {
            if (cell) {
                float dh = 0.f;
                if (s == 0) dh = dh0;
                else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) dh += sh.red[1][k][u];
                }
                if (p.dy) {
                    const float tc = persist2_tanh(c.cc);
                    float d = dyv;
                    if (p.dy_relu && !(c.gv[3] * tc > 0.f)) d = 0.f;
                    dh += d;
                }
                c.update(dh, dg);
#pragma unroll
                for (int g = 0; g < 4; ++g) bsum[g] += dg[g];
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) sh.dgbuf[1][g][ci][cj] = dg[g];
        }
        lds_barrier();                                                             // (C)
        if (act && cell) {
            float *gp = p.gates1 + (t * B + b) * K + j0 + cj;
            if (!p.skip_dg1) { gp[0] = dg[0]; gp[H] = dg[1]; gp[2 * H] = dg[2]; gp[3 * H] = dg[3]; }
            if (t > 0) {
#pragma unroll
                for (int g = 0; g < 4; ++g) c.gv[g] = p.gates1[((t - 1) * B + b) * K + g * H + j0 + cj];
                c.cc = c.cprev;
                c.cprev = p.c1[(t - 1) * BH + e0];
                if (p.dy) dyv = p.dy[(long)(t - 1) * p.dy_stride_t + (long)b * p.dy_stride_b + j0 + cj];
            }
        }
    }
    if (cell && p.dc1) p.dc1[e0] = c.dcarry;
    lds_barrier();
#pragma unroll
    for (int g = 0; g < 4; ++g) sh.dgbuf[1][g][ci][cj] = bsum[g];
    lds_barrier();
    if (p.bias_part1 && u < 64) {
        const int g = u >> 4, j = u & 15;
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += sh.dgbuf[1][g][r][j];
        p.bias_part1[(long)bt * K + g * H + j0 + j] = sum;
    }
}

template <int KC>
__global__ __launch_bounds__(512, 2) void lstm_persist2_bwd_kernel(const Persist2Bwd p) {
    __shared__ float red[3][4][256];
    __shared__ __attribute__((aligned(16))) float dgbuf[2][4][16][16];
    __shared__ int s_abort;
    __shared__ unsigned s_published;
    extern __shared__ __attribute__((aligned(16))) char wi_lds[];    // [K-quarter][4 KC] W_ih1^T fragments of 1 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int jt, bt;
    map_block(blockIdx.x, gridDim.x, p.H / 16, p.nbt, jt, bt);
    bt += p.bt0;
    if (tid == 0) { s_abort = 0; s_published = 0; }
    // the call's abort word is only ever raised after a 0.2 s wait: clearing it here, at the start of the launch, cannot lose one
    if (blockIdx.x == 0 && tid == 0 && p.abort_word != p.flags) __hip_atomic_store(p.abort_word, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (wave == 0) stamp(p.stamps, p.T + 1, 0, 14, lane);            // (diagnostic: launch entry / exit of this workgroup)
    const Bwd2Shared sh = {red, dgbuf, &s_abort, &s_published};
    if (wave < 4) bwd2_layer0_waves<KC>(p, sh, jt, bt, wave, lane, tid & 255);
    else bwd2_layer1_waves<KC>(p, sh, wi_lds, jt, bt, wave, lane, tid & 255);
    if (wave == 0) stamp(p.stamps, p.T + 1, 0, 15, lane);
}

int g_cu_count2 = 0;
inline int cu_count2() {
    if (!g_cu_count2) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) g_cu_count2 = prop.multiProcessorCount;
        else g_cu_count2 = -1;
    }
    return g_cu_count2;
}

// batch tiles a launch can hold: every workgroup of a launch must be resident (one per CU)
inline int tiles_per_launch(int H) { return max(1, cu_count2() / (H / 16)); }

constexpr size_t MIN_DYN_LDS = 64 * 1024;             // with the static arrays: more than half a CU's LDS -> one workgroup per CU

inline void set_mute(Persist2Fwd &a) { a.mute = halo_ctx_cur().mute_block; }
inline void set_mute(Persist2Bwd &) {}

template <typename K, typename A>
int launch2(K kernel, const A &a0, int blocks, size_t dyn, hipStream_t st) {
    static_assert(sizeof(A) <= 4096, "kernel arguments");
    A a = a0;
    static const int shift = getenv("HALO_PERSIST_REPLICA_SHIFT") ? atoi(getenv("HALO_PERSIST_REPLICA_SHIFT")) : 3;
    static const int nap = getenv("HALO_PERSIST_NAP") ? atoi(getenv("HALO_PERSIST_NAP")) : 2;
    a.poll_mode = 0; a.replica_shift = shift; a.nap = nap;
    a.status = halo_ctx_cur().status;
    set_mute(a);
    if (dyn < MIN_DYN_LDS) dyn = MIN_DYN_LDS;
    if (hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn) != hipSuccess) return HALO_ELAUNCH;
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(512), dyn, st, a);
    return halo_launch_status();
}

}  // namespace

void halo_lstm_persist2_enable(int on) { halo_ctx_cur().lstm_persistent2 = on ? 1 : 0; }
void halo_lstm_interleave_enable(int on) { halo_ctx_cur().lstm_interleave = on ? 1 : 0; }

bool halo_lstm_persist2_ok(int T, int B, int H, int L) {
    static const bool env_off = getenv("HALO_LSTM_PERSIST2") && atoi(getenv("HALO_LSTM_PERSIST2")) == 0;
    if (env_off || !halo_ctx_cur().lstm_persistent2) return false;
    // the per-layer recurrence's switch and shape rules, but not its limit on the batch: a large batch runs as several launches
    if (!halo_lstm_persist_ok(16, H) || B <= 0 || !halo_lstm_persist_fits(T, B, H)) return false;
    if (halo_math_mode() != HALO_MATH_BF16 || L < 2) return false;       // the top two layers of the stack; the ones below run per layer
    if (H % 128 != 0 || H > 1024 || T < 1) return false;
    // image byte offsets are 32-bit (buffer addressing): the largest is the backward's, (T + 1) images of ceil(B/16) * 4H/32 blocks
    const long nbt = (B + 15) / 16;
    if ((long)(T + 2) * nbt * (4 * H / 32) * 2048 >= (1L << 31)) return false;
    if ((long)(T + 1) * B * 4 * H >= (1L << 31)) return false;             // 32-bit element indices inside the kernels
    // the epoch words of a launch's tiles (local index) fit a replica
    return H / 16 <= cu_count2() && PERSIST_FLAG_HEADER + 2L * tiles_per_launch(H) * (H / 16) <= PERSIST_REPLICA_WORDS;      // (two tiles per workgroup: lstm_persist2x.hip)
}

// The batch rows are independent chains: a batch of more 16-row tiles than the chip holds workgroups for runs as consecutive launches over
// the same buffers, each on its own tiles (B = 128 at H = 1024: two launches of 256 workgroups).  The epochs of launch g start behind
// those of launch g - 1, so the (once zeroed) epoch words need no clearing between them.
// With halo_set_lstm_interleave (default on) a batch of more tiles than one launch holds runs TWO tiles per workgroup, interleaved
// (lstm_persist2x.hip): B = 128 at H = 1024 is one launch of 256 workgroups, each hiding one tile's hand-off behind the other tile's step.
inline bool interleave_on() {
    static const bool env_off = getenv("HALO_LSTM_INTERLEAVE") && atoi(getenv("HALO_LSTM_INTERLEAVE")) == 0;
    return !env_off && halo_ctx_cur().lstm_interleave != 0;
}

template <typename A, typename F, typename FX>
int launch_groups(const A &a0, int steps, F launch_one, FX launch_pairs) {
    const int nbt = (a0.B + 15) / 16, per = tiles_per_launch(a0.H);
    // (the interleaved kernels address the saved activations through 2 GiB buffer resources: a longer array keeps the consecutive launches)
    const bool pairs = nbt > per && interleave_on() && (long)(a0.T + 1) * a0.B * 4 * a0.H * (long)sizeof(float) < (1L << 31);
    for (int bt0 = 0, g = 0; bt0 < nbt; ++g) {
        A a = a0;
        const int left = nbt - bt0;
        a.bt0 = bt0; a.epoch0 = (unsigned)g * (unsigned)(steps + 2);
        int rc;
        // every workgroup of an interleaved launch has TWO tiles (an odd last tile: a plain launch), and it pays only when it holds more
        // tiles than a plain launch would (5 tiles at H = 1024: a plain launch of 4, then one of 1 -- not 2 pairs on half the chip)
        const int two = min(2 * per, left) & ~1;
        if (pairs && two > per) {
            a.nbt = two;
            rc = launch_pairs(a);
        } else {
            a.nbt = min(per, left);
            rc = launch_one(a, (a.H / 16) * a.nbt);
        }
        if (rc != HALO_OK) return rc;
        bt0 += a.nbt;
    }
    return HALO_OK;
}

int halo_lstm_persist2_fwd(const Persist2Fwd &a0, hipStream_t st) {
    const int kbq = a0.H / 128;
    const size_t dyn = (size_t)4 * 1024 * (4 * kbq - (kbq >= 8 ? 2 : 0));      // 4 quarters x NWLDS fragments of 1 KiB
    return launch_groups(a0, a0.T + 2, [&](const Persist2Fwd &a, int blocks) {
        switch (kbq) {
            case 2: return launch2(lstm_persist2_fwd_kernel<2>, a, blocks, dyn, st);
            case 4: return launch2(lstm_persist2_fwd_kernel<4>, a, blocks, dyn, st);
            case 6: return launch2(lstm_persist2_fwd_kernel<6>, a, blocks, dyn, st);
            case 8: return launch2(lstm_persist2_fwd_kernel<8>, a, blocks, dyn, st);
            default: return (int)HALO_ENOTSUP;
        }
    }, [&](const Persist2Fwd &a) { return halo_lstm_persist2x_fwd(a, st); });
}

int halo_lstm_persist2_bwd(const Persist2Bwd &a0, hipStream_t st) {
    const int kc = a0.H / 128;
    const size_t dyn = (size_t)4 * 4 * kc * 1024;     // 4 quarters x KBW fragments of 1 KiB
    return launch_groups(a0, a0.T + 1, [&](const Persist2Bwd &a, int blocks) {
        switch (kc) {
            case 2: return launch2(lstm_persist2_bwd_kernel<2>, a, blocks, dyn, st);
            case 4: return launch2(lstm_persist2_bwd_kernel<4>, a, blocks, dyn, st);
            case 6: return launch2(lstm_persist2_bwd_kernel<6>, a, blocks, dyn, st);
            case 8: return launch2(lstm_persist2_bwd_kernel<8>, a, blocks, dyn, st);
            default: return (int)HALO_ENOTSUP;
        }
    }, [&](const Persist2Bwd &a) { return halo_lstm_persist2x_bwd(a, st); });
}
