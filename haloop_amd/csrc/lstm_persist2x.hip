// The two-layer persistent LSTM launches of lstm_persist2.hip with TWO batch tiles per workgroup, interleaved (gfx950; ha/rnn.py:11,25;
// "larger B raises the achieved fraction").
//
// Why.  A combined step of lstm_persist2.hip is a dependent chain: publish -> the group's 64 epoch words -> poll -> fragment loads ->
// MFMA -> cell update -> publish; 1.7 of the forward's 4.3 us and ~1.5 of the backward's 5.2 us per step are the workgroup WAITING for
// the hand-off (profiles/r03_persist2_stamps.txt).  The weights in registers / LDS serve any batch rows, so a workgroup that owns two
// 16-row batch tiles (A, B) of the same 16 hidden units runs  A.step(s), B.step(s), A.step(s+1), ...: while A's pieces travel to the
// other 63 workgroups of its group, the workgroup computes B's step, and A's epoch has long matched when its turn comes again.  A
// batch of 128 rows is ONE launch of 256 workgroups instead of two launches run one after the other.
//
// What changes against lstm_persist2.hip (the protocol, buffers, images and epoch words are the same; each tile still has its own 64
// epoch words and its own s_published count).  A PHASE is one tile's combined step; a workgroup runs phases (s, A), (s, B), (s+1, A), ...
//  * what a phase needs besides the hand-off -- saved activations, pre-activations, the dropout multiplier; forward layer 1: the input half
//    dropout(h0_t) W_ih1^T, whose fragments were complete when the tile's previous poll matched -- is requested / multiplied during the
//    phase BEFORE it (the other tile's), so one set of registers carries it, not one per tile; only the cell state (forward) or the
//    carried cell gradient (backward) lives per tile;
//  * the publish is DEFERRED: a phase stores its pieces write-through and moves on -- operand images, saved gates, the next phase's
//    requests -- and the wait for those stores (s_waitcnt vmcnt(0); a counted wait would not do: stores and loads retire out of order
//    with respect to each other) and the epoch store happen at the top of the NEXT phase.  Nobody waits for this tile's epoch before the
//    other tile's phase is over, so the write-through acknowledgement (0.7-1 us at the end of every step of lstm_persist2.hip) is mostly
//    off the workgroup's path.  A launch always has an EVEN number of tiles (the launcher gives an odd last
//    tile a plain launch of its own): a workgroup with one tile would wait for its own deferred epoch;
//  * the bias-gradient sums of the two tiles are formed in one register set (written to the first tile's row of bias_part, zeros to the
//    second's: the launch behind the chain adds the rows).
#include <stdlib.h>
#include "halo_common.h"
#include "halo_internal.h"
#include "lstm_persist.h"
#include "lstm_persist_dev.h"

namespace {

constexpr int NG = 2;                                // batch tiles per workgroup

// ================================================================================================================
// forward
// ================================================================================================================
struct Fwd2xShared {
    float (*red)[4][4][256];          // [layer][K-quarter][gate][batch row * 16 + hidden unit]
    float (*hbuf)[16][16];            // h0_t, dropout(h0_t), h1_t of the tile in hand
    int *s_abort;
    unsigned *s_published;            // [NG]
};

// a wave's deferred publish: its pieces of (tile q, epoch) are stored but not yet known to have left (wave-uniform: kept in SGPRs)
struct Pending {
    int on, q, slot;
    unsigned target, epoch;
    __device__ __forceinline__ void set(int q_, int slot_, unsigned target_, unsigned epoch_) {
        on = 1;
        q = __builtin_amdgcn_readfirstlane(q_);
        slot = __builtin_amdgcn_readfirstlane(slot_);
        target = (unsigned)__builtin_amdgcn_readfirstlane((int)target_);
        epoch = (unsigned)__builtin_amdgcn_readfirstlane((int)epoch_);
    }
};

// wait until this wave's write-through stores have left (vmcnt(0): stores and loads share the counter but complete out of order with
// respect to each other, so no counted wait can tell them apart), then count the wave in; the wave that completes the tile's count
// for the step stores the epoch (Guideline 16, R1)
__device__ __forceinline__ void flush_pending(Pending &pd, unsigned *s_published, unsigned *flags, bool muted, int lane) {
    if (!pd.on) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned old = 0;
    if (lane == 0) old = atomicAdd(s_published + pd.q, 1u);
    old = __builtin_amdgcn_readfirstlane(old);
    if (old + 1u == pd.target && !muted) publish_epoch(flags, pd.slot, pd.epoch, lane);
    pd.on = 0;
}

// Forward, software-pipelined over the phases (a launch always has an EVEN number of tiles: every workgroup has two).  Per phase p = (s, q):
//     [fragments of p in flight]  flush(p-1) -> MFMA(p) -> partial sums            -> barrier (B)
//     wave 1 polls the epoch of phase p+1 (the OTHER tile) | cell update of p      -> barrier (AC)
//     request the fragments of p+1 -> pack and store the pieces of p (publish deferred), operand images, the input half of p+1 (layer 1)
// so the fragments' latency is covered by the pack / the input half, the poll by the cell update, and two barriers per phase remain.
template <int KBQ>
__device__ __forceinline__ void fwd2x_layer0_waves(const Persist2Fwd &p, const Fwd2xShared sh, int jt, int pr, int wave, int lane, int u) {
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
    const int BH = B * H;
    const bool muted = (int)blockIdx.x == p.mute;
    // the dropout stream with the device-side step counter read ONCE: read inside dropout_mult, every draw would be a load followed by
    // a wait for everything this wave has in flight -- the fragments it was meant to hide behind
    DropoutCfg drop = p.drop;
    if (drop.offset_dev) { drop.offset += *drop.offset_dev; drop.offset_dev = nullptr; }
    float cst[NG];
#pragma unroll
    for (int q = 0; q < NG; ++q) {
        const int b = (p.bt0 + 2 * pr + q) * 16 + ci;
        cst[q] = b < B ? p.c0[b * H + j0 + cj] : 0.f;
    }
    // the NEXT phase's pre-activations x W_ih0^T + biases (unconditional, on clamped indices: inside a branch the compiler waits for the
    // loads at the end of the branch)
    float pg[4];
    auto prefetch = [&](int ns, int nq) {
        const int b = min((p.bt0 + 2 * pr + nq) * 16 + ci, B - 1), t = min(ns, T - 1);
#pragma unroll
        for (int g = 0; g < 4; ++g) pg[g] = p.gates0[(t * B + b) * 4 * H + g * H + j0 + cj];
    };
    const __amdgpu_buffer_rsrc_t hp0_rsrc = make_rsrc(p.hp0), hp1_rsrc = make_rsrc(p.hp1);
    const __amdgpu_buffer_rsrc_t x_rsrc = p.xp ? make_rsrc(p.xp) : hp0_rsrc;
    const int my_replica = (xcc_id() + p.replica_shift) % PERSIST_REPLICAS;
    const unsigned *rep_flags = p.flags + my_replica * PERSIST_REPLICA_WORDS + PERSIST_FLAG_HEADER;
    bf16x8 ah[KBQ];                                  // the fragments of the phase in hand, requested during the phase before
    auto request = [&](int ns, int nq) {             // image ns of tile nq = h0_{ns-1}
        const int img = ((ns * NBT + p.bt0 + 2 * pr + nq) * nkb + wq * KBQ) * 2048;
#pragma unroll
        for (int i = 0; i < KBQ; ++i) ah[i] = load_sc1_u(hp0_rsrc, lane * 16, img + i * 2048);
    };
    Pending pd = {0, 0, 0, 0u, 0u};
    prefetch(0, 0);
    request(0, 0);                                   // (image 0: the initial state, written by the launch ahead)
    for (int s = 0; s <= T + 1; ++s) {
        const bool act0 = s < T, act1 = s >= 2;
#pragma unroll
        for (int q = 0; q < NG; ++q) {
            const int btl = 2 * pr + q, bt = p.bt0 + btl;
            const int b = bt * 16 + ci;
            const bool cell = b < B;
            const int e0 = b * H + j0 + cj;
            const int nq = q ^ 1, ns = q ? s + 1 : s;                    // the next phase: the other tile
            const bool nexists = ns <= T + 1;
            if (wave == 0) stamp(p.stamps, T + 2, s, q ? 9 : 0, lane);
            // the step's dropout multiplier of h0 (one Philox block per element): drawn while the fragments are on their way
            float dmul = 1.f;
            if (act0 && cell && p.xp) dmul = dropout_mult(drop, (uint64_t)((long)s * BH + e0));
            __builtin_amdgcn_sched_barrier(0);
            // the PREVIOUS phase's pieces (the other tile's): stored a fragment latency ago; its epoch is polled for behind barrier (B)
            flush_pending(pd, sh.s_published, p.flags, muted, lane);
            if (wave == 3) stamp(p.stamps, T + 2, s, q ? 13 : 4, lane);
            if (act0) {
                f32x4 acc[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < KBQ; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], wr[g][i], acc[g], 0, 0, 0);
                const int r = lane & 15, qq = lane >> 4;     // D layout: col = lane & 15 (hidden unit), row = 4 (lane >> 4) + reg (batch row)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) sh.red[0][wq][g][(4 * qq + e) * 16 + r] = acc[g][e];
            }
            lds_barrier();                                                             // (B)
            if (wave == 0) stamp(p.stamps, T + 2, s, q ? 11 : 2, lane);
            // ---- epoch ns of the NEXT phase's tile: every workgroup of its group has published h0_{ns-1}, dropout(h0_{ns-1}), h1_{ns-3}.
            //      (its pieces were published in front of this phase's MFMAs) ----
            bool ok = true;
            if (wave == 1 && nexists && ns > 0) ok = poll_group(rep_flags + (2 * pr + nq) * NJ, 0, NJ, p.epoch0 + (unsigned)ns, lane, p.nap);
            if (!ok && lane == 0) {
                *sh.s_abort = 1;
                raise_abort(p.flags, p.status);
            }
            float ig = 0.f, fg = 0.f, gg = 0.f, og = 0.f, h = 0.f, xv = 0.f;
            if (act0) {
                if (cell) {
                    float pre[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float sum = 0.f;
#pragma unroll
                        for (int k = 0; k < 4; ++k) sum += sh.red[0][k][g][u];
                        pre[g] = sum + pg[g];
                    }
                    h = persist2_fwd_cell(pre, cst[q], ig, fg, gg, og);
                    if (p.xp) xv = h * dmul;
                    float *gp = p.gates0 + (s * B + b) * 4 * H + j0 + cj;
                    gp[0] = ig; gp[H] = fg; gp[2 * H] = gg; gp[3 * H] = og;
                    p.c0[(s + 1) * BH + e0] = cst[q];
                    p.h0[(s + 1) * BH + e0] = h;
                    if (p.ydrop) p.ydrop[s * BH + e0] = xv;
                }
                sh.hbuf[0][ci][cj] = h;                      // rows >= B: zeros
                if (p.xp) sh.hbuf[1][ci][cj] = xv;
            }
            lds_barrier();                                                             // (AC)
            if (*sh.s_abort) return;
            if (wave == 0) stamp(p.stamps, T + 2, s, q ? 12 : 3, lane);
            if (nexists && ns < T) request(ns, nq);
            prefetch(ns, nq);
            if ((wave == 3 && act0) || (wave == 2 && act1)) {
                // wave 3: lanes 0-31 the piece of h0_s (image s+1 of layer 0), lanes 32-63 the piece of dropout(h0_s) (image s of xp);
                // wave 2: lanes 0-31 the piece of h1_{s-2} (image s-1 of layer 1)
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
                // through combined step s layer 0 has stored min(s + 1, T) times, layer 1 max(s - 1, 0) times
                pd.set(q, btl * NJ + jt, (unsigned)((s + 1 < T ? s + 1 : T) + (s >= 2 ? s - 1 : 0)), p.epoch0 + (unsigned)(s + 1));
            } else if (wave == 3 && !act0 && !act1) {
                publish_epoch(p.flags, btl * NJ + jt, p.epoch0 + (unsigned)(s + 1), lane);       // T = 1: neither layer has a step here, the epoch still moves
            }
            if (p.img_hT0 && ((wave == 3 && act0) || (wave == 2 && act1))) {
                // ---- off the hand-off path: the tile, transposed, in the operand images of the weight-gradient products (lstm_persist2.hip) ----
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
        }
    }
    flush_pending(pd, sh.s_published, p.flags, muted, lane);
}

template <int KBQ>
__device__ __forceinline__ void fwd2x_layer1_waves(const Persist2Fwd &p, const Fwd2xShared sh, char *wi_lds, int jt, int pr, int wave, int lane,
                                                   int u) {
    constexpr int NWREG = KBQ >= 8 ? 2 : 0;          // as lstm_persist2.hip: W_ih1's K-quarter in LDS but for NWREG fragments
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
            if (g == 3 && (i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
    const int ci = u >> 4, cj = u & 15;
    const int BH = B * H;
    float cst[NG], bias[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bias[g] = p.b_ih1[g * H + j0 + cj] + p.b_hh1[g * H + j0 + cj];
#pragma unroll
    for (int q = 0; q < NG; ++q) {
        const int b = (p.bt0 + 2 * pr + q) * 16 + ci;
        cst[q] = b < B ? p.c1[b * H + j0 + cj] : 0.f;
    }
    const __amdgpu_buffer_rsrc_t hp1_rsrc = make_rsrc(p.hp1);
    const __amdgpu_buffer_rsrc_t x_rsrc = make_rsrc(p.xp ? p.xp : p.hp0);
    f32x4 xacc[4];                                   // the phase's accumulators, seeded with its input half dropout(h0_t) W_ih1^T during the phase before
#pragma unroll
    for (int g = 0; g < 4; ++g) xacc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 ah[KBQ];                                  // the recurrent fragments of the phase in hand (image t of layer 1 = h1_{t-1}), requested behind barrier (AC)
    bf16x8 ax[KBQ];                                  // the input fragments of the NEXT phase (complete since that tile's previous poll), requested behind barrier (B)
    for (int s = 0; s <= T + 1; ++s) {
        const bool act1 = s >= 2;
        const int t = s - 2;
#pragma unroll
        for (int q = 0; q < NG; ++q) {
            const int bt = p.bt0 + 2 * pr + q;
            const int b = bt * 16 + ci;
            const bool cell = b < B;
            const int e0 = b * H + j0 + cj;
            const int nq = q ^ 1, ns = q ? s + 1 : s;
            const bool nact1 = ns >= 2 && ns <= T + 1;
            const int nbt = p.bt0 + 2 * pr + nq;
            if (act1) {
#pragma unroll
                for (int i = 0; i < KBQ; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g) xacc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], wr[g][i], xacc[g], 0, 0, 0);
                const int r = lane & 15, qq = lane >> 4;
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) sh.red[1][wq][g][(4 * qq + e) * 16 + r] = xacc[g][e];
            }
            lds_barrier();                                                             // (B)
            if (nact1) {
                // image ns-2 of xp (without dropout: image ns-1 of layer 0): on its way during this phase's cell update
                const int ximg = (((p.xp ? ns - 2 : ns - 1) * NBT + nbt) * nkb + wq * KBQ) * 2048;
#pragma unroll
                for (int i = 0; i < KBQ; ++i) ax[i] = load_sc1_u(x_rsrc, lane * 16, ximg + i * 2048);
            }
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
                    h = persist2_fwd_cell(pre, cst[q], ig, fg, gg, og);
                    float *gp = p.gates1 + (t * B + b) * 4 * H + j0 + cj;
                    gp[0] = ig; gp[H] = fg; gp[2 * H] = gg; gp[3 * H] = og;
                    p.c1[(t + 1) * BH + e0] = cst[q];
                    p.h1[(t + 1) * BH + e0] = h;
                    if (p.y_mode != 0) p.y[(long)t * p.y_stride_t + (long)b * p.y_stride_b + j0 + cj] = p.y_mode == 2 ? fmaxf(h, 0.f) : h;
                }
                sh.hbuf[2][ci][cj] = h;
            }
            lds_barrier();                                                             // (AC)
            if (*sh.s_abort) return;
#pragma unroll
            for (int g = 0; g < 4; ++g) xacc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (nact1) {
                const int img = (((ns - 2) * NBT + nbt) * nkb + wq * KBQ) * 2048;      // image ns-2 of layer 1 = h1_{ns-3}: the next phase's poll has matched
#pragma unroll
                for (int i = 0; i < KBQ; ++i) ah[i] = load_sc1_u(hp1_rsrc, lane * 16, img + i * 2048);
                __builtin_amdgcn_sched_barrier(0);
                // the next phase's input half, while its recurrent fragments are on their way
#pragma unroll
                for (int i = 0; i < KBQ; ++i) {
                    bf16x8 w[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        w[g] = i * 4 + g < NWREG ? wir[i * 4 + g < NWREG ? i * 4 + g : 0]
                                                 : *reinterpret_cast<const bf16x8 *>(my_wi + (i * 4 + g - NWREG) * 1024);
#pragma unroll
                    for (int g = 0; g < 4; ++g) xacc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax[i], w[g], xacc[g], 0, 0, 0);
                    if (i & 1) __builtin_amdgcn_sched_barrier(0);   // two k-blocks' LDS fragments in flight at a time
                }
            }
        }
    }
}

template <int KBQ>
__global__ __launch_bounds__(512, 2) void lstm_persist2x_fwd_kernel(const Persist2Fwd p) {
    __shared__ float red[2][4][4][256];
    __shared__ __attribute__((aligned(16))) float hbuf[3][16][16];
    __shared__ int s_abort;
    __shared__ unsigned s_published[NG];
    extern __shared__ __attribute__((aligned(16))) char wi_lds[];    // [K-quarter][NWLDS] W_ih1 fragments of 1 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int jt, pr;
    map_block(blockIdx.x, gridDim.x, p.H / 16, (p.nbt + 1) / 2, jt, pr);
    if (tid == 0) { s_abort = 0; s_published[0] = 0; s_published[1] = 0; }
    if (wave == 0) stamp(p.stamps, p.T + 2, 0, 14, lane);
    const Fwd2xShared sh = {red, hbuf, &s_abort, s_published};
    // (the first barrier of either loop orders the initialisations above before any use)
    if (wave < 4) fwd2x_layer0_waves<KBQ>(p, sh, jt, pr, wave, lane, tid & 255);
    else fwd2x_layer1_waves<KBQ>(p, sh, wi_lds, jt, pr, wave, lane, tid & 255);
    if (wave == 0) stamp(p.stamps, p.T + 2, 0, 15, lane);
}

// ================================================================================================================
// backward
// ================================================================================================================
struct Bwd2xShared {
    float (*red)[4][256];             // [layer 0 recurrent | layer 1 recurrent | from layer 1 into layer 0][K-quarter]
    float (*dgbuf)[4][16][16];        // [layer][gate][batch row][hidden unit]
    float (*dhinit)[NG][256];         // [layer][tile][cell thread]: the caller's d h_n, used by each layer's FIRST cell update only (parked here:
                                      // a load in that rare branch would make the compiler wait for every outstanding request in every phase)
    int *s_abort;
    unsigned *s_published;            // [NG]
};

// what a phase of the backward needs besides the hand-off, requested one phase ahead: the saved activations of its time step
struct Bwd2xSaved {
    float gv[4], cc, cprev, extra;                   // extra: layer 0's dropout multiplier / layer 1's dy
};

template <int KC>
__device__ __forceinline__ void bwd2x_layer0_waves(const Persist2Bwd &p, const Bwd2xShared sh, int jt, int pr, int wave, int lane, int u) {
    constexpr int KBW = 4 * KC, CH = 4, NCH = KBW / CH, NBUF = 2;     // two buffers of 4 fragments here (lstm_persist2.hip: three): the registers also carry the next phase's saved activations, and a phase no longer waits on this stream alone
    const int wq = wave & 3;
    const int H = p.H, B = p.B, T = p.T, K = 4 * H;
    const int NJ = H / 16, NBT = (B + 15) / 16, nkb4 = K / 32, j0 = jt * 16;
    bf16x8 wr[KBW];                                  // K-quarter wq of W_hh0^T: columns j0..j0+15, k-blocks [wq*KBW, +KBW)
#pragma unroll
    for (int i = 0; i < KBW; ++i) wr[i] = *reinterpret_cast<const bf16x8 *>(p.wpT0 + ((long)jt * nkb4 + wq * KBW + i) * 2048 + lane * 16);
    const int ci = u >> 4, cj = u & 15;
    const int BH = B * H;
    DropoutCfg drop = p.drop;                        // (the device-side step counter read once: see the forward)
    if (drop.offset_dev) { drop.offset += *drop.offset_dev; drop.offset_dev = nullptr; }
    const int ntile = (2 * pr + 1 < p.nbt) ? 2 : 1;
    const bool defer = ntile == 2;
    float dcarry[NG], bsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < NG; ++q) {
        const int b = (p.bt0 + 2 * pr + q) * 16 + ci;
        dcarry[q] = (q < ntile && b < B && p.dcinit0) ? p.dcinit0[b * H + j0 + cj] : 0.f;
        sh.dhinit[0][q][u] = (q < ntile && b < B && p.dhinit0) ? p.dhinit0[b * H + j0 + cj] : 0.f;      // (read back by this thread only)
    }
    Bwd2xSaved nx = {{0.f, 0.f, 0.f, 0.f}, 0.f, 0.f, 1.f};
    // (unconditional, on clamped indices: inside a branch the compiler waits for the loads at the end of the branch; a phase without a
    // cell update, or a row past the batch, does not use what it gets)
    // per-thread byte offsets of this thread's element inside one time step's [B][4H] / [B][H] slab, for either tile (rows past the batch
    // clamped: their loads are unused, their stores masked); the time step's part of an address is wave-uniform (load_f32_u)
    const __amdgpu_buffer_rsrc_t g0_rsrc = make_rsrc(p.gates0), c0_rsrc = make_rsrc(p.c0);
    int vg[NG], vc[NG];
#pragma unroll
    for (int q = 0; q < NG; ++q) {
        const int b = min((p.bt0 + 2 * pr + q) * 16 + ci, B - 1);
        vg[q] = (b * K + j0 + cj) * 4;
        vc[q] = (b * H + j0 + cj) * 4;
    }
    auto prefetch = [&](int ns, int nq) {            // layer 0's cell update of combined step ns is at time T - ns (ns >= 1)
        const int t = min(max(T - ns, 0), T - 1);
        const int sg = t * B * K * 4, sc = t * BH * 4, vgn = nq ? vg[1] : vg[0], vcn = nq ? vc[1] : vc[0];
#pragma unroll
        for (int g = 0; g < 4; ++g) nx.gv[g] = load_f32_u(g0_rsrc, vgn, sg + g * H * 4);
        nx.cc = load_f32_u(c0_rsrc, vcn, sc + BH * 4);
        nx.cprev = load_f32_u(c0_rsrc, vcn, sc);
    };
    Pending pd = {0, 0, 0, 0u, 0u};
    const __amdgpu_buffer_rsrc_t dg0_rsrc = make_rsrc(p.dgp0), dg1_rsrc = make_rsrc(p.dgp1);
    for (int s = 0; s <= T; ++s) {
        const bool act = s >= 1;                     // layer 0 has a cell update in combined step s ...
        const int t = T - s;                          // ... at this time
#pragma unroll
        for (int q = 0; q < NG; ++q) {
            if (q >= ntile) break;
            const int btl = 2 * pr + q, bt = p.bt0 + btl;
            const int b = bt * 16 + ci;
            const bool cell = b < B;
            if (wave == 0) stamp(p.stamps, T + 1, s, q ? 9 : 0, lane);
            const Bwd2xSaved cur = nx;
            lds_barrier();                                                             // (A)   (wave 5 polls: these four waves store the pieces, and a wave's poll result waits for all its older stores)
            if (*sh.s_abort) return;
            if (wave == 0) stamp(p.stamps, T + 1, s, q ? 10 : 1, lane);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            float dmul = 1.f;
            if (s >= 2) {                                 // dG0 of time t + 1 (image T-s+1): layer 0's recurrent term
                const int img = (((T - s + 1) * NBT + bt) * nkb4 + wq * KBW) * 2048;
                bf16x8 ah[NBUF][CH];
                auto loadc = [&](int buf, int cidx) {
#pragma unroll
                    for (int i = 0; i < CH; ++i) ah[buf][i] = load_sc1_u(dg0_rsrc, lane * 16, img + (cidx * CH + i) * 2048);
                };
#pragma unroll
                for (int cidx = 0; cidx < NBUF - 1; ++cidx) loadc(cidx, cidx);
                __builtin_amdgcn_sched_barrier(0);
                // layer 0's output mask at the step's time (one Philox block per element): drawn while the first fragments are on their way
                if (cell) dmul = dropout_mult(drop, (uint64_t)((long)t * BH + b * H + j0 + cj));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int cidx = 0; cidx < NCH; ++cidx) {
                    if (cidx + NBUF - 1 < NCH) loadc((cidx + NBUF - 1) % NBUF, cidx + NBUF - 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < CH; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[cidx % NBUF][i], wr[cidx * CH + i], acc, 0, 0, 0);
                }
            } else if (act && cell) {
                dmul = dropout_mult(drop, (uint64_t)((long)t * BH + b * H + j0 + cj));
            }
            {
                const int r = lane & 15, qq = lane >> 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) sh.red[0][wq][(4 * qq + e) * 16 + r] = acc[e];
            }
            // the PREVIOUS phase's pieces (the other tile's): this wave's fragment loads have all returned, so the wait is for stores that
            // have had most of a phase to leave; the tile's own next poll is still ~2 us away
            flush_pending(pd, sh.s_published, p.flags, false, lane);
            if (wave == 3) stamp(p.stamps, T + 1, s, q ? 13 : 4, lane);
            lds_barrier();                                                             // (B)
            if (wave == 0) stamp(p.stamps, T + 1, s, q ? 11 : 2, lane);
            {
                // the next phase's saved activations: a whole fragment stream ahead of their use
                const int nq = q + 1 < ntile ? q + 1 : 0;
                prefetch(nq ? s : s + 1, nq);
            }
            float dg[4] = {0.f, 0.f, 0.f, 0.f};
            if (act) {
                if (cell) {
                    float rec = 0.f, above = 0.f;
#pragma unroll
                    for (int k = 0; k < 4; ++k) { rec += sh.red[0][k][u]; above += sh.red[2][k][u]; }
                    if (s == 1) rec = sh.dhinit[0][q][u];                                 // (the first cell update: the caller's d h_n, no recurrent term yet)
                    dcarry[q] = persist2_bwd_cell(cur.gv, cur.cc, cur.cprev, dcarry[q], persist2_add_masked(rec, above, dmul), dg);   // (dmul: layer 0's own output mask)
#pragma unroll
                    for (int g = 0; g < 4; ++g) bsum[g] += dg[g];
#pragma unroll
                    for (int g = 0; g < 4; ++g) if (!p.skip_dg0) store_f32_u(g0_rsrc, vg[q], (t * B * K + g * H) * 4, dg[g]);
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) sh.dgbuf[0][g][ci][cj] = dg[g];
            }
            if (wave == 0 && q == 0) stamp(p.stamps, T + 1, s, 7, lane);              // (diagnostic: this half's arrival at barrier C)
            lds_barrier();                                                             // (C)
            if (wave == 0) stamp(p.stamps, T + 1, s, q ? 12 : 3, lane);
            {
                // pack: wave g takes gate g; lanes 0-31 layer 1's piece (time T-1-s, steps 0 .. T-1), lanes 32-63 layer 0's (time T-s, steps 1 .. T)
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
                pd.set(q, btl * NJ + jt, 4u * (unsigned)s + 4u, p.epoch0 + (unsigned)(s + 1));      // the fourth wave of the step to count in publishes
                if (!defer) flush_pending(pd, sh.s_published, p.flags, false, lane);
                // ---- off the hand-off path: the tile in the GEMM operand images (hi parts; gemm_bf16x3.hip layout) ----
                if (on) {
                    char *img_rows = lay ? nullptr : p.img_rows0;
                    char *img_cols = lay ? p.img_cols1 : p.img_cols0;
                    if (img_rows) {          // rows tt*B + b, k = g*H + j0 + 8 kg ..
                        const int grow = tt * B + bt * 16 + row, kcol = g * H + j0 + kg * 8;
                        const long blk = ((long)(grow >> 7) * nkb4 + (kcol >> 5)) * 2;
                        const int r = grow & 127, c4 = (kcol & 31) >> 3;
                        *reinterpret_cast<bf16x8 *>(img_rows + blk * 8192 + r * 64 + ((c4 ^ ((r >> 2) & 3)) << 4)) = hi;
                    }
                    if (img_cols) {          // rows g*H + j0 + row (hidden unit), k = tt*B + bt*16 + 8 kg ..
                        bf16x8 hit;
#pragma unroll
                        for (int e = 0; e < 8; ++e) hit[e] = (__bf16)sh.dgbuf[lay][g][kg * 8 + e][row];
                        const int grow = g * H + j0 + row, kcol = tt * B + bt * 16 + kg * 8;
                        const int KT = (T * B + 31) >> 5;
                        const long blk = ((long)(grow >> 7) * KT + (kcol >> 5)) * 2;
                        const int r = grow & 127, c4 = (kcol & 31) >> 3;
                        *reinterpret_cast<bf16x8 *>(img_cols + blk * 8192 + r * 64 + ((c4 ^ ((r >> 2) & 3)) << 4)) = hit;
                    }
                }
            }
        }
    }
    flush_pending(pd, sh.s_published, p.flags, false, lane);
#pragma unroll
    for (int q = 0; q < NG; ++q) {
        const int b = (p.bt0 + 2 * pr + q) * 16 + ci;
        if (q < ntile && b < B && p.dc0) p.dc0[b * H + j0 + cj] = dcarry[q];
    }
    lds_barrier();
#pragma unroll
    for (int g = 0; g < 4; ++g) sh.dgbuf[0][g][ci][cj] = bsum[g];              // rows >= B hold zeros
    lds_barrier();
    if (p.bias_part0 && u < 64) {
        const int g = u >> 4, j = u & 15;
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += sh.dgbuf[0][g][r][j];
        p.bias_part0[(long)(p.bt0 + 2 * pr) * K + g * H + j0 + j] = sum;                   // both tiles' rows summed ...
        if (ntile == 2) p.bias_part0[(long)(p.bt0 + 2 * pr + 1) * K + g * H + j0 + j] = 0.f;  // ... the second tile's row adds nothing
    }
}

template <int KC>
__device__ __forceinline__ void bwd2x_layer1_waves(const Persist2Bwd &p, const Bwd2xShared sh, char *wi_lds, int jt, int pr, int wave, int lane,
                                                   int u) {
    constexpr int KBW = 4 * KC, CH = 4, NCH = KBW / CH, NBUF = 3;
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
        if ((i & 15) == 15) __builtin_amdgcn_sched_barrier(0);
    }
    const int ci = u >> 4, cj = u & 15;
    const int BH = B * H;
    const int ntile = (2 * pr + 1 < p.nbt) ? 2 : 1;
    float dcarry[NG], bsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < NG; ++q) {
        const int b = (p.bt0 + 2 * pr + q) * 16 + ci;
        dcarry[q] = (q < ntile && b < B && p.dcinit1) ? p.dcinit1[b * H + j0 + cj] : 0.f;
        sh.dhinit[1][q][u] = (q < ntile && b < B && p.dhinit1) ? p.dhinit1[b * H + j0 + cj] : 0.f;
    }
    Bwd2xSaved nx = {{0.f, 0.f, 0.f, 0.f}, 0.f, 0.f, 0.f};
    const float *dyp = p.dy ? p.dy : p.c1;           // (no dy: a load from any valid address, unused)
    const __amdgpu_buffer_rsrc_t g1_rsrc = make_rsrc(p.gates1), c1_rsrc = make_rsrc(p.c1);
    int vg[NG], vc[NG];
    long vdy[NG];                                    // (dy has the caller's strides: plain pointer arithmetic, one load per phase)
#pragma unroll
    for (int q = 0; q < NG; ++q) {
        const int b = min((p.bt0 + 2 * pr + q) * 16 + ci, B - 1);
        vg[q] = (b * K + j0 + cj) * 4;
        vc[q] = (b * H + j0 + cj) * 4;
        vdy[q] = p.dy ? (long)b * p.dy_stride_b + j0 + cj : 0L;
    }
    const long dy_t = p.dy ? p.dy_stride_t : 0L;
    auto prefetch = [&](int ns, int nq) {            // layer 1's cell update of combined step ns is at time T - 1 - ns (ns < T)
        const int t = min(max(T - 1 - ns, 0), T - 1);
        const int sg = t * B * K * 4, sc = t * BH * 4, vgn = nq ? vg[1] : vg[0], vcn = nq ? vc[1] : vc[0];
#pragma unroll
        for (int g = 0; g < 4; ++g) nx.gv[g] = load_f32_u(g1_rsrc, vgn, sg + g * H * 4);
        nx.cc = load_f32_u(c1_rsrc, vcn, sc + BH * 4);
        nx.cprev = load_f32_u(c1_rsrc, vcn, sc);
        nx.extra = dyp[(nq ? vdy[1] : vdy[0]) + (long)t * dy_t];
    };
    prefetch(0, 0);
    const __amdgpu_buffer_rsrc_t dg1_rsrc = make_rsrc(p.dgp1);
    const int NJ = H / 16;
    const int my_replica = (xcc_id() + p.replica_shift) % PERSIST_REPLICAS;
    const unsigned *rep_flags = p.flags + my_replica * PERSIST_REPLICA_WORDS + PERSIST_FLAG_HEADER;
    for (int s = 0; s <= T; ++s) {
        const bool act = s < T;                      // layer 1 has a cell update in combined step s ...
        const int t = T - 1 - s;                      // ... at this time
#pragma unroll
        for (int q = 0; q < NG; ++q) {
            if (q >= ntile) break;
            const int bt = p.bt0 + 2 * pr + q;
            const int b = bt * 16 + ci;
            const bool cell = b < B;
            const Bwd2xSaved cur = nx;
            // ---- epoch s of THIS tile: every workgroup of its group has published dG1 of time T-s and dG0 of time T-s+1.  Polled by a
            //      layer-1 wave: it stores no pieces, and its saved-gate stores and requests are a phase old by now ----
            bool ok = true;
            if (s > 0 && wave == 5) ok = poll_group(rep_flags + (2 * pr + q) * NJ, 0, NJ, p.epoch0 + (unsigned)s, lane, p.nap);
            if (!ok && lane == 0) {
                *sh.s_abort = 1;
                raise_abort(p.abort_word, p.status);
            }
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
                const int r = lane & 15, qq = lane >> 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    sh.red[1][wq][(4 * qq + e) * 16 + r] = acc[e];
                    sh.red[2][wq][(4 * qq + e) * 16 + r] = acc2[e];
                }
            }
            lds_barrier();                                                             // (B)
            if (wave == 4 && q == 0) stamp(p.stamps, T + 1, s, 5, lane);              // (diagnostic)
            {
                const int nq = q + 1 < ntile ? q + 1 : 0;
                prefetch(nq ? s : s + 1, nq);
            }
            float dg[4] = {0.f, 0.f, 0.f, 0.f};
            if (act) {
                if (cell) {
                    float dh = 0.f;
                    if (s == 0) dh = sh.dhinit[1][q][u];
                    else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) dh += sh.red[1][k][u];
                    }
                    if (p.dy) {
                        const float tc = persist2_tanh(cur.cc);
                        float d = cur.extra;
                        if (p.dy_relu && !(cur.gv[3] * tc > 0.f)) d = 0.f;
                        dh += d;
                    }
                    dcarry[q] = persist2_bwd_cell(cur.gv, cur.cc, cur.cprev, dcarry[q], dh, dg);
#pragma unroll
                    for (int g = 0; g < 4; ++g) bsum[g] += dg[g];
#pragma unroll
                    for (int g = 0; g < 4; ++g) if (!p.skip_dg1) store_f32_u(g1_rsrc, vg[q], (t * B * K + g * H) * 4, dg[g]);
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) sh.dgbuf[1][g][ci][cj] = dg[g];
            }
            if (wave == 4 && q == 0) stamp(p.stamps, T + 1, s, 6, lane);              // (diagnostic: this half's arrival at barrier C)
            lds_barrier();                                                             // (C)
        }
    }
#pragma unroll
    for (int q = 0; q < NG; ++q) {
        const int b = (p.bt0 + 2 * pr + q) * 16 + ci;
        if (q < ntile && b < B && p.dc1) p.dc1[b * H + j0 + cj] = dcarry[q];
    }
    lds_barrier();
#pragma unroll
    for (int g = 0; g < 4; ++g) sh.dgbuf[1][g][ci][cj] = bsum[g];
    lds_barrier();
    if (p.bias_part1 && u < 64) {
        const int g = u >> 4, j = u & 15;
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += sh.dgbuf[1][g][r][j];
        p.bias_part1[(long)(p.bt0 + 2 * pr) * K + g * H + j0 + j] = sum;
        if (ntile == 2) p.bias_part1[(long)(p.bt0 + 2 * pr + 1) * K + g * H + j0 + j] = 0.f;
    }
}

template <int KC>
__global__ __launch_bounds__(512, 2) void lstm_persist2x_bwd_kernel(const Persist2Bwd p) {
    __shared__ float red[3][4][256];
    __shared__ __attribute__((aligned(16))) float dgbuf[2][4][16][16];
    __shared__ float dhinit[2][NG][256];
    __shared__ int s_abort;
    __shared__ unsigned s_published[NG];
    extern __shared__ __attribute__((aligned(16))) char wi_lds[];    // [K-quarter][4 KC] W_ih1^T fragments of 1 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int jt, pr;
    map_block(blockIdx.x, gridDim.x, p.H / 16, (p.nbt + 1) / 2, jt, pr);
    if (tid == 0) { s_abort = 0; s_published[0] = 0; s_published[1] = 0; }
    if (blockIdx.x == 0 && tid == 0 && p.abort_word != p.flags) __hip_atomic_store(p.abort_word, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (wave == 0) stamp(p.stamps, p.T + 1, 0, 14, lane);
    const Bwd2xShared sh = {red, dgbuf, dhinit, &s_abort, s_published};
    if (wave < 4) bwd2x_layer0_waves<KC>(p, sh, jt, pr, wave, lane, tid & 255);
    else bwd2x_layer1_waves<KC>(p, sh, wi_lds, jt, pr, wave, lane, tid & 255);
    if (wave == 0) stamp(p.stamps, p.T + 1, 0, 15, lane);
}

constexpr size_t MIN_DYN_LDS_X = 64 * 1024;           // with the static arrays: more than half a CU's LDS -> one workgroup per CU

inline void set_mute_x(Persist2Fwd &a) { a.mute = halo_ctx_cur().mute_block; }
inline void set_mute_x(Persist2Bwd &) {}

template <typename K, typename A>
int launch2x(K kernel, const A &a0, int blocks, size_t dyn, hipStream_t st) {
    static_assert(sizeof(A) <= 4096, "kernel arguments");
    A a = a0;
    static const int shift = getenv("HALO_PERSIST_REPLICA_SHIFT") ? atoi(getenv("HALO_PERSIST_REPLICA_SHIFT")) : 3;
    static const int nap = getenv("HALO_PERSIST_NAP") ? atoi(getenv("HALO_PERSIST_NAP")) : 2;
    a.poll_mode = 0; a.replica_shift = shift; a.nap = nap;
    a.status = halo_ctx_cur().status;
    set_mute_x(a);
    if (dyn < MIN_DYN_LDS_X) dyn = MIN_DYN_LDS_X;
    if (hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn) != hipSuccess) return HALO_ELAUNCH;
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(512), dyn, st, a);
    return halo_launch_status();
}

}  // namespace

// a.bt0 / a.nbt / a.epoch0 set by the caller (lstm_persist2.hip, launch_groups): the launch takes tiles [bt0, bt0 + nbt), two per workgroup
int halo_lstm_persist2x_fwd(const Persist2Fwd &a, hipStream_t st) {
    const int kbq = a.H / 128, blocks = (a.H / 16) * ((a.nbt + 1) / 2);
    const size_t dyn = (size_t)4 * 1024 * (4 * kbq - (kbq >= 8 ? 2 : 0));
    switch (kbq) {
        case 2: return launch2x(lstm_persist2x_fwd_kernel<2>, a, blocks, dyn, st);
        case 4: return launch2x(lstm_persist2x_fwd_kernel<4>, a, blocks, dyn, st);
        case 6: return launch2x(lstm_persist2x_fwd_kernel<6>, a, blocks, dyn, st);
        case 8: return launch2x(lstm_persist2x_fwd_kernel<8>, a, blocks, dyn, st);
        default: return (int)HALO_ENOTSUP;
    }
}

int halo_lstm_persist2x_bwd(const Persist2Bwd &a, hipStream_t st) {
    const int kc = a.H / 128, blocks = (a.H / 16) * ((a.nbt + 1) / 2);
    const size_t dyn = (size_t)4 * 4 * kc * 1024;
    switch (kc) {
        case 2: return launch2x(lstm_persist2x_bwd_kernel<2>, a, blocks, dyn, st);
        case 4: return launch2x(lstm_persist2x_bwd_kernel<4>, a, blocks, dyn, st);
        case 6: return launch2x(lstm_persist2x_bwd_kernel<6>, a, blocks, dyn, st);
        case 8: return launch2x(lstm_persist2x_bwd_kernel<8>, a, blocks, dyn, st);
        default: return (int)HALO_ENOTSUP;
    }
}
