// Weight-resident persistent LSTM recurrence for gfx950 ("step-fused persistent LSTM"; replaces the
// T dependent step launches of one nn.LSTM layer, ha/rnn.py:25, by ONE launch per layer and direction).
//
// Decomposition.  Workgroup (jt, bt) owns hidden units [16 jt, 16 jt + 16) of batch rows [16 bt, 16 bt + 16): the same
// 16 x 16 x 4-gate tile as the step kernels of lstm.hip, so (H/16) x ceil(B/16) workgroups (256 at H=1024, B=64: one per
// CU).  What is new is that the workgroup lives for all T steps and keeps its slice of the recurrent matrix IN REGISTERS:
// 64 gate rows x H (forward) or 16 columns x 4H (backward) as split bf16 hi|lo MFMA fragments = 256 KiB at H=1024, i.e.
// 128 VGPRs in each of 8 waves (wave w holds the k-blocks of K-eighth w).  W_hh is therefore read from HBM once per
// pass instead of once per time step (SURVEY.md 8d counts parameters once per pass), and a step moves only activations.
//
// The recurrence couples workgroups: step t needs ALL hidden units of h_{t-1} (forward) / all 4H gate gradients of step
// t+1 (backward) -- but only of the workgroup's own 16 batch rows.  So the exchange is not chip-wide: the (H/16) workgroups
// of one batch tile form a group (64 workgroups at H=1024), and block ids are mapped so that a group sits on as few XCDs
// as the dispatcher's round-robin allows (speed only).  Hand-off protocol (cdna_hip_programming.md Guideline 16, recipe
// R1, and MI355X_MICROARCH.md "Valid forms", first table row): the producer writes its piece of the packed operand image
// of step t with 16-byte WRITE-THROUGH (sc1) stores, every storing wave drains them (s_waitcnt vmcnt(0)), then ONE lane
// stores the workgroup's epoch word (agent-scope relaxed atomic store = sc1); a consumer polls its group's epoch words
// with one 64-lane sc1 load (lane l reads producer l's word), joins a workgroup barrier, and then every wave reads its
// fragments with sc1 buffer loads straight into registers (they bypass the CU's L1, so no acquire fence is needed).
// Images of different steps are different buffers and epochs only grow, so nothing is ever overwritten while it may
// still be read.  Every spin is bounded by the 100 MHz real-time counter: on a timeout the workgroup raises the abort
// word and leaves, its peers time out the same way, the grid drains (the host reads the word: halo_lstm_persist_status).
// All workgroups must be co-resident: the host launches at most one workgroup per CU (the LDS request forces 1/CU).
#include <stdlib.h>
#include "halo_common.h"
#include "halo_internal.h"
#include "lstm_persist.h"
#include "lstm_persist_dev.h"

namespace {

// ================================================================================================================
// forward
// ================================================================================================================
// KBW: k-blocks (32 deep) per wave = H / 256.  ONE: single-pass bf16 (only the hi halves are held and multiplied).
template <int KBW, bool ONE>
__global__ __launch_bounds__(512, 2) void lstm_persist_fwd_kernel(const PersistFwd p) {
    __shared__ float red[NWAVE][4][256];     // partial gate sums of the 8 K-slices
    __shared__ __attribute__((aligned(16))) float hbuf[16][16];
    __shared__ int s_abort;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = p.H, B = p.B, T = p.T;
    const int NJ = H / 16, NBT = (B + 15) / 16, nkb = H / 32;
    int jt, bt;
    map_block(blockIdx.x, gridDim.x, NJ, NBT, jt, bt);
    const int j0 = jt * 16;

    // this wave's slice of W_hh: gates 0..3, k-blocks [wave*KBW, wave*KBW + KBW)
    bf16x8 wh[4][KBW], wl[4][KBW];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int i = 0; i < KBW; ++i) {
            const char *src = p.wp + (((long)jt * 4 + g) * nkb + wave * KBW + i) * 2048 + lane * 16;
            wh[g][i] = *reinterpret_cast<const bf16x8 *>(src);
            if (!ONE) wl[g][i] = *reinterpret_cast<const bf16x8 *>(src + 1024);
        }

    const int ci = tid >> 4, cj = tid & 15;             // cell threads: tid < 256 -> (batch row, hidden unit) of the tile
    const int b = bt * 16 + ci;
    const bool cell = tid < 256 && b < B;
    const long BH = (long)B * H;
    const long e0 = (long)b * H + j0 + cj;
    float cst = cell ? p.c[e0] : 0.f;                   // c_{t-1}, carried in a register for the whole sequence
    float gin[4] = {0.f, 0.f, 0.f, 0.f};
    if (cell) {
#pragma unroll
        for (int g = 0; g < 4; ++g) gin[g] = p.gates[(long)b * 4 * H + (long)g * H + j0 + cj];
    }
    const __amdgpu_buffer_rsrc_t hp_rsrc = make_rsrc(p.hp);
    // The copy of the epoch words this workgroup polls, by XCD.  Measured (tools/persist_stamps.py): a copy in the poller's half of
    // the package's memory shows a new epoch ~0.8 us sooner than one in the other half, pages p and p + 2 lie in different halves,
    // and which half a page is in depends on the physical pages behind the buffer.  With an ODD shift the two XCDs of a batch group
    // (XCC 2g, 2g + 1 under round-robin dispatch) poll pages of different halves, so every group runs at the same, middle speed
    // (4.5 us per step at H=1024, B=64) instead of two groups at 4.0 and two at 5.0 -- and the slowest group sets the chain time.
    const int my_replica = (xcc_id() + p.replica_shift) % PERSIST_REPLICAS;
    const unsigned *grp_flags = p.flags + my_replica * PERSIST_REPLICA_WORDS + PERSIST_FLAG_HEADER + bt * NJ;
    if (tid == 0) s_abort = 0;
    if (p.stamps && tid == 0) p.stamps[((long)blockIdx.x * T + 0) * 16 + 14] = my_replica;
    if (p.stamps && tid == 0) p.stamps[((long)blockIdx.x * T + 0) * 16 + 15] = xcc_id();
    for (int t = 0; t < T; ++t) {
        if (wave == 0) stamp(p.stamps, T, t, 0, lane);
        if (wave == 5) stamp(p.stamps, T, t, 8, lane);
        // ---- the pieces of image t (= h_{t-1}) this wave contracts: k-blocks [wave*KBW, +KBW) = hidden tiles [2*wave*KBW, +2*KBW).
        //      Every wave polls for itself (its own loads follow its own matched poll); a timeout is agreed on at barrier (B). ----
        // poll_mode 2: every wave polls for itself; 0 / 1: ONE otherwise idle wave polls the whole group (with / without a nap between
        // polls) and releases the others at a barrier -- fewer pollers load the fabric less (MI355X_MICROARCH.md, polling-cost)
        bool ok = true;
        if (t > 0) {
            if (p.poll_mode == 2) ok = poll_group(grp_flags, 2 * wave * KBW, 2 * KBW, (unsigned)t, lane, 0);
            else if (wave == 5) ok = poll_group(grp_flags, 0, NJ, (unsigned)t, lane, p.nap);
        }
        if (!ok && lane == 0) {
            s_abort = 1;
            raise_abort(p.flags, p.status);
        }
        if (wave == 5) stamp(p.stamps, T, t, 9, lane);
        if (p.poll_mode != 2) {
            lds_barrier();                                                         // (A)
            if (s_abort) return;
        }
        if (wave == 0) stamp(p.stamps, T, t, 1, lane);
        // ---- this wave's fragments of the group's h_{t-1} tile, L1-bypassing ----
        const int img = (int)((((long)t * NBT + bt) * nkb + wave * KBW) * 2048) + lane * 16;
        bf16x8 ah[KBW], al[KBW];
#pragma unroll
        for (int i = 0; i < KBW; ++i) {
            ah[i] = load_sc1(hp_rsrc, img + i * 2048);
            if (!ONE) al[i] = load_sc1(hp_rsrc, img + i * 2048 + 1024);
        }
        __builtin_amdgcn_sched_barrier(0);               // every fragment load is in flight before the first MFMA waits
        f32x4 acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < KBW; ++i) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (!ONE) {
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], wh[g][i], acc[g], 0, 0, 0);
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], wl[g][i], acc[g], 0, 0, 0);
                }
                acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], wh[g][i], acc[g], 0, 0, 0);
            }
        }
        {   // D layout: col = lane & 15 (hidden unit), row = 4 (lane >> 4) + reg (batch row)
            const int r = lane & 15, q = lane >> 4;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) red[wave][g][(4 * q + e) * 16 + r] = acc[g][e];
        }
        if (wave == 0) stamp(p.stamps, T, t, 2, lane);
        if (wave == 7) stamp(p.stamps, T, t, 10, lane);
        lds_barrier();                                                             // (B)
        if (s_abort) return;
        if (wave == 0) stamp(p.stamps, T, t, 3, lane);
        float ig = 0.f, fg = 0.f, gg = 0.f, og = 0.f, h = 0.f;
        if (tid < 256) {
            if (cell) {
                float pre[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float s = 0.f;
#pragma unroll
                    for (int k = 0; k < NWAVE; ++k) s += red[k][g][tid];
                    pre[g] = s + gin[g];
                }
                ig = fast_sigmoid(pre[0]); fg = fast_sigmoid(pre[1]); gg = fast_tanh(pre[2]); og = fast_sigmoid(pre[3]);
                cst = fg * cst + ig * gg;
                h = og * fast_tanh(cst);
            }
            hbuf[ci][cj] = h;                            // rows >= B: zeros
        }
        if (wave == 0) stamp(p.stamps, T, t, 4, lane);
        lds_barrier();                                                             // (C)
        if (wave == 4) stamp(p.stamps, T, t, 5, lane);
        if (wave == 4) {
            // packed image of h_t: this tile is k-groups (2 (jt & 1)) and (2 (jt & 1) + 1) of k-block jt / 2, hi part | lo part
            const int part = lane >> 5, kg = (lane >> 4) & 1, row = lane & 15;
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = hbuf[row][kg * 8 + e];
            bf16x8 hi, lo;
            split8(x, hi, lo);
            const int dst = (int)((((long)(t + 1) * NBT + bt) * nkb + (jt >> 1)) * 2048) + part * 1024 +
                            (((jt & 1) * 2 + kg) * 16 + row) * 16;
            store_sc1(hp_rsrc, dst, part ? lo : hi);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // the write-through stores have left
            stamp(p.stamps, T, t, 6, lane);
            if ((int)blockIdx.x != p.mute) publish_epoch(p.flags, bt * NJ + jt, (unsigned)(t + 1), lane);
            stamp(p.stamps, T, t, 7, lane);
        }
        if (cell) {
            float *gp = p.gates + ((long)t * B + b) * 4 * H + j0 + cj;
            gp[0] = ig; gp[H] = fg; gp[2 * (long)H] = gg; gp[3 * (long)H] = og;
            p.c[(long)(t + 1) * BH + e0] = cst;
            p.h[(long)(t + 1) * BH + e0] = h;
            if (p.y_mode != Y_NONE) {
                float v = h;
                if (p.y_mode == Y_RELU) v = fmaxf(h, 0.f);
                else if (p.y_mode == Y_DROPOUT) v = h * dropout_mult(p.drop, (uint64_t)t * BH + (uint64_t)e0);
                p.y[(long)t * p.y_stride_t + (long)b * p.y_stride_b + j0 + cj] = v;
            }
            if (t + 1 < T) {
#pragma unroll
                for (int g = 0; g < 4; ++g) gin[g] = p.gates[((long)(t + 1) * B + b) * 4 * H + (long)g * H + j0 + cj];
            }
        }
    }
}

// ================================================================================================================
// backward
// ================================================================================================================
// KC: chunks of 4 k-blocks per wave = (4H / 32 / 8) / 4 = H / 256.
template <int KC, bool ONE>
__global__ __launch_bounds__(512, 2) void lstm_persist_bwd_kernel(const PersistBwd p) {
    __shared__ float red[NWAVE][256];
    __shared__ __attribute__((aligned(16))) float dgbuf[4][16][16];
    __shared__ int s_abort;
    __shared__ unsigned s_published;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = p.H, B = p.B, T = p.T, K = 4 * H;
    const int NJ = H / 16, NBT = (B + 15) / 16, nkb4 = K / 32;
    constexpr int KBW = 4 * KC;                           // k-blocks per wave
    int jt, bt;
    map_block(blockIdx.x, gridDim.x, NJ, NBT, jt, bt);
    const int j0 = jt * 16;

    // this wave's slice of W_hh^T: columns j0..j0+15, k-blocks [wave*KBW, wave*KBW + KBW) of the 4H-deep contraction
    bf16x8 wh[KBW], wl[KBW];
#pragma unroll
    for (int i = 0; i < KBW; ++i) {
        const char *src = p.wpT + ((long)jt * nkb4 + wave * KBW + i) * 2048 + lane * 16;
        wh[i] = *reinterpret_cast<const bf16x8 *>(src);
        if (!ONE) wl[i] = *reinterpret_cast<const bf16x8 *>(src + 1024);
    }

    const int ci = tid >> 4, cj = tid & 15;
    const int b = bt * 16 + ci;
    const bool cell = tid < 256 && b < B;
    const long BH = (long)B * H;
    const long e0 = (long)b * H + j0 + cj;
    float gv[4] = {0.f, 0.f, 0.f, 0.f}, cc = 0.f, cprev = 0.f, dyv = 0.f, dcarry = 0.f, dh0 = 0.f;
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};                 // this thread's (batch row, hidden unit) gate gradients summed over time
    if (cell) {
        const int t = T - 1;
#pragma unroll
        for (int g = 0; g < 4; ++g) gv[g] = p.gates[((long)t * B + b) * K + (long)g * H + j0 + cj];
        cc = p.c[(long)(t + 1) * BH + e0];
        cprev = p.c[(long)t * BH + e0];
        if (p.dy) dyv = p.dy[(long)t * p.dy_stride_t + (long)b * p.dy_stride_b + j0 + cj];
        if (p.dcinit) dcarry = p.dcinit[e0];
        if (p.dhinit) dh0 = p.dhinit[e0];
    }
    const __amdgpu_buffer_rsrc_t dg_rsrc = make_rsrc(p.dgp);
    const int my_replica = (xcc_id() + p.replica_shift) % PERSIST_REPLICAS;       // see the forward kernel
    const unsigned *grp_flags = p.flags + my_replica * PERSIST_REPLICA_WORDS + PERSIST_FLAG_HEADER + bt * NJ;
    if (tid == 0) { s_abort = 0; s_published = 0; }

    for (int s = 0; s < T; ++s) {
        const int t = T - 1 - s;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (s > 0) {
            // ---- the pieces of image t+1 (gate gradients of the step done before) this wave contracts: k-blocks
            //      [wave*KBW, +KBW) of the 4H-deep contraction = gate wave/2, hidden tiles [(wave & 1) * NJ/2, +NJ/2) ----
            const int per_gate = H / 32;                       // k-blocks per gate
            const int kb0 = wave * KBW;
            const int jt_first = (kb0 % per_gate) * 2;
            bool ok = true;
            if (p.poll_mode == 2) ok = poll_group(grp_flags, jt_first, 2 * KBW > NJ ? NJ : 2 * KBW, (unsigned)s, lane, 0);
            else if (wave == 7) ok = poll_group(grp_flags, 0, NJ, (unsigned)s, lane, p.nap);
            if (!ok && lane == 0) {
                s_abort = 1;
                raise_abort(p.flags, p.status);
            }
            if (p.poll_mode != 2) {
                lds_barrier();                                                     // (A)
                if (s_abort) return;
            }
            const int img = (int)((((long)(t + 1) * NBT + bt) * nkb4 + wave * KBW) * 2048) + lane * 16;
            bf16x8 ah[2][4], al[2][4];
            auto load4 = [&](int buf, int c) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ah[buf][i] = load_sc1(dg_rsrc, img + (c * 4 + i) * 2048);
                    if (!ONE) al[buf][i] = load_sc1(dg_rsrc, img + (c * 4 + i) * 2048 + 1024);
                }
            };
            auto mma4 = [&](int buf, int c) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (!ONE) {
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[buf][i], wh[c * 4 + i], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[buf][i], wl[c * 4 + i], acc, 0, 0, 0);
                    }
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[buf][i], wh[c * 4 + i], acc, 0, 0, 0);
                }
            };
            load4(0, 0);
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                if (c + 1 < KC) load4((c + 1) & 1, c + 1);
                __builtin_amdgcn_sched_barrier(0);       // the next chunk's loads are issued before this chunk's MFMAs wait
                mma4(c & 1, c);
            }
        }
        {
            const int r = lane & 15, q = lane >> 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) red[wave][(4 * q + e) * 16 + r] = acc[e];
        }
        lds_barrier();                                                             // (B)
        if (s_abort) return;
        float dg[4] = {0.f, 0.f, 0.f, 0.f};
        if (tid < 256) {
            if (cell) {
                float dh = s == 0 ? dh0 : 0.f;
                if (s > 0) {
#pragma unroll
                    for (int k = 0; k < NWAVE; ++k) dh += red[k][tid];
                }
                const float ig = gv[0], fg = gv[1], gg = gv[2], og = gv[3];
                const float tc = fast_tanh(cc);
                if (p.dy) {
                    float d = dyv;
                    if (p.dy_relu && !(og * tc > 0.f)) d = 0.f;
                    dh += d;
                }
                const float dcc = dcarry + dh * og * (1.f - tc * tc);
                const float d_o = dh * tc;
                const float d_i = dcc * gg, d_f = dcc * cprev, d_g = dcc * ig;
                dcarry = dcc * fg;
                dg[0] = d_i * ig * (1.f - ig);
                dg[1] = d_f * fg * (1.f - fg);
                dg[2] = d_g * (1.f - gg * gg);
                dg[3] = d_o * og * (1.f - og);
#pragma unroll
                for (int g = 0; g < 4; ++g) bsum[g] += dg[g];
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) dgbuf[g][ci][cj] = dg[g];
        }
        lds_barrier();                                                             // (C)
        if (tid >= 256) {
            // packed image of dG_t: gate g's columns j0..j0+15 are k-groups (2 (jt & 1)), (2 (jt & 1) + 1) of k-block g H/32 + jt/2
            const int u = tid - 256;
            const int g = u >> 6, part = (u >> 5) & 1, kg = (u >> 4) & 1, row = u & 15;
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = dgbuf[g][row][kg * 8 + e];
            bf16x8 hi, lo;
            split8(x, hi, lo);
            const int dst = (int)((((long)t * NBT + bt) * nkb4 + g * (H / 32) + (jt >> 1)) * 2048) + part * 1024 +
                            (((jt & 1) * 2 + kg) * 16 + row) * 16;
            store_sc1(dg_rsrc, dst, part ? lo : hi);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // the last of the four storing waves to get here signals for the workgroup (counter in LDS: Guideline 16)
            unsigned old = 0;
            if (lane == 0) old = atomicAdd(&s_published, 1u);
            old = __builtin_amdgcn_readfirstlane(old);
            if (old == 4u * (unsigned)s + 3u) publish_epoch(p.flags, bt * NJ + jt, (unsigned)(s + 1), lane);
            // ---- off the hand-off path: this tile of dG_t in the two GEMM operand images (plain stores, read by later launches) ----
            // (single-pass mode reads the hi parts only: the lo halves of the threads skip the images)
            if (p.img_rows && !(ONE && part)) {           // rows t*B + b, k = g*H + j0 + cj: item (g, part, kg = which 8 of the 16 columns, row = batch row)
                const int grow = t * B + bt * 16 + row, kcol = g * H + j0 + kg * 8;
                const long blk = ((long)(grow >> 7) * nkb4 + (kcol >> 5)) * 2 + part;
                const int r = grow & 127, c = (kcol & 31) >> 3;
                *reinterpret_cast<bf16x8 *>(p.img_rows + blk * 8192 + r * 64 + ((c ^ ((r >> 2) & 3)) << 4)) = part ? lo : hi;
            }
            if (p.img_cols && !(ONE && part)) {           // rows g*H + j0 + cj, k = t*B + b: item (g, part, kg = which 8 of the 16 batch rows, row = hidden unit)
                float xt[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) xt[e] = dgbuf[g][kg * 8 + e][row];
                bf16x8 hit, lot;
                split8(xt, hit, lot);
                const int grow = g * H + j0 + row, kcol = t * B + bt * 16 + kg * 8;
                const int KT = (T * B + 31) >> 5;
                const long blk = ((long)(grow >> 7) * KT + (kcol >> 5)) * 2 + part;
                const int r = grow & 127, c = (kcol & 31) >> 3;
                *reinterpret_cast<bf16x8 *>(p.img_cols + blk * 8192 + r * 64 + ((c ^ ((r >> 2) & 3)) << 4)) = part ? lot : hit;
            }
        }
        if (cell) {
            float *gp = p.gates + ((long)t * B + b) * K + j0 + cj;
            if (!p.skip_dg) { gp[0] = dg[0]; gp[H] = dg[1]; gp[2 * (long)H] = dg[2]; gp[3 * (long)H] = dg[3]; }
            if (t > 0) {
#pragma unroll
                for (int g = 0; g < 4; ++g) gv[g] = p.gates[((long)(t - 1) * B + b) * K + (long)g * H + j0 + cj];
                cc = cprev;
                cprev = p.c[(long)(t - 1) * BH + e0];
                if (p.dy) dyv = p.dy[(long)(t - 1) * p.dy_stride_t + (long)b * p.dy_stride_b + j0 + cj];
            }
        }
    }
    if (cell && p.dc) p.dc[e0] = dcarry;
    if (p.bias_part) {          // sum over the tile's 16 batch rows (fixed order), one value per (gate, hidden unit) of the workgroup
        lds_barrier();
        if (tid < 256) {
#pragma unroll
            for (int g = 0; g < 4; ++g) dgbuf[g][ci][cj] = bsum[g];       // rows >= B hold zeros
        }
        lds_barrier();
        if (tid < 64) {
            const int g = tid >> 4, j = tid & 15;
            float sum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) sum += dgbuf[g][r][j];
            p.bias_part[(long)bt * K + (long)g * H + j0 + j] = sum;
        }
    }
}

int g_cu_count = 0;
inline int poll_mode() {
    static const int m = getenv("HALO_PERSIST_POLL") ? atoi(getenv("HALO_PERSIST_POLL")) : 0;
    return m;
}

inline int cu_count() {
    if (!g_cu_count) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) g_cu_count = prop.multiProcessorCount;
        else g_cu_count = -1;
    }
    return g_cu_count;
}

constexpr size_t FORCE_ONE_PER_CU_LDS = 64 * 1024;    // dynamic LDS request on top of the static arrays: one workgroup per CU

inline void set_mute(PersistFwd &a) { a.mute = halo_ctx_cur().mute_block; }
inline void set_mute(PersistBwd &) {}

template <typename K, typename A>
int launch_persist(K kernel, const A &a0, int blocks, hipStream_t st) {
    static_assert(sizeof(A) <= 4096, "kernel arguments");
    A a = a0;
    a.poll_mode = poll_mode();
    static const int shift = getenv("HALO_PERSIST_REPLICA_SHIFT") ? atoi(getenv("HALO_PERSIST_REPLICA_SHIFT")) : 3;
    static const int nap = getenv("HALO_PERSIST_NAP") ? atoi(getenv("HALO_PERSIST_NAP")) : 2;
    a.replica_shift = shift; a.nap = nap;
    a.status = halo_ctx_cur().status;
    set_mute(a);
    // a.flags is zeroed by the caller's prologue launch (lstm.hip, persist_prologue_kernel)
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(512), FORCE_ONE_PER_CU_LDS, st, a);
    return halo_launch_status();
}

template <typename K>
int allow_lds(K kernel) {
    return hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FORCE_ONE_PER_CU_LDS) == hipSuccess
               ? HALO_OK : HALO_ELAUNCH;
}

}  // namespace

void halo_lstm_persist_enable(int on) { halo_ctx_cur().lstm_persistent = on ? 1 : 0; }
unsigned long long *halo_lstm_persist_stamp_buffer() { return halo_ctx_cur().stamps; }
extern "C" int halo_debug_mute_workgroup(int block) {
    halo_ctx_cur().mute_block = block;
    return HALO_OK;
}
extern "C" int halo_lstm_persist_stamps(void *buf) {
    halo_ctx_cur().stamps = (unsigned long long *)buf;
    return HALO_OK;
}

// image byte offsets inside the kernels are 32-bit (buffer addressing): the largest is the backward's, T images of ceil(B/16) * 4H/32 blocks
bool halo_lstm_persist_fits(int T, int B, int H) {
    const long nbt = (B + 15) / 16;
    return (long)(T + 1) * nbt * (4 * H / 32) * 2048 < (1L << 31) && (long)(T + 1) * B * 4 * H < (1L << 31);   // ... and 32-bit element indices
}

static bool wide_enabled() {
    static const bool off = getenv("HALO_LSTM_PERSIST32") && atoi(getenv("HALO_LSTM_PERSIST32")) == 0;
    return !off;
}

bool halo_lstm_persist_ok(int B, int H) {
    static const bool env_off = getenv("HALO_LSTM_PERSIST") && atoi(getenv("HALO_LSTM_PERSIST")) == 0;
    if (env_off || !halo_ctx_cur().lstm_persistent) return false;
    if (halo_math_mode() == HALO_MATH_F32) return false;
    // split-bf16 weight slices (64 gate rows x H x 4 bytes) fill half a CU's registers at H = 1024; single-pass bf16 slices at H = 1536
    // (the reference's wider variant, ha/init.py:171) still leave room
    if (H % 256 != 0 || B <= 0 || H > (halo_math_mode() == HALO_MATH_BF16 ? 1536 : 1024)) return false;
    const int blocks = (H / 16) * ((B + 15) / 16);
    if (blocks <= cu_count()) return true;
    // 32 batch rows per workgroup (lstm_persist32.hip): single-pass bf16 only -- the split-bf16 kernels have no registers for a second sub-tile
    return halo_math_mode() == HALO_MATH_BF16 && wide_enabled() && (H / 16) * ((B + 31) / 32) <= cu_count();
}

int halo_lstm_persist_fwd(const PersistFwd &a, hipStream_t st) {
    const int blocks = (a.H / 16) * ((a.B + 15) / 16);
    const bool one = halo_math_mode() == HALO_MATH_BF16;
    if (blocks > cu_count()) return one ? halo_lstm_persist_fwd32(a, st) : HALO_ENOTSUP;
    static bool attr = false;
    if (!attr) {
        int rc = allow_lds(lstm_persist_fwd_kernel<1, false>);
        if (!rc) rc = allow_lds(lstm_persist_fwd_kernel<2, false>);
        if (!rc) rc = allow_lds(lstm_persist_fwd_kernel<3, false>);
        if (!rc) rc = allow_lds(lstm_persist_fwd_kernel<4, false>);
        if (!rc) rc = allow_lds(lstm_persist_fwd_kernel<1, true>);
        if (!rc) rc = allow_lds(lstm_persist_fwd_kernel<2, true>);
        if (!rc) rc = allow_lds(lstm_persist_fwd_kernel<3, true>);
        if (!rc) rc = allow_lds(lstm_persist_fwd_kernel<4, true>);
        if (!rc) rc = allow_lds(lstm_persist_fwd_kernel<5, true>);
        if (!rc) rc = allow_lds(lstm_persist_fwd_kernel<6, true>);
        if (rc) return rc;
        attr = true;
    }
    switch ((a.H / 256) * 2 + (one ? 1 : 0)) {
        case 2: return launch_persist(lstm_persist_fwd_kernel<1, false>, a, blocks, st);
        case 3: return launch_persist(lstm_persist_fwd_kernel<1, true>, a, blocks, st);
        case 4: return launch_persist(lstm_persist_fwd_kernel<2, false>, a, blocks, st);
        case 5: return launch_persist(lstm_persist_fwd_kernel<2, true>, a, blocks, st);
        case 6: return launch_persist(lstm_persist_fwd_kernel<3, false>, a, blocks, st);
        case 7: return launch_persist(lstm_persist_fwd_kernel<3, true>, a, blocks, st);
        case 8: return launch_persist(lstm_persist_fwd_kernel<4, false>, a, blocks, st);
        case 9: return launch_persist(lstm_persist_fwd_kernel<4, true>, a, blocks, st);
        case 11: return launch_persist(lstm_persist_fwd_kernel<5, true>, a, blocks, st);
        case 13: return launch_persist(lstm_persist_fwd_kernel<6, true>, a, blocks, st);
        default: return HALO_ENOTSUP;
    }
}

int halo_lstm_persist_bwd(const PersistBwd &a, hipStream_t st) {
    const int blocks = (a.H / 16) * ((a.B + 15) / 16);
    const bool one = halo_math_mode() == HALO_MATH_BF16;
    if (blocks > cu_count()) return one ? halo_lstm_persist_bwd32(a, st) : HALO_ENOTSUP;
    static bool attr = false;
    if (!attr) {
        int rc = allow_lds(lstm_persist_bwd_kernel<1, false>);
        if (!rc) rc = allow_lds(lstm_persist_bwd_kernel<2, false>);
        if (!rc) rc = allow_lds(lstm_persist_bwd_kernel<3, false>);
        if (!rc) rc = allow_lds(lstm_persist_bwd_kernel<4, false>);
        if (!rc) rc = allow_lds(lstm_persist_bwd_kernel<1, true>);
        if (!rc) rc = allow_lds(lstm_persist_bwd_kernel<2, true>);
        if (!rc) rc = allow_lds(lstm_persist_bwd_kernel<3, true>);
        if (!rc) rc = allow_lds(lstm_persist_bwd_kernel<4, true>);
        if (!rc) rc = allow_lds(lstm_persist_bwd_kernel<5, true>);
        if (!rc) rc = allow_lds(lstm_persist_bwd_kernel<6, true>);
        if (rc) return rc;
        attr = true;
    }
    switch ((a.H / 256) * 2 + (one ? 1 : 0)) {
        case 2: return launch_persist(lstm_persist_bwd_kernel<1, false>, a, blocks, st);
        case 3: return launch_persist(lstm_persist_bwd_kernel<1, true>, a, blocks, st);
        case 4: return launch_persist(lstm_persist_bwd_kernel<2, false>, a, blocks, st);
        case 5: return launch_persist(lstm_persist_bwd_kernel<2, true>, a, blocks, st);
        case 6: return launch_persist(lstm_persist_bwd_kernel<3, false>, a, blocks, st);
        case 7: return launch_persist(lstm_persist_bwd_kernel<3, true>, a, blocks, st);
        case 8: return launch_persist(lstm_persist_bwd_kernel<4, false>, a, blocks, st);
        case 9: return launch_persist(lstm_persist_bwd_kernel<4, true>, a, blocks, st);
        case 11: return launch_persist(lstm_persist_bwd_kernel<5, true>, a, blocks, st);
        case 13: return launch_persist(lstm_persist_bwd_kernel<6, true>, a, blocks, st);
        default: return HALO_ENOTSUP;
    }
}
