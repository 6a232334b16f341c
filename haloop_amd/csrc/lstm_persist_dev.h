// Device-side helpers shared by the persistent LSTM recurrences (lstm_persist.hip: one layer per launch; lstm_persist2.hip: both
// layers of a 2-layer stack in one launch): write-through hand-off accesses, epoch words, block placement, the gate non-linearities.
#pragma once
#include <hip/hip_runtime.h>
#include "halo_common.h"
#include "lstm_persist.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int NWAVE = 8;
constexpr unsigned long long SPIN_TIMEOUT_TICKS = 20000000ull;   // 0.2 s of the 100 MHz s_memrealtime counter

enum YMode { Y_NONE = 0, Y_PLAIN = 1, Y_RELU = 2, Y_DROPOUT = 3 };

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, 0x7fffffff, 0x00020000);
}
// 16-byte write-through / L1-bypassing accesses (aux 16 = sc1)
__device__ __forceinline__ bf16x8 load_sc1(__amdgpu_buffer_rsrc_t r, int byte_off) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16));
}
// the same with the wave-uniform part of the offset in an SGPR (soffset): the lane part (voffset) is then ONE register for all loads
__device__ __forceinline__ bf16x8 load_sc1_u(__amdgpu_buffer_rsrc_t r, int lane_off, int uniform_off) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, lane_off, uniform_off, 16));
}
__device__ __forceinline__ void store_sc1(__amdgpu_buffer_rsrc_t r, int byte_off, bf16x8 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, byte_off, 0, 16);
}

// one float per lane through a buffer resource: the lane part of the address is a per-thread constant (voff, bytes), the part that
// moves with the time step is wave-uniform and lives in an SGPR (soff, bytes) -- no vector address arithmetic per access
// (lstm_persist2x.hip: the saved activations of a phase; the arrays must be shorter than 2 GiB: halo_lstm_persist2x_ok)
__device__ __forceinline__ float load_f32_u(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void store_f32_u(__amdgpu_buffer_rsrc_t r, int voff, int soff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 0);
}

// block id -> (hidden tile, batch tile).  Blocks b and b + 8 share an XCD under the dispatcher's round-robin: when the number
// of batch tiles divides 8, group g takes the XCD labels [g * 8/NBT, (g + 1) * 8/NBT).  Any placement is CORRECT; this one
// keeps a group's 64 KB .. 256 KB per-step exchange inside one or two L2s.
__device__ __forceinline__ void map_block(int bid, int nblocks, int NJ, int NBT, int &jt, int &bt) {
    if (NBT <= 8 && 8 % NBT == 0 && nblocks % 8 == 0) {
        const int per = 8 / NBT, x = bid & 7;
        bt = x / per;
        jt = (bid >> 3) * per + (x % per);
    } else {
        jt = bid / NBT;
        bt = bid % NBT;
    }
}

// wait until producers [first, first + count) of this batch group have published epoch >= need (count <= 64); the calling wave
// polls with one load per pass, lane l reading producer first + l; the result is wave-uniform
__device__ __forceinline__ bool poll_group(const unsigned *grp_flags, int first, int count, unsigned need, int lane, int nap) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        const unsigned v = lane < count ? __hip_atomic_load(grp_flags + first + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                        : 0xffffffffu;
        if (__all(v >= need)) return true;
        if (nap == 1) __builtin_amdgcn_s_sleep(1);
        else if (nap == 2) __builtin_amdgcn_s_sleep(4);
        else if (nap == 3) __builtin_amdgcn_s_sleep(16);
        if (__builtin_amdgcn_s_memrealtime() - t0 > SPIN_TIMEOUT_TICKS) return false;
    }
}

// a bounded wait timed out: the call's own abort word (read by tests, zeroed by the next prologue) and the caller's sticky status word
__device__ __forceinline__ void raise_abort(unsigned *flags, unsigned *status) {
    __hip_atomic_store(flags, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (status) __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ int xcc_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return (int)(x & 0xf);
}
// one wave instruction publishes the epoch to every replica (lane r -> replica r)
__device__ __forceinline__ void publish_epoch(unsigned *flags, int slot, unsigned epoch, int lane) {
    if (lane < PERSIST_REPLICAS)
        __hip_atomic_store(flags + lane * PERSIST_REPLICA_WORDS + PERSIST_FLAG_HEADER + slot, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// diagnostic time stamps (100 MHz counter), one slot per (block, step, point); compiled in, dormant while p.stamps == NULL
__device__ __forceinline__ void stamp(unsigned long long *stamps, int T, int step, int point, int lane) {
    if (stamps && lane == 0) stamps[((long)blockIdx.x * T + step) * 16 + point] = __builtin_amdgcn_s_memrealtime();
}

// Gate non-linearities on the hardware exp2 / reciprocal (v_exp_f32, v_rcp_f32: 1 ulp each): the cell update sits on the serial
// path of every step.  |error| <= ~2e-7 absolute for both (tanh as 1 - 2 / (1 + e^{2x}), the forms the parity tests bound).
__device__ __forceinline__ float fast_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float fast_tanh(float x) {
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
}

// ---- the cell updates of the two-layer launches (lstm_persist2.hip, and lstm_persist2x.hip with two tiles per workgroup): ONE definition,
// floating-point contraction off and the fused operations written out, so that both kernels round every intermediate the same way
// whatever code surrounds the call (the compiler otherwise decides per call site which a*b+c becomes an fma; the interleaved launches are
// tested BIT for bit against the consecutive ones) ----
__device__ __forceinline__ float persist2_tanh(float x) {
#pragma clang fp contract(off)
    const float r = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
    return __builtin_fmaf(-2.0f, r, 1.0f);
}
// forward: pre-activations (recurrent sums already added to the input part) -> activated gates, new cell state, h
__device__ __forceinline__ float persist2_fwd_cell(const float (&pre)[4], float &cst, float &ig, float &fg, float &gg, float &og) {
#pragma clang fp contract(off)
    ig = fast_sigmoid(pre[0]); fg = fast_sigmoid(pre[1]); gg = persist2_tanh(pre[2]); og = fast_sigmoid(pre[3]);
    cst = __builtin_fmaf(fg, cst, ig * gg);
    return og * persist2_tanh(cst);
}
// backward: dh -> gate gradients w.r.t. the pre-activations; returns the cell gradient carried to the step before
__device__ __forceinline__ float persist2_bwd_cell(const float (&gv)[4], float cc, float cprev, float dcarry, float dh, float (&dg)[4]) {
#pragma clang fp contract(off)
    const float ig = gv[0], fg = gv[1], gg = gv[2], og = gv[3];
    const float tc = persist2_tanh(cc);
    const float dcc = __builtin_fmaf(dh * og, __builtin_fmaf(-tc, tc, 1.f), dcarry);
    const float d_o = dh * tc;
    const float d_i = dcc * gg, d_f = dcc * cprev, d_g = dcc * ig;
    dg[0] = d_i * ig * (1.f - ig);
    dg[1] = d_f * fg * (1.f - fg);
    dg[2] = d_g * __builtin_fmaf(-gg, gg, 1.f);
    dg[3] = d_o * og * (1.f - og);
    return dcc * fg;
}
// a + b * m without contraction (layer 0's incoming gradient: recurrent term + masked gradient from the layer above)
__device__ __forceinline__ float persist2_add_masked(float a, float b, float m) {
#pragma clang fp contract(off)
    const float bm = b * m;
    return a + bm;
}

__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ void split8(const float *x, bf16x8 &hi, bf16x8 &lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const __bf16 h = (__bf16)x[e];
        hi[e] = h;
        lo[e] = (__bf16)(x[e] - (float)h);
    }
}


}  // namespace
