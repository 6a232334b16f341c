// Argument blocks shared by the attention kernels of attn.hip (exact-f32 MFMA) and attn_mx.hip (split-bf16 MFMA).
#pragma once
#include <hip/hip_runtime.h>
#include "halo_common.h"

struct AttnArgs {
    const float *q, *k, *v;
    float *y, *lse, *ent;
    long q_rs, q_bs, kv_rs, kv_bs, y_rs, y_bs;   // row / batch strides in elements
    long q_hs, kv_hs;                            // head strides: HD for packed rows, Tc*HD for a [N, heads, Tc, HD] cache
    const int *key_len;                          // [N] keys >= key_len[n] are masked, may be NULL
    int Tq, Tk, heads, causal;
    float scale;
    DropoutCfg drop;                             // dropout on the attention probabilities (training), see attn_drop_index
    int use_drop;
    // attn_mx.hip: the output also as row-major bf16 (the operand of the Linear behind the attention: halo_attention_fwd_bf16)
    __bf16 *y_bf = nullptr;
    long ybf_rs = 0, ybf_bs = 0;
};

// Philox element index of attention probability (n, h, i, j): the four 16-key sub-tiles (j%64)/16 = 0..3 that one lane holds
// for a query row sit in ONE Philox group, so a 64-key tile costs a lane one Philox call per row:
//     e = ((((n*H + h)*Tq + i) * ceil(Tk/64) + j/64) * 64 + 4*(j%16) + (j%64)/16
// (oracle/transformer_ref.py: attention_dropout_mask restates it)
__device__ __forceinline__ uint64_t attn_drop_tile_base(int n, int H, int h, int Tq, int i, int KT, int kt) {
    return ((((uint64_t)n * H + h) * Tq + i) * KT + kt) * 64;
}

struct AttnBwdArgs {
    const float *q, *k, *v, *dy, *lse, *delta;
    float *dq, *dk, *dv;
    // attn_mx.hip: when y (the forward output, laid out like dy) is given, the dQ sweep computes delta = rowsum(dy * y) of its own
    // queries, uses it and writes it to delta_w for the dK/dV sweep behind it -- no separate delta launch
    const float *y;
    float *delta_w;
    long q_rs, q_bs, kv_rs, kv_bs, dy_rs, dy_bs, dq_rs, dq_bs, dkv_rs, dkv_bs;
    const int *key_len;
    int Tq, Tk, heads, causal;
    float scale;
    DropoutCfg drop;
    int use_drop;
    // attn_mx.hip: the three gradients as row-major bf16 INSTEAD of fp32 (they are only ever operands of the c_attn Linear's two
    // gradient products: halo_attention_bwd_bf16); one row / batch stride for all three
    __bf16 *dq_bf = nullptr, *dk_bf = nullptr, *dv_bf = nullptr;
    long dqb_rs = 0, dqb_bs = 0;
    int dkv_longest_first = 0;                   // attn_mx.hip, set by its launcher: grid (heads * N, key tiles), one tile per workgroup
};

// attn_mx.hip: the same products on v_mfma_f32_16x16x32_bf16 with operands split hi + lo (passes = 3) or rounded to bf16 (passes = 1)
int halo_attention_fwd_mx(const AttnArgs &a, int N, int head_dim, int passes, hipStream_t st);
int halo_attention_bwd_mx(const AttnBwdArgs &a, int N, int head_dim, int passes, hipStream_t st);
