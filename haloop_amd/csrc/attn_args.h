// Argument blocks shared by the attention kernels of attn.hip (exact-f32 MFMA) and attn_mx.hip (split-bf16 MFMA).
#pragma once
#include <hip/hip_runtime.h>

struct AttnArgs {
    const float *q, *k, *v;
    float *y, *lse, *ent;
    long q_rs, q_bs, kv_rs, kv_bs, y_rs, y_bs;   // row / batch strides in elements
    long q_hs, kv_hs;                            // head strides: HD for packed rows, Tc*HD for a [N, heads, Tc, HD] cache
    const int *key_len;                          // [N] keys >= key_len[n] are masked, may be NULL
    int Tq, Tk, heads, causal;
    float scale;
};

struct AttnBwdArgs {
    const float *q, *k, *v, *dy, *lse, *delta;
    float *dq, *dk, *dv;
    long q_rs, q_bs, kv_rs, kv_bs, dy_rs, dy_bs, dq_rs, dq_bs, dkv_rs, dkv_bs;
    const int *key_len;
    int Tq, Tk, heads, causal;
    float scale;
};

// attn_mx.hip: the same products on v_mfma_f32_16x16x32_bf16 with operands split hi + lo (passes = 3) or rounded to bf16 (passes = 1)
int halo_attention_fwd_mx(const AttnArgs &a, int N, int head_dim, int passes, hipStream_t st);
int halo_attention_bwd_mx(const AttnBwdArgs &a, int N, int head_dim, int passes, hipStream_t st);
