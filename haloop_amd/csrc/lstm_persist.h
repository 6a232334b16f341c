// Host-side interface of the weight-resident persistent LSTM recurrence (lstm_persist.hip); not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include "halo_common.h"

constexpr int PERSIST_FLAG_HEADER = 16;                  // words: [0] abort / timeout word (replica 0), rest reserved
constexpr int PERSIST_MAX_BLOCKS = 256;                  // one epoch word per workgroup
// The epoch words exist in PERSIST_REPLICAS copies, 4 KiB apart (different memory channels): a producer stores its epoch to every
// copy with ONE wave instruction (lane r -> copy r), a consumer polls the copy that answers its XCD fastest (MI355X_MICROARCH.md,
// "Valid forms", second table row: a counter kept in R replicas, each on lines of its own).
constexpr int PERSIST_REPLICAS = 8;
constexpr int PERSIST_REPLICA_WORDS = 1024;              // 4 KiB
constexpr size_t PERSIST_FLAG_BYTES = (size_t)PERSIST_REPLICAS * PERSIST_REPLICA_WORDS * sizeof(unsigned);

struct PersistFwd {
    const char *wp;      // packed W_hh: tile (jt*4 + gate), H/32 k-blocks of 2 KiB (hi | lo), lstm.hip Packed<true>
    char *hp;            // packed h images [T+1][ceil(B/16)][H/32] blocks; image 0 = initial state, image t+1 written by step t
    float *gates;        // [T][B][4H] in: x W_ih^T + b_ih + b_hh ; out: activated i,f,g,o
    float *h;            // [T+1][B][H] row-major, row 0 = initial state (read by the weight-gradient GEMM)
    float *c;            // [T+1][B][H]
    float *y;            // optional second output of h_t (next layer's input / features), may be NULL with y_mode 0
    long y_stride_t, y_stride_b;
    int y_mode;          // 0 none, 1 plain, 2 relu, 3 dropout
    DropoutCfg drop;
    unsigned *flags;     // PERSIST_FLAG_BYTES, zeroed by the launch ahead of this one (lstm.hip, persist_prologue_kernel)
    unsigned long long *stamps;   // diagnostic: [block][T][16] s_memrealtime stamps (halo_lstm_persist_stamps), normally NULL
    int poll_mode;       // set by the launcher
    int replica_shift;   // copy of the epoch words polled = (XCC id + replica_shift) % PERSIST_REPLICAS
    int nap;             // s_sleep between polls: 0 none, 1 / 2 / 3 = s_sleep 1 / 4 / 16
    int T, B, H;
};

struct PersistBwd {
    const char *wpT;     // packed W_hh^T: tile jt, 4H/32 k-blocks
    char *dgp;           // packed gate-gradient images [T][ceil(B/16)][4H/32] blocks; image t written by the step of time t
    float *gates;        // [T][B][4H] in: activated gates ; out: gradients w.r.t. the pre-activations
    const float *c;      // [T+1][B][H]
    float *dc;           // [B][H] final cell-gradient carry (may be NULL)
    const float *dy;     // gradient arriving from above, dy[t*stride_t + b*stride_b + j], may be NULL
    long dy_stride_t, dy_stride_b;
    int dy_relu;
    const float *dhinit, *dcinit;   // [B][H] added at t = T-1, may be NULL
    unsigned *flags;
    unsigned long long *stamps;
    // optional: the split-bf16 tiled GEMM images of the gate gradients, written by the chain itself as each step's gradients leave the
    // cell update (gemm_bf16x3.hip layout: [row tile 128][k tile 32][hi | lo][64-byte rows, 16-byte chunks XOR-swizzled]), so that the
    // operand-image launch behind the chain need not read the fp32 gradients again: img_rows = dG as [T*B rows][4H] (the A operand of
    // the input gradient), img_cols = dG^T as [4H rows][T*B] (both weight gradients).  Either may be NULL.  Needs B % 32 == 0; rows of
    // img_rows past T*B must have been zeroed by the launch ahead (lstm.hip).
    char *img_rows, *img_cols;
    // optional: this workgroup's share of the bias gradient, sum over its 16 batch rows and all T of dG, to bias_part[bt][4H] (the
    // batch tiles are summed by the launch behind the chain, lstm.hip): the column sums of dG without re-reading it
    float *bias_part;
    int poll_mode;
    int replica_shift, nap;
    int T, B, H;
};

unsigned long long *halo_lstm_persist_stamp_buffer();   // NULL unless a diagnostic buffer was set (halo_lstm_persist_stamps)
bool halo_lstm_persist_ok(int B, int H);        // shape, arithmetic mode, CU count, switch
void halo_lstm_persist_enable(int on);
int halo_lstm_persist_fwd(const PersistFwd &a, hipStream_t st);
int halo_lstm_persist_bwd(const PersistBwd &a, hipStream_t st);
