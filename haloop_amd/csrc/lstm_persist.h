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
    unsigned *status;    // optional caller-owned sticky word (halo_set_status_word): set to 1 when a bounded wait times out; set by the launcher
    int mute;            // test hook (halo_debug_mute_workgroup): the workgroup with this blockIdx never publishes (-1: none); set by the launcher
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
    unsigned *status;    // see PersistFwd
    unsigned long long *stamps;
    // optional: the split-bf16 tiled GEMM images of the gate gradients, written by the chain itself as each step's gradients leave the
    // cell update (gemm_bf16x3.hip layout: [row tile 128][k tile 32][hi | lo][64-byte rows, 16-byte chunks XOR-swizzled]), so that the
    // operand-image launch behind the chain need not read the fp32 gradients again: img_rows = dG as [T*B rows][4H] (the A operand of
    // the input gradient), img_cols = dG^T as [4H rows][T*B] (both weight gradients).  Either may be NULL.  Needs B % 32 == 0; rows of
    // img_rows past T*B must have been zeroed by the launch ahead (lstm.hip).
    char *img_rows, *img_cols;
    int skip_dg;                   // the fp32 gate gradients are not stored back into the gates buffer: every consumer reads the emitted images / partials
    // optional: this workgroup's share of the bias gradient, sum over its 16 batch rows and all T of dG, to bias_part[bt][4H] (the
    // batch tiles are summed by the launch behind the chain, lstm.hip): the column sums of dG without re-reading it
    float *bias_part;
    int poll_mode;
    int replica_shift, nap;
    int T, B, H;
};

unsigned long long *halo_lstm_persist_stamp_buffer();   // NULL unless a diagnostic buffer was set (halo_lstm_persist_stamps)
bool halo_lstm_persist_ok(int B, int H);        // shape, arithmetic mode, CU count, switch
bool halo_lstm_persist_fits(int T, int B, int H);   // the sequence is short enough for the kernels' 32-bit image offsets
void halo_lstm_persist_enable(int on);
int halo_lstm_persist_fwd(const PersistFwd &a, hipStream_t st);
// 32 batch rows per workgroup (lstm_persist32.hip; bf16 arithmetic only): chosen by halo_lstm_persist_fwd/bwd when the 16-row grid exceeds the CU count
int halo_lstm_persist_fwd32(const PersistFwd &a, hipStream_t st);
int halo_lstm_persist_bwd32(const PersistBwd &a, hipStream_t st);
int halo_lstm_persist_bwd(const PersistBwd &a, hipStream_t st);

// ---- both layers of a 2-layer stack in ONE persistent launch (lstm_persist2.hip; single-pass bf16 arithmetic only) ----
// Combined step s = 0 .. T runs layer 0's time step s and layer 1's time step s - 1 in the same workgroup (waves 0-3: layer 0,
// waves 4-7: layer 1), so the T + 1 combined steps need ONE hand-off each instead of the 2 T of two chains run back to back, and
// layer 1's input projection (dropout(h0_t) W_ih1^T) is part of its step: no batched GEMM, no operand images between the layers.
struct Persist2Fwd {
    const char *wp0, *wp1, *wpi;   // packed W_hh0, W_hh1, W_ih1 (lstm.hip Packed<true>, tile jt*4 + gate, H/32 blocks of 2 KiB; hi halves used)
    char *hp0, *hp1;               // packed h images of the layers, [T+1][ceil(B/16)][H/32] blocks; image 0 = initial state
    char *xp;                      // packed dropout(h0_t) images [T][ceil(B/16)][H/32] blocks, or NULL (no dropout: layer 1 reads hp0)
    float *gates0, *h0, *c0;       // layer 0: gates [T][B][4H] in: x W_ih0^T + b_ih0 + b_hh0, out: activated i,f,g,o; h, c [T+1][B][H]
    float *gates1, *h1, *c1;       // layer 1: gates out only
    const float *b_ih1, *b_hh1;
    float *ydrop;                  // [T][B][H] row-major dropout(h0_t) (operand of layer 1's weight gradient); NULL without dropout
    float *y;                      // layer 1's second output (features), may be NULL with y_mode 0
    long y_stride_t, y_stride_b;
    int y_mode;                    // 0 none, 1 plain, 2 relu
    DropoutCfg drop;               // layer 0's output dropout (the stream of lstm.hip's Y_DROPOUT epilogue)
    // optional: the tiled GEMM operand images (hi parts; gemm_bf16x3.hip layout, rows = hidden unit, k = t * B + b) that the backward's
    // weight-gradient products read: h0_{t-1}^T, h1_{t-1}^T (column block t holds the state BEFORE step t; block 0 -- the zero initial
    // state -- is cleared by the launch ahead) and dropout(h0_t)^T.  Written off the hand-off path as each step's tile leaves the cell
    // update, so the operand-image launch behind the backward chain need not read the fp32 states again.  Needs B % 32 == 0, H % 128 == 0.
    char *img_hT0, *img_hT1, *img_xT1;
    unsigned *flags;
    unsigned *status;              // see PersistFwd
    int mute;                      // see PersistFwd
    unsigned long long *stamps;
    int poll_mode, replica_shift, nap;
    int T, B, H;
    // the launch works on batch tiles [bt0, bt0 + nbt) of the ceil(B/16) (a batch larger than the chip holds workgroups for runs as several
    // launches over the same buffers); its epoch words are those of the tiles' LOCAL index and count from epoch0, so a later launch never
    // mistakes an earlier one's epochs for its own (set by halo_lstm_persist2_fwd / _bwd)
    int bt0, nbt;
    unsigned epoch0;
};

// Combined step s = 0 .. T of the backward runs layer 1's time step T-1-s and layer 0's time step T-s: the gate gradients of
// layer 1 at time T-s feed BOTH layer 1's recurrence (times W_hh1) and layer 0's incoming gradient (times W_ih1, its dropout
// mask applied), from one set of fragment loads.
struct Persist2Bwd {
    const char *wpT0, *wpT1, *wpTi;   // packed W_hh0^T, W_hh1^T, W_ih1^T (tile jt, 4H/32 blocks; hi halves used)
    char *dgp0, *dgp1;                // packed gate-gradient images per layer [T][ceil(B/16)][4H/32] blocks
    float *gates0, *gates1;           // in: activated gates, out: gradients w.r.t. the pre-activations
    const float *c0, *c1;             // [T+1][B][H]
    float *dc0, *dc1;                 // final cell-gradient carries [B][H] (may be NULL)
    const float *dy;                  // gradient w.r.t. layer 1's output, dy[t*stride_t + b*stride_b + j]
    long dy_stride_t, dy_stride_b;
    int dy_relu;
    const float *dhinit0, *dcinit0, *dhinit1, *dcinit1;   // [B][H], may be NULL
    DropoutCfg drop;                  // layer 0's output dropout mask (applied to the gradient arriving from layer 1)
    unsigned *flags;
    unsigned *abort_word;   // the call's own abort word (halo_lstm_status_offset of the backward workspace): zeroed by workgroup 0 at the start of the launch
   
    unsigned *status;                 // see PersistFwd
    unsigned long long *stamps;
    // optional GEMM operand images of the gate gradients (see PersistBwd): rows image of layer 0 only (layer 1's input gradient
    // is formed in this kernel), column images of both
    char *img_rows0, *img_cols0, *img_cols1;
    float *bias_part0, *bias_part1;   // [ceil(B/16)][4H] per layer, may be NULL
    int skip_dg0, skip_dg1;           // the layer's fp32 gate gradients are not stored back into its gates buffer: every consumer reads the emitted images / partials
    int poll_mode, replica_shift, nap;
    int T, B, H;
    // the launch works on batch tiles [bt0, bt0 + nbt) of the ceil(B/16) (a batch larger than the chip holds workgroups for runs as several
    // launches over the same buffers); its epoch words are those of the tiles' LOCAL index and count from epoch0, so a later launch never
    // mistakes an earlier one's epochs for its own (set by halo_lstm_persist2_fwd / _bwd)
    int bt0, nbt;
    unsigned epoch0;
};

bool halo_lstm_persist2_ok(int T, int B, int H, int L);   // shape, arithmetic mode (bf16), CU count, switch
void halo_lstm_persist2_enable(int on);
void halo_lstm_interleave_enable(int on);
int halo_lstm_persist2_fwd(const Persist2Fwd &a, hipStream_t st);
int halo_lstm_persist2_bwd(const Persist2Bwd &a, hipStream_t st);
// the same launches with TWO batch tiles per workgroup, interleaved (lstm_persist2x.hip): tiles [a.bt0, a.bt0 + a.nbt), a.epoch0 set by the caller
int halo_lstm_persist2x_fwd(const Persist2Fwd &a, hipStream_t st);
int halo_lstm_persist2x_bwd(const Persist2Bwd &a, hipStream_t st);
