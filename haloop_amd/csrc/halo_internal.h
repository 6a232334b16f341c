// Host-side helpers shared between translation units of libhalo (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include "halo_common.h"
#include "small_jobs.h"

int halo_transpose(const float *in, float *out, int rows, int cols, hipStream_t st);
int halo_fill(float *p, size_t n, float v, hipStream_t st);
int halo_colsum2(const float *x, int rows, int cols, int ld, float *out, float *out2, hipStream_t st);

// ---- split-bf16 GEMM on pre-tiled operand images (gemm_bf16x3.hip) ----
size_t halo_tiled_image_bytes(int R, int K);
// image <- split/tiled copy of logical X[R][K]; src_transposed: memory is [K][R] (leading dim ld)
int halo_prep_tiles(const float *src, int R, int K, int ld, int src_transposed, void *image, hipStream_t st);
int halo_prep_pair(const float *src, int R, int C, int ld, void *image_rm, void *image_tr, hipStream_t st);   // both images, one read
// several images in one launch.  kind 0: image <- row-major src [R][K]; 1: image <- src stored [K][R] (logical X[r][k] = src[k*ld + r]);
// 2: image (rows R, k = K) and image_tr (rows K, k = R) from one read of a row-major src [R][K] (halo_prep_pair);
// 3: not an image: fp32 column sums over the R rows of src [R][K], written to (float *)image and, when given, (float *)image_tr
struct HaloPrepJob {
    int kind;
    const float *src;
    int R, K, ld;
    void *image, *image_tr;
};
int halo_prep_jobs(const HaloPrepJob *jobs, int n, hipStream_t st);
// C[M,N] = A[M,K] * B[N,K]^T from images, epilogue as halo_gemm_f32
int halo_gemm_bf16x3_tiled(const void *Aimg, const void *Bimg, int M, int N, int K, float *C, int ldc,
                           const float *bias1, const float *bias2, int relu, const DropoutCfg *drop, hipStream_t st);
int halo_gemm_bf16x3_tiled_slices(const void *Aimg, const void *Bimg, int M, int N, int K, float *slab, int want, int *slices, hipStream_t st);
// ---- settings record (include/halo.h, "Contexts"): every switch the halo_set_* entries change lives here.  A thread that has selected
// a caller-owned context with halo_ctx_use() reads and writes THAT record; every other thread the process-wide default one.
// queue a small reduction on the context (true) or tell the caller to launch it itself (false: deferral off or the queue is full)
bool halo_defer_small_job(const HaloSmallJob &job);
struct HaloCtx {
    int math_mode = 0;
    int lstm_fusion = 0;
    int lstm_persistent = 1, lstm_persistent2 = 1;
    int lstm_interleave = 1;             // two batch tiles per workgroup in the two-layer launches when the batch has more tiles than one launch holds (lstm_persist2x.hip)
    int persist_emit = -1;               // -1: HALO_PERSIST_EMIT from the environment (default on)
    int beam_vec_chunk = 32;
    void *scratch = nullptr;
    size_t scratch_bytes = 0;
    int scratch_slot = 0;
    unsigned *status = nullptr;          // device word, sticky: set to non-zero by a persistent recurrence whose bounded wait timed out
    hipEvent_t chain_ev0 = nullptr, chain_ev1 = nullptr;
    int bwd_mid_recorded = 0;            // ... times since it was set
    hipEvent_t bwd_mid_event = nullptr;  // halo_set_lstm_bwd_mid_event: recorded when the top layer's weight gradients have been launched
    unsigned long long *stamps = nullptr;
    // halo_set_grad_sumsq: the squared-norm partials of a training step's clipped gradients, written by the launches that produce those
    // gradients (host bookkeeping: destination, capacity, how many slots are taken, which producers have contributed)
    float *grad_sumsq = nullptr;
    int grad_sumsq_cap = 0, grad_sumsq_n = 0;
    unsigned grad_sumsq_cover = 0;       // bit 0 / 1: the two-layer launch's upper / lower weight gradients; 2 / 3: their bias gradients; 4: the conv front end
    int defer_small_jobs = 0;            // halo_set_defer_small_jobs: small reductions wait in `small_jobs` for a launch that carries them
    HaloSmallJobs small_jobs = {};
    int lstm_dx_slabs = 1;               // halo_set_lstm_dx_slabs: K-slices the caller's dx buffer has room for
    int lstm_dx_slabs_left = 1;          // how many the last halo_lstm_bwd left unreduced there (1: dx itself)
    int lstm_expect_backward = 1;        // the two-layer forward also packs the backward's transposed weight images (halo_set_lstm_expect_backward)
    const float *packT_reserve = nullptr, *packT_w[3] = {nullptr, nullptr, nullptr};   // ... into this reserve, from these weights (host bookkeeping)
    const float *bwdflags_reserve = nullptr;   // ... and zeroed the backward launch's epoch words in this reserve (the backward then needs no prologue launch)
    const float *fwdT_src[2] = {nullptr, nullptr};
    const float *fwdT_reserve = nullptr;    // ... and in^T / W_ih^T of the pair's lower layer (the backward's operand launch then has nothing left to write)
    const float *emitT_reserve = nullptr;   // the two-layer forward wrote the weight-gradient products' h_prev^T / dropout(h0)^T operand images into this reserve
    // halo_set_lstm_weights_stamp: a non-zero stamp is the caller's promise that the LSTM weights only change when the stamp does; the
    // forward-only two-layer launch then keeps the packed weight images a previous call with the same reserve, weights, shape and stamp left
    uint64_t lstm_weights_stamp = 0, packF_stamp = 0;
    const float *packF_reserve = nullptr, *packF_w[3] = {nullptr, nullptr, nullptr};
    int packF_dims[5] = {0, 0, 0, 0, 0};
    int mute_block = -1;                 // test hook (halo_debug_mute_workgroup): this workgroup of a persistent forward never publishes
};
HaloCtx &halo_ctx_cur();
// CUs of the current device (hipDeviceProp_t::multiProcessorCount, cached per device; <= 0: the query failed)
int halo_cu_count();
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per DEVICE: one flag per (kernel slot, device) instead of a process-wide bool.
// Slots: 0 ctc_head_train_kernel, 1 / 2 gemm256_kernel (product / lab variants), 8 .. 31 the gemm_rows_kernel instantiations
bool halo_func_attr_done(int slot);
void halo_func_attr_set(int slot);
// C | C2 = A x (B stacked on B2)^T in one launch (columns [n_split, N) of the result go to C2); plain sums
int halo_gemm_bf16x3_tiled_nsplit(const void *Aimg, const void *Bimg, int M, int N, int K, float *C, int ldc, int n_split, float *C2, int ldc2,
                                  hipStream_t st);
int halo_gemm_bf16x3_tiled_nsplit_carry(const void *Aimg, const void *Bimg, int M, int N, int K, float *C, int ldc, int n_split, float *C2, int ldc2,
                                        const void *rA, const void *rB, int rM, int rN, int rK, float *rslab, int want, int *slices,
                                        float *sumsq_part, int *sumsq_parts, hipStream_t st);
// up to three plain-sum products C [| C2] = A x B^T over tiled operand images in ONE launch of 256 x 256 tiles (gemm256.hip; single-pass bf16
// only).  kslices > 1: K-slice s goes to C + s * slab_stride.  On return tiles = the workgroups the problem used (= sumsq partials written,
// when sumsq != NULL and kslices == 1) and kslices = the slices actually cut.
struct HaloG256Problem {
    const void *A, *B;
    int M, N, K;
    float *C; int ldc;
    int n_split; float *C2; int ldc2;      // n_split == N: one output
    int kslices; long slab_stride;
    float *sumsq;
    int tiles;
};
int halo_gemm256_enabled();
int halo_gemm256_fits(const HaloG256Problem &q);
int halo_gemm256_launch(HaloG256Problem *probs, int n, hipStream_t st);
int halo_math_mode();
int halo_lstm_fusion();   // 1: run multi-layer LSTMs as layer-diagonal fused launches (halo_set_lstm_fusion)   // 0 = exact f32 MFMA, 1 = split-bf16 (3-pass) for the large LSTM GEMMs

// ---- caller-provided scratch (halo_set_scratch) and split-K helpers ----
void halo_get_scratch(void **ptr, size_t *bytes);
void halo_set_scratch_slot(int slot);
int halo_side_stream(hipStream_t *side, hipEvent_t *fork_ev, hipEvent_t *join_ev);
int halo_pick_ksplit(long tiles, int k_steps, long out_elems);
int halo_splitk_reduce(const float *slab, int ksplit, int M, int N, float *C, int ldc, const float *bias1,
                       const float *bias2, int relu, const DropoutCfg &drop, int use_drop, hipStream_t st);
