// Token-tape batching for the LM loops (the data formats either side of the LSTM-LM / GPT paths): HBM-resident flat
// token tapes (uint8 bytes, int16/uint16 sentencepiece ids, int32 word ids) are cut into model batches on the device
// instead of by per-column Python slicing on the host.  Pure integer gathers, HBM-bound, bit-exact.
//   tape_batch_kernel   SymbolTapeNoPad.__getitem__            ha/symbol_tape.py:239-279
//   lm_batch_u16_kernel get_batch (objective "lm" / "cond")    ha/attention_loop.py:98-125
#include "halo_common.h"

namespace {

// out[t, b] = data[b*(tape_len-1) + i*bptt + t]  (pad where the index runs past the tape), t < rows
template <typename T>
__global__ __launch_bounds__(256) void tape_batch_kernel(const T *__restrict__ data, long n_tokens, int batch, long tape_len, long part,
                                                         int bptt, int rows, T pad, T *__restrict__ out) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)rows * batch) return;
    const int b = (int)(idx % batch);
    const long t = idx / batch;
    const long src = (long)b * (tape_len - 1) + part * bptt + t;
    out[idx] = (src >= 0 && src < n_tokens) ? data[src] : pad;
}

// x[b, t] = data[ix[b] + t];  lm: y[b, t] = x[b, t+1] (0 in the last column);  cond: y keeps only the column
// (number of non-zero x) - 2 of each row (the final token before the padding), zero elsewhere
__global__ __launch_bounds__(256) void lm_batch_u16_kernel(const uint16_t *__restrict__ data, long n_tokens, const int64_t *__restrict__ ix,
                                                           int B, int T, int cond, int64_t *__restrict__ x, int64_t *__restrict__ y) {
    __shared__ int nz[4];
    const int b = blockIdx.x;
    const long base = ix[b];
    int cnt = 0;
    for (int t = threadIdx.x; t < T; t += 256) {
        const long s = base + t;
        const int64_t v = (s >= 0 && s < n_tokens) ? (int64_t)data[s] : 0;
        x[(long)b * T + t] = v;
        cnt += v != 0;
    }
    if (cond) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
        if ((threadIdx.x & 63) == 0) nz[threadIdx.x >> 6] = cnt;
        __syncthreads();
        cnt = nz[0] + nz[1] + nz[2] + nz[3];
    }
    const int keep = cnt - 2;
    for (int t = threadIdx.x; t < T; t += 256) {
        const long s = base + t + 1;
        int64_t v = (t < T - 1 && s < n_tokens) ? (int64_t)data[s] : 0;
        if (cond && t != keep) v = 0;
        y[(long)b * T + t] = v;
    }
}

template <typename T>
int launch_tape(const void *data, long n, int batch, long tape_len, long part, int bptt, int rows, long pad, void *out, hipStream_t st) {
    const long total = (long)rows * batch;
    hipLaunchKernelGGL(tape_batch_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const T *)data, n, batch, tape_len,
                       part, bptt, rows, (T)pad, (T *)out);
    return halo_launch_status();
}

}  // namespace

extern "C" {

int halo_tape_batch(const void *data, int elem_bytes, long n_tokens, int batch_size, int bptt_len, long part_index, int rows,
                    long pad_value, void *out, halo_stream_t stream) {
    HALO_CHECK_ARG(data && out && n_tokens > 0 && batch_size > 0 && bptt_len > 0 && part_index >= 0 && rows > 0 && rows <= bptt_len);
    const long tape_len = (n_tokens + batch_size - 1) / batch_size;
    hipStream_t st = (hipStream_t)stream;
    switch (elem_bytes) {
        case 1: return launch_tape<uint8_t>(data, n_tokens, batch_size, tape_len, part_index, bptt_len, rows, pad_value, out, st);
        case 2: return launch_tape<uint16_t>(data, n_tokens, batch_size, tape_len, part_index, bptt_len, rows, pad_value, out, st);
        case 4: return launch_tape<uint32_t>(data, n_tokens, batch_size, tape_len, part_index, bptt_len, rows, pad_value, out, st);
        case 8: return launch_tape<uint64_t>(data, n_tokens, batch_size, tape_len, part_index, bptt_len, rows, pad_value, out, st);
        default: return HALO_ENOTSUP;
    }
}

int halo_lm_batch_u16(const uint16_t *data, long n_tokens, const int64_t *offsets, int B, int T, int objective_cond, int64_t *x,
                      int64_t *y, halo_stream_t stream) {
    HALO_CHECK_ARG(data && offsets && x && y && n_tokens > 0 && B > 0 && T > 0);
    hipLaunchKernelGGL(lm_batch_u16_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, data, n_tokens, offsets, B, T, objective_cond, x, y);
    return halo_launch_status();
}

}  // extern "C"
