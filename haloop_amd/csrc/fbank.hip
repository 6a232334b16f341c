// 80-mel log filterbank features on the device (the reference computes them in its DataLoader workers on the CPU,
// ha/data.py:136-140: torchaudio.compliance.kaldi.fbank(wav, num_mel_bins=80)).  Three small kernels around two exact-f32 products:
//   halo_fbank_frames   waveform -> [m][padded] frames: snip-edges framing, DC removal, pre-emphasis, window, zero padding
//   (halo_gemm_f32)     frames x [cos | -sin] DFT matrix -> [m][2 * (padded/2 + 1)] spectrum (a 512-point real DFT as one product)
//   halo_fbank_power    |X|^2 per bin -> [m][ld] (columns past padded/2 + 1 zero)
//   (halo_gemm_f32)     power x mel-filter matrix -> [m][num_bins]
//   halo_fbank_log      log(max(x, eps)) in place
// The filter bank, window and DFT matrices are constants the caller builds once (haloop_amd/fbank.py).
#include "halo_common.h"

namespace {

// one workgroup per frame
__global__ __launch_bounds__(256) void fbank_frames_kernel(const float *__restrict__ wav, int frame_len, int shift, int padded,
                                                           float preemph, int remove_dc, const float *__restrict__ window,
                                                           float *__restrict__ frames) {
    __shared__ float red[4];
    const float *x = wav + (long)blockIdx.x * shift;
    float *out = frames + (long)blockIdx.x * padded;
    float mean = 0.f;
    if (remove_dc) {
        float s = 0.f;
        for (int i = threadIdx.x; i < frame_len; i += 256) s += x[i];
        s = wave_sum(s);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        mean = ((red[0] + red[1]) + (red[2] + red[3])) / (float)frame_len;
    }
    for (int i = threadIdx.x; i < padded; i += 256) {
        float v = 0.f;
        if (i < frame_len) {
            const float cur = x[i] - mean, prev = x[i > 0 ? i - 1 : 0] - mean;      // the first sample is pre-emphasised against itself
            v = (cur - preemph * prev) * window[i];
        }
        out[i] = v;
    }
}

__global__ __launch_bounds__(256) void fbank_power_kernel(const float *__restrict__ spec, int bins, float *__restrict__ power, int ld, long n) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const long row = idx / ld;
    const int k = (int)(idx % ld);
    float v = 0.f;
    if (k < bins) {
        const float re = spec[row * 2 * bins + k], im = spec[row * 2 * bins + bins + k];
        v = re * re + im * im;
    }
    power[idx] = v;
}

__global__ __launch_bounds__(256) void fbank_log_kernel(float *__restrict__ x, long n, float eps) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx < n) x[idx] = logf(fmaxf(x[idx], eps));
}

// The spectrum, the mel energies and their log for one frame per workgroup, in FLOAT64: X[k] = sum_n x[n] e^{-2 pi i k n / N} against a
// table of the N twiddles (the index k n mod N is exact), |X|^2, the triangular filters, log(max(., eps)).  The fp32 product this
// replaces put an error of ~eps_32 * |largest bin| on EVERY bin, i.e. 2e-3 on the logs of bins 6 nepers below the frame's peak; the
// arithmetic is tiny (N / 2 + 1 bins x N terms per frame), so the double-precision vector rate is not a cost.
__global__ __launch_bounds__(256) void fbank_spectrum_mel_kernel(const float *__restrict__ frames, int padded, const double *__restrict__ twiddle,
                                                                 const double *__restrict__ banks, int num_bins, float eps,
                                                                 float *__restrict__ out) {
    extern __shared__ double sh[];                 // [x: padded][cos: padded][sin: padded][power: padded / 2 + 1]
    double *x = sh, *c = sh + padded, *sn = sh + 2 * padded, *pw = sh + 3 * padded;
    const int bins = padded / 2 + 1;
    const float *f = frames + (long)blockIdx.x * padded;
    for (int i = threadIdx.x; i < padded; i += 256) {
        x[i] = (double)f[i];
        c[i] = twiddle[2 * i];
        sn[i] = twiddle[2 * i + 1];
    }
    __syncthreads();
    for (int k = threadIdx.x; k < bins; k += 256) {
        double re = 0.0, im = 0.0;
        int idx = 0;                               // k n mod N, advanced by k per term
        for (int n = 0; n < padded; ++n) {
            re += x[n] * c[idx];
            im -= x[n] * sn[idx];
            idx = (idx + k) & (padded - 1);
        }
        pw[k] = re * re + im * im;
    }
    __syncthreads();
    for (int b = threadIdx.x; b < num_bins; b += 256) {
        const double *w = banks + (long)b * bins;
        double e = 0.0;
        for (int k = 0; k < bins; ++k) e += pw[k] * w[k];
        out[(long)blockIdx.x * num_bins + b] = (float)log(fmax(e, (double)eps));
    }
}

}  // namespace

extern "C" {

int halo_fbank_spectrum_mel(const float *frames, int n_frames, int padded, const double *twiddle, const double *banks, int num_bins,
                            float eps, float *out, halo_stream_t stream) {
    HALO_CHECK_ARG(frames && twiddle && banks && out && n_frames > 0 && num_bins > 0);
    HALO_CHECK_ARG(padded >= 2 && (padded & (padded - 1)) == 0 && padded <= 2048);
    const size_t lds = (size_t)(3 * padded + padded / 2 + 1) * sizeof(double);
    hipLaunchKernelGGL(fbank_spectrum_mel_kernel, dim3(n_frames), dim3(256), lds, (hipStream_t)stream, frames, padded, twiddle, banks, num_bins,
                       eps, out);
    return halo_launch_status();
}


int halo_fbank_frames(const float *wav, long n_samples, int frame_len, int shift, int padded, float preemphasis, int remove_dc,
                      const float *window, float *frames, int n_frames, halo_stream_t stream) {
    HALO_CHECK_ARG(wav && window && frames && frame_len > 0 && shift > 0 && padded >= frame_len && n_frames > 0);
    HALO_CHECK_ARG((long)(n_frames - 1) * shift + frame_len <= n_samples);
    hipLaunchKernelGGL(fbank_frames_kernel, dim3(n_frames), dim3(256), 0, (hipStream_t)stream, wav, frame_len, shift, padded, preemphasis,
                       remove_dc, window, frames);
    return halo_launch_status();
}

int halo_fbank_power(const float *spectrum, int n_frames, int bins, float *power, int ld, halo_stream_t stream) {
    HALO_CHECK_ARG(spectrum && power && n_frames > 0 && bins > 0 && ld >= bins);
    const long n = (long)n_frames * ld;
    hipLaunchKernelGGL(fbank_power_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, spectrum, bins, power, ld, n);
    return halo_launch_status();
}

int halo_fbank_log(float *x, long n, float eps, halo_stream_t stream) {
    HALO_CHECK_ARG(x && n > 0);
    hipLaunchKernelGGL(fbank_log_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, n, eps);
    return halo_launch_status();
}

}  // extern "C"
