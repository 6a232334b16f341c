// The 256 x 256-tile single-pass bf16 product (gemm256.h) behind the library's internal interface: up to three plain-sum products over
// tiled operand images in ONE launch.  Used by lstm.hip for the two layers' weight gradients + the carried K-slices of the input gradient.
#include <stdlib.h>
#include "halo_common.h"
#include "halo_internal.h"
#include "gemm256.h"

static int g256_switch = -1;

int halo_gemm256_enabled() {
    if (g256_switch < 0) {
        const char *e = getenv("HALO_GEMM256");
        g256_switch = (e && e[0] == '0') ? 0 : 1;
    }
    return g256_switch && halo_math_mode() == HALO_MATH_BF16;
}

int halo_gemm256_fits(const HaloG256Problem &q) {
    if (!q.A || !q.B || !q.C || q.M <= 0 || q.N <= 0 || q.K <= 0) return 0;
    if (q.n_split != q.N && (q.n_split <= 0 || q.n_split % 256 != 0 || q.n_split > q.N || !q.C2)) return 0;
    // the epilogue addresses an output through a 1 GiB buffer descriptor
    if ((long)q.M * q.ldc * 4 >= (1l << 30) || (q.n_split != q.N && (long)q.M * q.ldc2 * 4 >= (1l << 30))) return 0;
    return 1;
}

extern "C" int halo_set_gemm256(int on) { g256_switch = on ? 1 : 0; return HALO_OK; }

int halo_gemm256_launch(HaloG256Problem *probs, int n, hipStream_t st) {
    if (n < 1 || n > halo_g256::MAXP) return HALO_EINVAL;
    halo_g256::Args a = {};
    a.nprob = n;
    int first = 0;
    for (int i = 0; i < n; ++i) {
        if (!halo_gemm256_fits(probs[i])) return HALO_EINVAL;
        halo_g256::Prob &g = a.p[i];
        g.A = (const char *)probs[i].A; g.B = (const char *)probs[i].B;
        g.M = probs[i].M; g.N = probs[i].N; g.KT = (probs[i].K + 31) / 32;
        g.kslices = probs[i].kslices; g.slab_stride = probs[i].slab_stride;
        g.C = probs[i].C; g.C2 = probs[i].C2; g.ldc = probs[i].ldc; g.ldc2 = probs[i].ldc2; g.n_split = probs[i].n_split;
        g.sumsq = probs[i].sumsq;
        halo_g256::finish(g, first);
        probs[i].tiles = g.tiles_m * g.tiles_n * g.kslices;
        probs[i].kslices = g.kslices;
        first += probs[i].tiles;
    }
    if (halo_g256::launch<0>(a, first, st) != hipSuccess) return HALO_ELAUNCH;
    return halo_launch_status();
}
