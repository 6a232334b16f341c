// Host-side helpers shared between translation units of libhalo (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

int halo_transpose(const float *in, float *out, int rows, int cols, hipStream_t st);
int halo_fill(float *p, size_t n, float v, hipStream_t st);
