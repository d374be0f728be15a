// GPU box: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/mmc tools/model_math_check.hip && /tmp/mmc
// graal_amd/csrc/model_math.h against double-precision references computed by the device library, over EVERY float32 of the
// model's ranges:
//   mm_powf(x, y)  vs  (float)pow((double)x, (double)y)   x in [2^-15, 2^16), eight exponents      -> mismatches (ties aside: 0)
//   mm_expf(t)     vs  (float)exp((double)t)              t in (-88, -2^-20] and [2^-20, 88)        -> mismatches
//   mm_ln(x)       vs  log((double)x)                     every positive finite float32             -> largest difference in ulps of double
// (the double-precision pow / exp / log of the device library are good to < 1 ulp of double: rounded to float32 they are the
// correctly rounded result except within ~1e-8 ulp of a tie)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include "../graal_amd/csrc/model_math.h"

__global__ void k_pow(unsigned lo_bits, unsigned n, float y, unsigned long long* bad, unsigned* first)
{
    unsigned long long local = 0;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float x = __uint_as_float(lo_bits + i);
        const float a = (float)pow((double)x, (double)y), b = mm_powf(x, y);
        if (__float_as_uint(a) != __float_as_uint(b)) { local++; atomicMin(first, lo_bits + i); }
    }
    if (local) atomicAdd(bad, local);
}
__global__ void k_exp(unsigned lo_bits, unsigned n, unsigned long long* bad, unsigned* first)
{
    unsigned long long local = 0;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float t = __uint_as_float(lo_bits + i);
        const float a = (float)exp((double)t), b = mm_expf(t);
        if (__float_as_uint(a) != __float_as_uint(b)) { local++; atomicMin(first, lo_bits + i); }
    }
    if (local) atomicAdd(bad, local);
}
__global__ void k_ln(unsigned lo_bits, unsigned n, unsigned long long* worst)
{
    unsigned long long w = 0;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float x = __uint_as_float(lo_bits + i);
        const double a = log((double)x), b = mm_ln(x);
        long long d = __double_as_longlong(a) - __double_as_longlong(b);   // (same sign, or both tiny around ln 1 = 0: handled below)
        if ((a < 0) != (b < 0)) d = a == b ? 0 : (1ll << 40);
        if (d < 0) d = -d;
        if ((unsigned long long)d > w) w = (unsigned long long)d;
    }
    atomicMax(worst, w);
}
int main()
{
    unsigned long long* bad; unsigned* first;
    hipMalloc(&bad, 8); hipMalloc(&first, 4);
    auto reset = [&]() { unsigned long long z = 0; unsigned f = 0xffffffffu; hipMemcpy(bad, &z, 8, hipMemcpyHostToDevice); hipMemcpy(first, &f, 4, hipMemcpyHostToDevice); };
    auto get = [&](unsigned long long& z, unsigned& f) { hipDeviceSynchronize(); hipMemcpy(&z, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&f, first, 4, hipMemcpyDeviceToHost); };
    const float ys[] = {-1.5f, -1.0f, -0.5f, -2.25f, -1.3797f, 0.75f, -3.0f, -1.5000001f};
    const unsigned lo = 0x38000000u /* 2^-15 */, hi = 0x47800000u /* 2^16 */;
    for (float y : ys) {
        unsigned long long z; unsigned f;
        reset(); k_pow<<<4096, 256>>>(lo, hi - lo, y, bad, first); get(z, f);
        printf("pow y=%g: %u values, mismatches %llu (first bits 0x%08x)\n", y, hi - lo, z, f);
    }
    {
        unsigned long long z1, z2; unsigned f1, f2;
        const unsigned p_lo = 0x35800000u /* 2^-20 */, p_hi = 0x42b00000u /* 88 */;
        reset(); k_exp<<<4096, 256>>>(p_lo, p_hi - p_lo, bad, first); get(z1, f1);
        reset(); k_exp<<<4096, 256>>>(p_lo | 0x80000000u, p_hi - p_lo, bad, first); get(z2, f2);
        printf("exp: 2 x %u values, mismatches %llu (first bits 0x%08x 0x%08x)\n", p_hi - p_lo, z1 + z2, f1, f2);
    }
    {
        unsigned long long z; unsigned f;
        reset(); k_ln<<<4096, 256>>>(1u, 0x7f800000u - 1u, bad); get(z, f);
        printf("ln: every positive finite float32, largest difference %llu ulp of double\n", z);
    }
    return 0;
}
