#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef float v2f __attribute__((ext_vector_type(2)));
extern "C" __device__ v2f __ocmlpriv_epln_f32(float);
extern "C" __device__ float __ocmlpriv_expep_f32(v2f);
__device__ __forceinline__ float powf_pos(float x, float y)
{
    const v2f p = __ocmlpriv_epln_f32(x);
    const float yh = y * p.y;
    const float err = fmaf(y, p.y, -yh);
    const float t = fmaf(y, p.x, err);
    const float s = yh + t;
    const float e = t - (s - yh);
    v2f a; a.x = e; a.y = s;
    return __ocmlpriv_expep_f32(a);
}
__global__ void k_cmp(unsigned lo_bits, unsigned n, float y, unsigned long long* bad, unsigned* first)
{
    unsigned long long local = 0;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float x = __uint_as_float(lo_bits + i);
        const float a = powf(x, y), b = powf_pos(x, y);
        if (__float_as_uint(a) != __float_as_uint(b)) { local++; atomicMin(first, lo_bits + i); }
    }
    if (local) atomicAdd(bad, local);
}
int main()
{
    unsigned long long* bad; unsigned* first;
    hipMalloc(&bad, 8); hipMalloc(&first, 4);
    const float ys[] = {-1.5f, -1.0f, -0.5f, -2.25f, -1.3797f, 0.75f, -3.0f, -1.5000001f};
    const unsigned lo = 0x38000000u /* 2^-15 */, hi = 0x47800000u /* 2^16 */;
    for (float y : ys) {
        unsigned long long z = 0; unsigned f = 0xffffffffu;
        hipMemcpy(bad, &z, 8, hipMemcpyHostToDevice); hipMemcpy(first, &f, 4, hipMemcpyHostToDevice);
        k_cmp<<<4096, 256>>>(lo, hi - lo, y, bad, first);
        hipDeviceSynchronize();
        hipMemcpy(&z, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&f, first, 4, hipMemcpyDeviceToHost);
        printf("y=%g: %u values, mismatches %llu (first bits 0x%08x)\n", y, hi - lo, z, f);
    }
    return 0;
}
