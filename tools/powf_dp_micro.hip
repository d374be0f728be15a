// GPU box: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/powf_dp tools/powf_dp_micro.hip && /tmp/powf_dp
// powf for x > 0 three ways -- the device library's powf, powf_pos (its internals without the special cases: what the engine called
// until round 3) and the engine's SELF-CONTAINED double-precision version (model_math.h, no ROCm-internal symbols) -- timed (ALU-bound loop) and compared, on a sample, with
// the HOST's powf (glibc: what the oracle calls) and with the correctly rounded value (powl rounded to float).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
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

// the engine's functions (self-contained, double precision, correctly rounded)
#include "../graal_amd/csrc/model_math.h"
__device__ __forceinline__ float powf_dp(float x, float y) { return mm_powf(x, y); }
__device__ __forceinline__ float expf_dp(float t) { return mm_expf(t); }

template <int WHICH> __device__ __forceinline__ float f(float x, float y)
{
    return WHICH == 0 ? powf(x, y) : (WHICH == 1 ? powf_pos(x, y) : (WHICH == 2 ? powf_dp(x, y) : (WHICH == 3 ? expf(-x) : expf_dp(-x))));
}
template <int WHICH> __global__ void k_eval(const float* x, float y, float* out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = f<WHICH>(x[i], y);
}
template <int WHICH> __global__ void k_time(float y, float* out, int iters)
{
    float x = 0.5f + 1e-6f * (float)(blockIdx.x * blockDim.x + threadIdx.x);
    float acc = 0.0f;
    for (int i = 0; i < iters; i++) { acc += f<WHICH>(x, y); x += 1e-3f; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int WHICH> static double time_it(float y, float* d_out)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 256 * 16, threads = 256, iters = 2000;
    k_time<WHICH><<<blocks, threads>>>(y, d_out, 10);
    hipEventRecord(a);
    k_time<WHICH><<<blocks, threads>>>(y, d_out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return (double)ms * 1e6 / ((double)blocks * threads * iters);   // ns per evaluation per ... whole chip: divide work
}
int main()
{
    const int n = 1 << 22;
    std::vector<float> x(n), o0(n), o1(n), o2(n);
    srand(5);
    for (int i = 0; i < n; i++) { const double u = (double)rand() / RAND_MAX; x[i] = (float)exp(log(1e-3) + u * (log(3e4) - log(1e-3))); }
    float *dx, *dout;
    hipMalloc(&dx, 4 * n); hipMalloc(&dout, 4 * n);
    hipMemcpy(dx, x.data(), 4 * n, hipMemcpyHostToDevice);
    const float ys[] = {-1.5f, -1.3797f, -0.5f, -2.25f};
    for (float y : ys) {
        k_eval<0><<<n / 256, 256>>>(dx, y, dout, n); hipMemcpy(o0.data(), dout, 4 * n, hipMemcpyDeviceToHost);
        k_eval<1><<<n / 256, 256>>>(dx, y, dout, n); hipMemcpy(o1.data(), dout, 4 * n, hipMemcpyDeviceToHost);
        k_eval<2><<<n / 256, 256>>>(dx, y, dout, n); hipMemcpy(o2.data(), dout, 4 * n, hipMemcpyDeviceToHost);
        long d01 = 0, h0 = 0, h2 = 0, c0 = 0, c2 = 0, hc = 0;
        for (int i = 0; i < n; i++) {
            const float host = powf(x[i], y);
            const float cr = (float)powl((long double)x[i], (long double)y);
            d01 += o0[i] != o1[i]; h0 += o0[i] != host; h2 += o2[i] != host; c0 += o0[i] != cr; c2 += o2[i] != cr; hc += host != cr;
        }
        printf("y=%g, %d values: device powf != powf_pos %ld | != host powf: device %ld, dp %ld | != correctly rounded: device %ld, dp %ld, host %ld\n",
               y, n, d01, h0, h2, c0, c2, hc);
    }
    {   // expf on the model's argument range (d - 2) / (s^2 + d) in (-1.5, 0]
        for (int i = 0; i < n; i++) x[i] = 1.5f * (float)rand() / (float)RAND_MAX;
        hipMemcpy(dx, x.data(), 4 * n, hipMemcpyHostToDevice);
        k_eval<3><<<n / 256, 256>>>(dx, 0.f, dout, n); hipMemcpy(o0.data(), dout, 4 * n, hipMemcpyDeviceToHost);
        k_eval<4><<<n / 256, 256>>>(dx, 0.f, dout, n); hipMemcpy(o2.data(), dout, 4 * n, hipMemcpyDeviceToHost);
        long h0 = 0, h2 = 0;
        for (int i = 0; i < n; i++) { const float host = expf(-x[i]); h0 += o0[i] != host; h2 += o2[i] != host; }
        printf("expf(-x), x in (0, 1.5): != host expf: device %ld, dp %ld of %d\n", h0, h2, n);
    }
    const double t0 = time_it<0>(-1.5f, dout), t1 = time_it<1>(-1.5f, dout), t2 = time_it<2>(-1.5f, dout), t3 = time_it<3>(0.f, dout), t4 = time_it<4>(0.f, dout);
    printf("whole-chip time per evaluation (ps): powf %.2f  powf_pos %.2f  powf_dp %.2f | expf %.2f  expf_dp %.2f\n", 1e3 * t0, 1e3 * t1, 1e3 * t2, 1e3 * t3, 1e3 * t4);
    return 0;
}
