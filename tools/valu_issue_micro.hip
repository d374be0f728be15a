// GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_issue tools/valu_issue_micro.hip && /tmp/valu_issue
//
// What does ONE wave64 vector instruction cost a SIMD to issue, by instruction type, with 1, 2 and 4 waves resident per SIMD?  (round-4 review
// item 5: the "VALU issue ceiling" the late stage's kernel is priced against was 1,024 SIMDs x 2.4 GHz / 4 cycles -- the figure of one wave
// ALONE on a SIMD, and of float64; k_strict2 runs 4 waves per SIMD and a third of its instructions are float32 / integer.)
//
// Each kernel is an unrolled stream of ONE instruction on 8 independent register sets (no dependent chain shorter than 8 instructions),
// written in inline assembly so that the compiler neither removes nor fuses anything; a wave brackets its stream with s_memtime (shader
// clock).  Reported per (instruction, waves per SIMD): cycles between two instructions of the SAME wave, cycles per instruction of the
// SIMD (= the former / waves per SIMD: the issue cost), and the chip's rate in wave instructions per second from the kernel's duration.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int UNROLL = 8;      // independent register sets
constexpr int INNER = 16;      // instructions per set and loop iteration

#define REP8(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)

enum Kind { FMA_F32, ADD_F32, MUL_F32, FMA_F64, ADD_F64, MUL_F64, RCP_F64, RCP_F32, CVT_F64_F32, CVT_F32_F64, CVT_I32_F64, ADD_U32, MUL_LO_U32,
            LSHL_B64, CNDMASK, RNDNE_F64, LDEXP_F64, PK_FMA_F32, N_KINDS };
static const char* kind_name[N_KINDS] = {"v_fma_f32", "v_add_f32", "v_mul_f32", "v_fma_f64", "v_add_f64", "v_mul_f64", "v_rcp_f64", "v_rcp_f32",
                                         "v_cvt_f64_f32", "v_cvt_f32_f64", "v_cvt_i32_f64", "v_add_u32", "v_mul_lo_u32", "v_lshlrev_b64",
                                         "v_cndmask_b32", "v_rndne_f64", "v_ldexp_f64", "v_pk_fma_f32"};

template <int KIND> __device__ __forceinline__ void one(float& a, double& d, int& i, long long& l, float b, double e)
{
    if (KIND == FMA_F32) asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(a) : "v"(b));
    else if (KIND == ADD_F32) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a) : "v"(b));
    else if (KIND == MUL_F32) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a) : "v"(b));
    else if (KIND == FMA_F64) asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(d) : "v"(e));
    else if (KIND == ADD_F64) asm volatile("v_add_f64 %0, %1, %0" : "+v"(d) : "v"(e));
    else if (KIND == MUL_F64) asm volatile("v_mul_f64 %0, %1, %0" : "+v"(d) : "v"(e));
    else if (KIND == RCP_F64) asm volatile("v_rcp_f64 %0, %0" : "+v"(d));
    else if (KIND == RCP_F32) asm volatile("v_rcp_f32 %0, %0" : "+v"(a));
    else if (KIND == CVT_F64_F32) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d) : "v"(a));
    else if (KIND == CVT_F32_F64) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a) : "v"(d));
    else if (KIND == CVT_I32_F64) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(i) : "v"(d));
    else if (KIND == ADD_U32) asm volatile("v_add_u32 %0, %1, %0" : "+v"(i) : "v"(i));
    else if (KIND == MUL_LO_U32) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(i) : "v"(i));
    else if (KIND == LSHL_B64) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(l));
    else if (KIND == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(i) : "v"(i) : );
    else if (KIND == RNDNE_F64) asm volatile("v_rndne_f64 %0, %0" : "+v"(d));
    else if (KIND == LDEXP_F64) asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(d));
    else if (KIND == PK_FMA_F32) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(d) : "v"(e));   // (two float32 in a 64-bit pair)
}

template <int KIND> __global__ __launch_bounds__(256) void k_issue(int iters, unsigned long long* cyc, float* sink)
{
    float a[UNROLL];
    double d[UNROLL];
    int ii[UNROLL];
    long long ll[UNROLL];
    const float b = 1.0000001f + 1e-9f * (float)threadIdx.x;
    const double e = 1.0000000001 + 1e-12 * (double)threadIdx.x;
#pragma unroll
    for (int u = 0; u < UNROLL; u++) { a[u] = 0.5f + u; d[u] = 0.5 + u; ii[u] = threadIdx.x + u; ll[u] = threadIdx.x + u; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < INNER; k++) {
#pragma unroll
            for (int u = 0; u < UNROLL; u++) one<KIND>(a[u], d[u], ii[u], ll[u], b, e);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.0f;
#pragma unroll
    for (int u = 0; u < UNROLL; u++) s += a[u] + (float)d[u] + (float)ii[u] + (float)ll[u];
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if ((threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND> static void run(int waves_per_simd, int n_cu, double clock_hz, unsigned long long* d_cyc, float* d_sink, double* out_cyc_wave, double* out_rate, double* out_mhz)
{
    // a block = 4 waves = one wave per SIMD of a CU; waves_per_simd blocks per CU
    const int blocks = n_cu * waves_per_simd, iters = 400;
    k_issue<KIND><<<blocks, 256>>>(20, d_cyc, d_sink);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    k_issue<KIND><<<blocks, 256>>>(iters, d_cyc, d_sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.0f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> cyc((size_t)blocks * 4);
    CK(hipMemcpy(cyc.data(), d_cyc, cyc.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::sort(cyc.begin(), cyc.end());
    const double n_instr = (double)iters * INNER * UNROLL;
    // (s_memtime ticks are shader cycles on gfx950: MI355X_MICROARCH.md, "s_memtime tick vs SQ PMC units"; the median wave's)
    (void)clock_hz;
    *out_cyc_wave = (double)cyc[cyc.size() / 2] / n_instr;
    *out_mhz = (double)cyc[cyc.size() / 2] / ((double)ms * 1e-3) / 1e6;   // (ticks per second of the kernel's duration: the clock the stream ran at)
    *out_rate = n_instr * (double)blocks * 4.0 / ((double)ms * 1e-3);
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    const double clock_hz = (double)prop.clockRate * 1e3;
    printf("# %s, %d CUs, clockRate %.0f MHz; s_memtime ticks = shader cycles\n", prop.name, n_cu, clock_hz / 1e6);
    unsigned long long* d_cyc; float* d_sink;
    CK(hipMalloc(&d_cyc, sizeof(unsigned long long) * (size_t)n_cu * 8 * 4));
    CK(hipMalloc(&d_sink, sizeof(float) * (size_t)n_cu * 8 * 256));
    printf("%-16s %6s %24s %20s %26s %10s\n", "instruction", "w/SIMD", "cycles/instr (one wave)", "cycles/instr (SIMD)", "chip wave-instr/s (1e12)", "MHz");
#define ROW(K) for (int w : {1, 2, 4, 8}) { double c, r, f; run<K>(w, n_cu, clock_hz, d_cyc, d_sink, &c, &r, &f); \
        printf("%-16s %6d %24.2f %20.2f %26.4f %10.0f\n", kind_name[K], w, c, c / w, r / 1e12, f); }
    ROW(FMA_F32) ROW(ADD_F32) ROW(MUL_F32) ROW(PK_FMA_F32) ROW(FMA_F64) ROW(ADD_F64) ROW(MUL_F64) ROW(RCP_F64) ROW(RCP_F32) ROW(CVT_F64_F32) ROW(CVT_F32_F64)
    ROW(CVT_I32_F64) ROW(RNDNE_F64) ROW(LDEXP_F64) ROW(ADD_U32) ROW(MUL_LO_U32) ROW(LSHL_B64) ROW(CNDMASK)
    return 0;
}
