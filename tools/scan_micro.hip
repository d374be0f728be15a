// Standalone exploration of k_scan's hit-free fast path (not part of the product): how fast can ONE isolated launch stream the
// row words of a 20 M-contact list (80 MB) -- and of a 120 M-contact one (480 MB, larger than the 256 MiB Infinity Cache) --
// through a 1-bit-per-id LDS bitmap test?  Variants of the loop structure, timed with a HIP event pair around every launch
// (isolated: the device idles between launches, like inside an MCMC step) and back to back.
// Build: hipcc --offload-arch=gfx950 -O3 -o scan_micro scan_micro.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int BM_WORDS = 1566;

__device__ __forceinline__ unsigned test4(const unsigned* bm, const v4i q)
{
    return ((bm[q.x >> 5] >> (q.x & 31)) & 1u) | (((bm[q.y >> 5] >> (q.y & 31)) & 1u) << 1) | (((bm[q.z >> 5] >> (q.z & 31)) & 1u) << 2)
           | (((bm[q.w >> 5] >> (q.w & 31)) & 1u) << 3);
}

// does any id in [lo, hi] (hi - lo < 32) have its bit set?  (one 64-bit LDS read at a wave-uniform address)
__device__ __forceinline__ bool range_hit(const unsigned* bm, int lo, int hi)
{
    const unsigned long long w = (unsigned long long)bm[lo >> 5] | ((unsigned long long)bm[(lo >> 5) + 1] << 32);
    const unsigned long long m = (w >> (lo & 31)) & ((2ull << (hi - lo)) - 1ull);
    return m != 0;
}

// VAR 0: round-1 structure (G loads, test, next G loads).  VAR 1: two buffers of G groups, software pipelined, 64-bit
// addresses.  VAR 2: same through raw buffer loads (one 32-bit offset register per batch).  VAR 3: VAR 2 + wave-range test
// (sorted rows: the wave's batch of a group covers ids [first lane's x, last lane's w]).
template <int VAR, int G>
__global__ __launch_bounds__(1024, (G <= 4 ? 8 : 4)) void k(const v4i* __restrict__ row4, int n4, const unsigned* __restrict__ bits, int n_bits,
                                                        unsigned long long* out)
{
    __shared__ unsigned s_bm[BM_WORDS + 2];
    const int t = threadIdx.x, lane = t & 63;
    const int stride = (int)(gridDim.x * blockDim.x);
    const int g0 = (int)(blockIdx.x * blockDim.x) + t;
    const int S = G * stride;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)row4, 0, (n4 + 1) * 16, 0x00020000);
    auto ldg = [&](int g) { return __builtin_nontemporal_load(row4 + (g < n4 ? g : n4)); };
    auto ldb = [&](int g, int i) { return __builtin_amdgcn_raw_buffer_load_b128(rs, g * 16, i * stride * 16, 2); }; // OOB -> 0
    v4i A[G], B[G];
    if (t >= 64) {
#pragma unroll
        for (int i = 0; i < G; i++) A[i] = (VAR >= 2) ? ldb(g0, i) : ldg(g0 + i * stride);
    }
    if (t < 64) {
        for (int i = t; i < BM_WORDS + 2; i += 64) s_bm[i] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
        if (t < n_bits) atomicOr(&s_bm[bits[t] >> 5], 1u << (bits[t] & 31));
#pragma unroll
        for (int i = 0; i < G; i++) A[i] = (VAR >= 2) ? ldb(g0, i) : ldg(g0 + i * stride);
    }
    __syncthreads();
    unsigned long long hits = 0;
    auto test = [&](const v4i (&rr)[G], int ga) -> unsigned {
        unsigned hit = 0;
        if (VAR == 3) {
            bool any = false;
#pragma unroll
            for (int i = 0; i < G; i++) {
                const int lo = __builtin_amdgcn_readfirstlane(rr[i].x), hi = __builtin_amdgcn_readlane(rr[i].w, 63);
                any = any || (unsigned)(hi - lo) >= 32u || range_hit(s_bm, lo, hi);
            }
            if (!any) return 0;
        }
#pragma unroll
        for (int i = 0; i < G; i++) hit |= test4(s_bm, rr[i]) << (4 * i);
        if (ga + (G - 1) * stride >= n4) {
#pragma unroll
            for (int i = 0; i < G; i++) if (ga + i * stride >= n4) hit &= ~(0xfu << (4 * i));
        }
        return hit;
    };
    if (VAR == 0) {
        hits += __popc(test(A, g0));
        for (int g = g0 + S; g <= n4; g += S) {
            v4i q[G];
#pragma unroll
            for (int i = 0; i < G; i++) q[i] = ldg(g + i * stride);
            const unsigned h = test(q, g);
            if (__ballot(h != 0)) hits += __popc(h);
        }
    } else {
#pragma unroll
        for (int i = 0; i < G; i++) B[i] = (VAR >= 2) ? ldb(g0 + S, i) : ldg(g0 + S + i * stride);
        const int g0w = __builtin_amdgcn_readfirstlane(g0);
        const int nb = g0w <= n4 ? (n4 - g0w) / S + 1 : 0;
        int g = g0;
        for (int it = 0; it < nb; it += 2) {
            { const unsigned h = test(A, g); if (__ballot(h != 0)) hits += __popc(h); }
#pragma unroll
            for (int i = 0; i < G; i++) A[i] = (VAR >= 2) ? ldb(g + 2 * S, i) : ldg(g + 2 * S + i * stride);
            { const unsigned h = test(B, g + S); if (__ballot(h != 0)) hits += __popc(h); }
#pragma unroll
            for (int i = 0; i < G; i++) B[i] = (VAR >= 2) ? ldb(g + 3 * S, i) : ldg(g + 3 * S + i * stride);
            g += 2 * S;
        }
    }
    if (hits) atomicAdd(out, hits);
    (void)lane;
}

template <int VAR, int G>
void run(const char* name, int blocks, int threads, const v4i* row4, int n4, const unsigned* bits, int n_bits, unsigned long long* out, double bytes)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    CK(hipMemsetAsync(out, 0, 8, st));
    for (int i = 0; i < 3; i++) k<VAR, G><<<blocks, threads, 0, st>>>(row4, n4, bits, n_bits, out);
    CK(hipStreamSynchronize(st));
    unsigned long long h = 0; CK(hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost));
    const int reps = 20;
    CK(hipEventRecord(a, st));
    for (int i = 0; i < reps; i++) k<VAR, G><<<blocks, threads, 0, st>>>(row4, n4, bits, n_bits, out);
    CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); const float b2b = ms / reps;
    std::vector<float> iso;
    for (int i = 0; i < 30; i++) {
        usleep(60);
        CK(hipEventRecord(a, st));
        k<VAR, G><<<blocks, threads, 0, st>>>(row4, n4, bits, n_bits, out);
        CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
        CK(hipEventElapsedTime(&ms, a, b)); iso.push_back(ms);
    }
    std::sort(iso.begin(), iso.end());
    const float med = iso[iso.size() / 2];
    printf("%-28s %4d x %4d  back-to-back %6.2f us %5.0f GB/s | isolated median %6.2f us %5.0f GB/s (min %.2f)  hits/launch %llu\n", name, blocks, threads,
           b2b * 1e3, bytes / b2b / 1e6, med * 1e3, bytes / med / 1e6, iso[0] * 1e3, h / 3);
    CK(hipStreamDestroy(st));
}

int main(int argc, char** argv)
{
    const int n = 50000;
    for (long long nnz : {20000000ll, 120000000ll}) {
        std::vector<int> row(nnz + 8, 0);
        for (long long i = 0; i < nnz; i++) row[i] = (int)(i * n / nnz);
        std::vector<unsigned> bits;
        srand(1);
        for (int i = 0; i < 12; i++) bits.push_back(rand() % n);
        int* drow; unsigned* dbits; unsigned long long* dout;
        CK(hipMalloc(&drow, (nnz + 8) * 4)); CK(hipMalloc(&dbits, 64 * 4)); CK(hipMalloc(&dout, 64));
        CK(hipMemcpy(drow, row.data(), (nnz + 8) * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dbits, bits.data(), bits.size() * 4, hipMemcpyHostToDevice));
        const int n4 = (int)(nnz >> 2); const double bytes = 4.0 * nnz;
        const v4i* r4 = (const v4i*)drow; const int nb = (int)bits.size();
        printf("---- %lld contacts (%.0f MB of row words)\n", nnz, bytes / 1e6);
        run<0, 4>("r1 G4", 496, 1024, r4, n4, dbits, nb, dout, bytes);
        run<0, 4>("r1 G4", 512, 1024, r4, n4, dbits, nb, dout, bytes);
        run<1, 2>("pipe2 G2 global", 496, 1024, r4, n4, dbits, nb, dout, bytes);
        run<2, 2>("pipe2 G2 buffer", 496, 1024, r4, n4, dbits, nb, dout, bytes);
        run<2, 4>("pipe2 G4 buffer", 496, 1024, r4, n4, dbits, nb, dout, bytes);
        run<3, 4>("pipe2 G4 buffer + range", 496, 1024, r4, n4, dbits, nb, dout, bytes);
        run<3, 2>("pipe2 G2 buffer + range", 496, 1024, r4, n4, dbits, nb, dout, bytes);
        run<3, 4>("pipe2 G4 buffer + range", 512, 1024, r4, n4, dbits, nb, dout, bytes);
        run<3, 4>("pipe2 G4 buffer + range", 1024, 512, r4, n4, dbits, nb, dout, bytes);
        run<3, 4>("pipe2 G4 buffer + range", 2048, 256, r4, n4, dbits, nb, dout, bytes);
        run<3, 8>("pipe2 G8 buffer + range", 256, 1024, r4, n4, dbits, nb, dout, bytes);
        run<3, 8>("pipe2 G8 buffer + range", 512, 512, r4, n4, dbits, nb, dout, bytes);
        run<0, 8>("r1 G8", 256, 1024, r4, n4, dbits, nb, dout, bytes);
        CK(hipFree(drow)); CK(hipFree(dbits)); CK(hipFree(dout));
    }
    return 0;
}
