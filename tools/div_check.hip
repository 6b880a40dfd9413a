// Checks, over random operands in the ranges the SSIM expressions produce, that a quotient computed from a shared
// refined reciprocal with two residual corrections equals the IEEE quotient n / d bit for bit.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt tools/div_check.hip -o tools/div_check
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__device__ __forceinline__ float refined_rcp(float d)
{
    float r = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float div_by(float n, float d, float r)
{
    float q = n * r;
    float e = __builtin_fmaf(-d, q, n);
    q = __builtin_fmaf(e, r, q);
    e = __builtin_fmaf(-d, q, n);
    return __builtin_fmaf(e, r, q);
}

__device__ __forceinline__ uint32_t rng(uint64_t& s)
{
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    return (uint32_t)(s >> 33) ^ (uint32_t)s;
}
// magnitude log-uniform in [2^lo, 2^hi), random mantissa, optional random sign
__device__ __forceinline__ float rnd(uint64_t& s, int lo, int hi, bool sign)
{
    const uint32_t m = rng(s) & 0x7fffff, e = 127 + lo + rng(s) % (uint32_t)(hi - lo);
    return __uint_as_float((sign && (rng(s) & 1) ? 0x80000000u : 0u) | (e << 23) | m);
}

__global__ void k_check(unsigned long long* bad, unsigned long long* worst, int iters)
{
    uint64_t s = 0x9E3779B97F4A7C15ull * (blockIdx.x * blockDim.x + threadIdx.x + 1);
    unsigned long long nb = 0;
    for (int i = 0; i < iters; i++) {
        const float d = rnd(s, -40, 8, false), n = rnd(s, -60, 12, true);
        const float want = n / d, got = div_by(n, d, refined_rcp(d));
        if (__float_as_uint(want) != __float_as_uint(got)) {
            nb++;
            atomicMax(worst, (unsigned long long)abs((int)(__float_as_uint(want) - __float_as_uint(got))));
        }
    }
    if (nb) atomicAdd(bad, nb);
}

int main()
{
    unsigned long long *bad, *worst, h[2] = { 0, 0 };
    if (hipMalloc(&bad, 8) != hipSuccess || hipMalloc(&worst, 8) != hipSuccess) return 2;
    (void)hipMemset(bad, 0, 8);
    (void)hipMemset(worst, 0, 8);
    const int blocks = 4096, threads = 256, iters = 4096;
    hipLaunchKernelGGL(k_check, dim3(blocks), dim3(threads), 0, 0, bad, worst, iters);
    if (hipMemcpy(&h[0], bad, 8, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    if (hipMemcpy(&h[1], worst, 8, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    printf("pairs %llu  mismatches %llu  worst ulp distance %llu\n", (unsigned long long)blocks * threads * iters, h[0], h[1]);
    return h[0] != 0;
}
