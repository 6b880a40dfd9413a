// north_star names a "wavefront-wide prefix scan for compositing".  This probe measures what it would buy on this chip, on
// the compositor's own arithmetic (forward.cu:346-386), for one 16x16 tile against lists of n entries:
//   seq  : lanes = pixels (what the kernels do): every lane walks the list front to back, T = T (1 - alpha) sequentially,
//          stops when T (1 - alpha) < 1e-4 (the whole wavefront once every lane has);
//   scan : lanes = entries: for one pixel at a time the 64 entries of a chunk evaluate their alpha side by side, the
//          transmittance in front of each is an exclusive prefix product over the wavefront (DPP row shifts + row broadcasts),
//          the colour a wave reduction per channel (C of them; C = 0: transmittance and last contributor only).
// Both produce {final T, last contributor, C colours} per pixel; the scan re-associates the product (results agree to ~1e-6,
// not bit for bit).  Opaque lists end early for `seq` (T falls below 1e-4 after a few entries); the scan cannot know before
// it has evaluated the chunk.  Build + run (GPU box):
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/scan_probe.hip -o tools/scan_probe && tools/scan_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

struct Entry {
    float x, y, a, b, c, w;   // centre, conic, opacity
};

__device__ __forceinline__ float alpha_of(const Entry& e, float px, float py, bool& ok)
{
    const float dx = e.x - px, dy = e.y - py;
    const float power = -0.5f * (e.a * dx * dx + e.c * dy * dy) - e.b * dx * dy;
    const float al = fminf(0.99f, e.w * __expf(power));
    ok = !(power > 0.0f) && !(al < 1.0f / 255.0f);
    return al;
}

template <int C>
__global__ __launch_bounds__(256) void k_seq(int n, const Entry* __restrict__ list, const float* __restrict__ feat, float* __restrict__ out)
{
    // block = tile replica, wave = 4-row strip, lane = pixel
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const float px = (float)(lane & 15), py = (float)(4 * wv + (lane >> 4));
    float T = 1.0f, acc[C > 0 ? C : 1] = {};
    int last = 0;
    bool done = false;
    for (int k = 0; k < n; k++) {
        if (__all(done)) break;
        const Entry e = list[k];   // wave-uniform: scalar loads
        bool ok;
        const float al = alpha_of(e, px, py, ok);
        const float tT = T * (1 - al);
        const bool pass = !done && ok, stop = pass && tT < 0.0001f, acc_ = pass && !stop;
#pragma unroll
        for (int ch = 0; ch < C; ch++) acc[ch] += acc_ ? feat[k * (C > 0 ? C : 1) + ch] * al * T : 0.0f;
        if (acc_) { T = tT; last = k + 1; }
        done = done || stop;
    }
    float* o = out + ((size_t)blockIdx.x * 256 + threadIdx.x) * (2 + C);
    o[0] = T; o[1] = (float)last;
#pragma unroll
    for (int ch = 0; ch < C; ch++) o[2 + ch] = acc[ch];
}

template <int CTRL, int ROWMASK = 0xf>
__device__ __forceinline__ float dpp_f(float old, float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, ROWMASK, 0xf, false));
}
// inclusive prefix product over the 64 lanes
__device__ __forceinline__ float wave_scan_mul(float v)
{
    v *= dpp_f<0x111>(1.0f, v);         // row_shr:1
    v *= dpp_f<0x112>(1.0f, v);         // row_shr:2
    v *= dpp_f<0x114>(1.0f, v);         // row_shr:4
    v *= dpp_f<0x118>(1.0f, v);         // row_shr:8
    v *= dpp_f<0x142, 0xa>(1.0f, v);    // row_bcast:15 into rows 1 and 3
    v *= dpp_f<0x143, 0xc>(1.0f, v);    // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ float wave_sum_all(float v)
{
    for (int off = 32; off; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <int C>
__global__ __launch_bounds__(256) void k_scan(int n, const Entry* __restrict__ list, const float* __restrict__ feat, float* __restrict__ out)
{
    // block = tile replica, wave = 4-row strip; lane = ENTRY of the current chunk; the strip's 64 pixels one after the other
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int p = 0; p < 64; p++) {
        const float px = (float)(p & 15), py = (float)(4 * wv + (p >> 4));
        float T = 1.0f, acc[C > 0 ? C : 1] = {};
        int last = 0;
        for (int c0 = 0; c0 < n; c0 += 64) {
            const int k = c0 + lane;
            bool ok = false;
            float al = 0.0f;
            if (k < n) al = alpha_of(list[k], px, py, ok);
            const float f = ok ? 1.0f - al : 1.0f;
            const float incl = wave_scan_mul(f);
            const float Tk = T * (incl / f);                       // transmittance in front of entry k
            const bool stop = ok && Tk * (1 - al) < 0.0001f;
            const unsigned long long sm = __ballot(stop);
            const int first_stop = sm ? __builtin_ctzll(sm) : 64;
            const bool acc_ = ok && lane < first_stop;
            const float w = acc_ ? al * Tk : 0.0f;
#pragma unroll
            for (int ch = 0; ch < C; ch++) acc[ch] += wave_sum_all(k < n ? feat[k * (C > 0 ? C : 1) + ch] * w : 0.0f);
            const unsigned long long am = __ballot(acc_);
            if (am) last = c0 + 64 - __builtin_clzll(am);
            // T behind the last accepted entry of the chunk
            const int src = first_stop < 64 ? (first_stop > 0 ? first_stop - 1 : -1) : 63;
            if (src >= 0) T = T * __shfl(incl, src, 64);
            if (sm) break;
        }
        if (lane == 0) {
            float* o = out + ((size_t)blockIdx.x * 256 + wv * 64 + p) * (2 + C);
            o[0] = T; o[1] = (float)last;
#pragma unroll
            for (int ch = 0; ch < C; ch++) o[2 + ch] = acc[ch];
        }
    }
}

template <int C>
static void run(int n, float opacity, const char* what)
{
    std::vector<Entry> h(n);
    std::vector<float> hf((size_t)n * (C > 0 ? C : 1));
    srand(1);
    for (int k = 0; k < n; k++) {
        h[k] = { 8.0f + 6.0f * (rand() / (float)RAND_MAX - 0.5f), 8.0f + 6.0f * (rand() / (float)RAND_MAX - 0.5f), 0.02f, 0.0f, 0.02f, opacity };
        for (int ch = 0; ch < C; ch++) hf[(size_t)k * C + ch] = (k % (C > 0 ? C : 1)) == ch ? 1.0f : 0.0f;
    }
    Entry* d; float *df, *o1, *o2;
    const int tiles = 8192;
    hipMalloc(&d, n * sizeof(Entry)); hipMalloc(&df, hf.size() * 4);
    hipMalloc(&o1, (size_t)tiles * 256 * (2 + C) * 4); hipMalloc(&o2, (size_t)tiles * 256 * (2 + C) * 4);
    hipMemcpy(d, h.data(), n * sizeof(Entry), hipMemcpyHostToDevice); hipMemcpy(df, hf.data(), hf.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float ms1 = 0, ms2 = 0;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(a); hipLaunchKernelGGL(k_seq<C>, dim3(tiles), dim3(256), 0, 0, n, d, df, o1); hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms1, a, b);
        hipEventRecord(a); hipLaunchKernelGGL(k_scan<C>, dim3(tiles), dim3(256), 0, 0, n, d, df, o2); hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms2, a, b);
    }
    std::vector<float> r1((size_t)256 * (2 + C)), r2(r1.size());
    hipMemcpy(r1.data(), o1, r1.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(r2.data(), o2, r2.size() * 4, hipMemcpyDeviceToHost);
    double md = 0; int lastdiff = 0;
    for (size_t i = 0; i < r1.size(); i++) {
        if (i % (2 + C) == 1) lastdiff += r1[i] != r2[i];
        else md = fmax(md, fabs((double)r1[i] - r2[i]));
    }
    printf("%-34s n = %4d C = %2d: lanes = pixels %8.1f us   lanes = entries (scan) %8.1f us   ratio %5.2f   max |diff| %.1e, last-contributor mismatches %d / 256\n",
           what, n, C, 1e3 * ms1, 1e3 * ms2, ms2 / ms1, md, lastdiff);
    hipFree(d); hipFree(df); hipFree(o1); hipFree(o2);
}

int main()
{
    // faint entries: nothing ends early (the best case for the scan: every pair is evaluated either way)
    run<0>(64, 0.02f, "faint, T only");
    run<0>(512, 0.004f, "faint, T only");
    run<1>(512, 0.004f, "faint, one channel");
    run<17>(512, 0.004f, "faint, 17 channels");
    run<17>(5, 0.02f, "stress-scene length, 17 channels");
    // opaque entries (the skeleton scenes: opacity 1): the sequential walk ends after a few entries
    run<0>(512, 1.0f, "opaque, T only");
    run<17>(512, 1.0f, "opaque, 17 channels");
    return 0;
}
