// Micro-benchmark: how fast can gfx950 zero-fill (C+1) planar fp32 images, by access pattern?  (tools only)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
#ifdef USE_NT
#define ST(p, v) __builtin_nontemporal_store(v, &(p))
#else
#define ST(p, v) (p) = (v)
#endif
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// V1: classic grid-stride
__global__ __launch_bounds__(256) void fill_gridstride(v4f* p, size_t n4)
{
    v4f z = { 0, 0, 0, 0 };
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) ST(p[i], z);
}
// V2: each WG owns a contiguous chunk of `chunk4` float4
__global__ __launch_bounds__(256) void fill_chunk(v4f* p, size_t n4, int chunk4)
{
    v4f z = { 0, 0, 0, 0 };
    size_t b = (size_t)blockIdx.x * chunk4;
    size_t e = b + chunk4 < n4 ? b + chunk4 : n4;
    for (size_t i = b + threadIdx.x; i < e; i += 256) ST(p[i], z);
}
// V2b: like fill_chunk, plus what the rasterizer's fill role carries: a dependent uniform load deciding the path,
// 4 KB of static LDS, `nskip` early-exit blocks interleaved every 8th id
__global__ __launch_bounds__(256) void fill_chunk_like(v4f* p, size_t n4, int chunk4, const unsigned* cover, int nskip)
{
    __shared__ unsigned s_cw[1024];
    const int bid = blockIdx.x;
    const bool slot = (bid % 8 == 0) && (bid / 8 < nskip);
    if (slot) return;
    const int f = bid - min(bid / 8 + 1, nskip);
    const unsigned any = cover[f & 255];
    if (any) {
        if (threadIdx.x < 64) s_cw[threadIdx.x] = cover[threadIdx.x];
        __syncthreads();
    }
    v4f z = { 0, 0, 0, 0 };
    size_t b = (size_t)f * chunk4;
    size_t e = b + chunk4 < n4 ? b + chunk4 : n4;
    for (size_t i = b + threadIdx.x; i < e; i += 256) {
        if (any && ((s_cw[(i >> 2) & 63] >> (i & 31)) & 1u)) continue;
        ST(p[i], z);
    }
}
// V2c: 4 KB chunks with per-block resources like the fused rasterizer kernel (dynamic LDS, many VGPRs)
__global__ __launch_bounds__(256) void fill_chunk_dynlds(v4f* p, size_t n4, int chunk4, int heavy)
{
    extern __shared__ char dyn[];
    float acc[40];
    if (heavy) {  // never taken at run time (heavy == 0), but forces the register allocation
#pragma unroll
        for (int i = 0; i < 40; i++) acc[i] = p[i][0] * (float)i;
        float t = 0;
#pragma unroll
        for (int i = 0; i < 40; i++) t += acc[i] * acc[(i * 7) % 40];
        dyn[threadIdx.x] = (char)t;
        __syncthreads();
        if (dyn[(threadIdx.x + 1) & 255] == 3) return;
    }
    v4f z = { 0, 0, 0, 0 };
    size_t b = (size_t)blockIdx.x * chunk4;
    size_t e = b + chunk4 < n4 ? b + chunk4 : n4;
    for (size_t i = b + threadIdx.x; i < e; i += 256) ST(p[i], z);
}
// V3: WG (chunk, band) writes `passes` x 4 KB in each of `planes` planes (plane stride = plane4 float4)
__global__ __launch_bounds__(256) void fill_planes(v4f* p, size_t plane4, int planes, int band4, int passes)
{
    v4f z = { 0, 0, 0, 0 };
    size_t base = (size_t)blockIdx.y * band4 + (size_t)blockIdx.x * passes * 256;
    for (int ps = 0; ps < passes; ps++) {
        size_t o = base + ps * 256 + threadIdx.x;
        if (blockIdx.x * passes * 256 + ps * 256 + threadIdx.x >= (unsigned)band4) continue;
        for (int c = 0; c < planes; c++) ST(p[(size_t)c * plane4 + o], z);
    }
}
// V4: persistent: G WGs, each loops over 4 KB tiles t = blockIdx.x, +G, ... (tile = 256 float4 contiguous)
__global__ __launch_bounds__(256) void fill_tiles(v4f* p, size_t n4, int unroll_dummy)
{
    v4f z = { 0, 0, 0, 0 };
    size_t ntiles = n4 / 256;
    for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x) ST(p[t * 256 + threadIdx.x], z);
}
// V5: 1024-thread WGs grid-stride
__global__ __launch_bounds__(1024) void fill_gridstride1024(v4f* p, size_t n4)
{
    v4f z = { 0, 0, 0, 0 };
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 1024) ST(p[i], z);
}

int main()
{
    const int V = 4, C1 = 18, H = 1000, W = 1000;
    const size_t n = (size_t)V * C1 * H * W, n4 = n / 4, bytes = n * 4;
    v4f* dbase;
    const int NWIN = 8;   // rotate over 8 windows (2.3 GB) so the 256 MB Infinity Cache cannot absorb the writes
    CK(hipMalloc(&dbase, bytes * NWIN));
    v4f* d = dbase;
    int win = 0;
    hipEvent_t evb, eve;
    CK(hipEventCreate(&evb)); CK(hipEventCreate(&eve));
    auto timeit = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; i++) launch();
        hipDeviceSynchronize();
        float best = 1e9, tot = 0;
        for (int r = 0; r < 10; r++) {
            win = (win + 1) % NWIN; d = dbase + (size_t)win * n4;
            hipEventRecord(evb, 0); launch(); hipEventRecord(eve, 0); hipEventSynchronize(eve);
            float ms; hipEventElapsedTime(&ms, evb, eve); best = ms < best ? ms : best; tot += ms;
        }
        printf("%-44s best %7.1f us (%5.0f GB/s)  avg %7.1f us\n", name, best * 1e3, bytes / best / 1e6, tot * 100);
    };
    printf("buffer %.1f MB\n", bytes / 1e6);
    timeit("hipMemsetAsync", [&] { hipMemsetAsync(d, 0, bytes, 0); });
    for (int G : { 512, 1024, 2048, 4096, 8192, 16384 }) {
        char nm[64]; snprintf(nm, 64, "gridstride 256thr G=%d", G);
        timeit(nm, [&] { hipLaunchKernelGGL(fill_gridstride, dim3(G), dim3(256), 0, 0, d, n4); });
    }
    for (int G : { 256, 512, 1024, 2048 }) {
        char nm[64]; snprintf(nm, 64, "gridstride 1024thr G=%d", G);
        timeit(nm, [&] { hipLaunchKernelGGL(fill_gridstride1024, dim3(G), dim3(1024), 0, 0, d, n4); });
    }
    for (int kb : { 4, 16, 64, 256, 1024 }) {
        int chunk4 = kb * 1024 / 16; int G = (int)((n4 + chunk4 - 1) / chunk4);
        char nm[64]; snprintf(nm, 64, "chunk %d KB/WG (G=%d)", kb, G);
        timeit(nm, [&] { hipLaunchKernelGGL(fill_chunk, dim3(G), dim3(256), 0, 0, d, n4, chunk4); });
    }
    for (int lds : { 0, 5120, 16384 }) {
        int chunk4 = 4 * 1024 / 16; int G = (int)((n4 + chunk4 - 1) / chunk4);
        char nm[64]; snprintf(nm, 64, "chunk 4 KB/WG, %d B dynamic LDS, ~50 VGPR", lds);
        timeit(nm, [&] { hipLaunchKernelGGL(fill_chunk_dynlds, dim3(G), dim3(256), lds, 0, d, n4, chunk4, 0); });
    }
    unsigned* dcover;
    CK(hipMalloc(&dcover, 4096));
    CK(hipMemset(dcover, 0, 4096));
    for (int nskip : { 0, 1088 }) {
        int chunk4 = 16 * 1024 / 16; int G = (int)((n4 + chunk4 - 1) / chunk4) + nskip;
        char nm[64]; snprintf(nm, 64, "chunk16K + cover load + LDS, %d skip blocks", nskip);
        timeit(nm, [&] { hipLaunchKernelGGL(fill_chunk_like, dim3(G), dim3(256), 0, 0, d, n4, chunk4, dcover, nskip); });
    }
    for (int G : { 1024, 2048, 4096 }) {
        char nm[64]; snprintf(nm, 64, "persistent 4KB tiles G=%d", G);
        timeit(nm, [&] { hipLaunchKernelGGL(fill_tiles, dim3(G), dim3(256), 0, 0, d, n4, 0); });
    }
    const int band4 = 16 * W / 4, bands = (H + 15) / 16;  // last band overruns a little inside the buffer: fine for timing
    for (int passes : { 1, 2, 4, 16 }) {
        int chunks = (band4 + passes * 256 - 1) / (passes * 256);
        char nm[64]; snprintf(nm, 64, "planes x%d, %d passes (G=%d)", V * C1, passes, chunks * (bands - 1));
        timeit(nm, [&] { hipLaunchKernelGGL(fill_planes, dim3(chunks, bands - 1), dim3(256), 0, 0, d, (size_t)H * W / 4, V * C1, band4, passes); });
    }
    return 0;
}
