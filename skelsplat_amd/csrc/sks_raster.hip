// sks_raster.hip -- skeletal-Gaussian rasterizer for MI355X (gfx950): geometry, binning, compositing, C ABI.
//
// Replaces DGR/cuda_rasterizer/{forward.cu,backward.cu,rasterizer_impl.cu} + DGR/rasterize_points.cu of the
// reference ("DGR/" = submodules/diff-gaussian-rasterization-h36m/).  Design:
//   * one launch sequence renders V views that share the Gaussian parameters (blockIdx.z = view);
//   * P <= SKS_SMALL_P ("skeleton" regime, the reference's real configs have P = 15..19): NO global binning.
//     Forward = one kernel with two roles, fill blocks streaming zeros over everything no Gaussian rect covers and
//     composite blocks for the covered tiles (lists ordered by (depth bits, index): the order the reference gets
//     from its stable radix sort of (tile | depth) keys with index-major emission);
//     backward = gather by Gaussian: a workgroup re-composites the pixels of one Gaussian's rect and keeps only that
//     Gaussian's terms (no atomics, reproducible); final_T / n_contrib are never read back from HBM;
//   * larger P: tile-centric binning (count -> scan -> scatter -> per-tile sort), the same fill + composite forward
//     over the non-empty tiles, one workgroup per non-empty tile in backward;
//   * HBM-bound: no MFMA anywhere (there is no dense contraction in this path).
//
// One translation unit, laid out as included parts (all inside the anonymous namespace below, in this order):
//   sks_common.inc     error helpers, launch-timing hook, constants, per-view records, scratch carving
//   sks_geom_fwd.inc   k_geom_fwd (preprocessCUDA), k_mark_visible
//   sks_fwd_small.inc  LDS list + compositing core, fwd_fill_role, k_render_fwd_sparse
//   sks_bwd_small.inc  backward cores, k_render_bwd_gather / k_render_bwd_wave (+ fused loss), k_gt_tile_stats
//   sks_geom_bwd.inc   k_geom_bwd, k_step_tail (geometry backward + Adam + next geometry in one workgroup)
//   sks_binned.inc     k_bin_*, k_render_fwd_binned, k_render_bwd_binned
// followed here by the launchers and the extern "C" entry points of include/skelsplat_hip.h.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdarg.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

#include "../../include/skelsplat_hip.h"
#include "sks_math.h"
#include "sks_loop_dev.h"

using namespace sks;

namespace {

#include "sks_common.inc"
#include "sks_geom_fwd.inc"
#include "sks_fwd_small.inc"
#include "sks_bwd_small.inc"
#include "sks_geom_bwd.inc"
#include "sks_binned.inc"

// ------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------
inline int pick_cg(int C) { return C <= 4 ? 4 : C <= 16 ? 16 : C <= 20 ? 20 : 32; }

int check_common(int V, int P, int C, int W, int H)
{
    if (V < 1 || V > SKS_MAX_VIEWS) return fail(-1, "V=%d out of range [1,%d]", V, SKS_MAX_VIEWS);
    if (P < 0) return fail(-1, "P=%d negative", P);
    if (C < 1 || C > SKS_MAX_CHANNELS) return fail(-1, "C=%d out of range [1,%d]", C, SKS_MAX_CHANNELS);
    if (W < 1 || H < 1 || W > 4096 * TILE || H > 65535 * TILE) return fail(-1, "image %dx%d out of range", W, H);
    return 0;
}

// Fill-block geometry shared by the two forward launchers.  Row-aligned mode when the image width keeps >= 90% of
// the lanes of a 1024-pixel chunk busy AND a row is a whole number of 128-byte lines (1920, 2048 ... wide images;
// measured +2% at 1920, +12% on the 2048-wide stress config): blocks per band = chunks * row blocks (~8 per band),
// `pb` is returned NEGATIVE = -(rows per block).  Otherwise linear mode, pb passes of 4 KB per block: every pass is
// 32 whole lines whatever the width (at W = 1000 a row is 31.25 lines and the row-aligned mode loses 7% to the
// partial lines at both ends of every row).
// 16-byte stores need every band of every plane on a 16-byte boundary: W % 4 == 0, or W % 4 == 2 with an even H (then a
// float4 is two independently masked pairs, fwd_fill_role<.., HALF>); anything else streams 4 bytes per lane
inline bool fill_half_mode(const FwdArgs& a) { return a.W % 4 == 2 && a.H % 2 == 0 && !(a.flags & SKS_NO_NT_STORES); }

inline void fill_geometry(const FwdArgs& a, bool have_cover, bool binned, int& fsplit, int& pb)
{
    const int ppt = (a.W % 4 == 0 || fill_half_mode(a)) ? 4 : 1;
    // (+31: a band that does not start on a 128-byte line begins with a short pass, fwd_fill_role's `shift`, which the
    // kernel derives from the ADDRESS: the plane and band strides must be whole lines and so must the two base pointers --
    // torch allocations are, a direct C-ABI caller's 16-byte aligned sub-buffer need not be)
    const bool lines = (((uintptr_t)a.out_color | (uintptr_t)a.out_invdepth) & 127u) == 0;
    const int lead = (lines && (long long)a.H * a.W % 32 == 0 && TILE * a.W % 32 == 0) ? 0 : 31;
    const int passes = (TILE * a.W + lead + 256 * ppt - 1) / (256 * ppt);
    const int tune = (int)((a.flags >> 8) & 0xff);                    // tuning knob: passes (or rows) per fill block
    const int chunks = (a.W + 1023) / 1024;
    const bool rowmode = ((ppt == 4 && a.W % 4 == 0 && have_cover && a.W % 32 == 0 && (long long)a.W * 10 >= (long long)chunks * 1024 * 9) ||
                          (ppt == 4 && a.W % 4 == 0 && have_cover && (a.flags & SKS_FILL_ROWS))) && !(a.flags & SKS_FILL_LINEAR);
    if (rowmode) {
        int pbr = tune;
        // rows per block, interleaved A/B on one box: 1920 wide small path 2 rows 860 us / 3: 898 / 4: 907 / 6: 915;
        // 2048 wide binned path (8 composite-role blocks per row) 2 rows 0.632 ms / 3: 0.579 / 4: 0.637 / 6: 0.571
        if (pbr <= 0) pbr = binned ? 3 : 2;
        if (pbr > TILE) pbr = TILE;
        fsplit = chunks * ((TILE + pbr - 1) / pbr);
        pb = -pbr;
        return;
    }
    pb = tune;
    if (pb <= 0) pb = passes / 8 > 2 ? (passes + 4) / 8 : 2;          // ~8 fill blocks per (plane, band) row measured best
    if (pb > 32) pb = 32;                                             // (one skip bit per pass)
    fsplit = (passes + pb - 1) / pb;                                  // fill blocks per (plane, band) row
}

template <int CG>
void launch_fwd_small(const FwdArgs& a_in, int V, int gy, hipStream_t st, const ProfScope& prof)
{
    FwdArgs a = a_in;
    const int ncomp = a.tslots * a.P * V;
    a.ncomp = ncomp;
    int fsplit, pb;
    fill_geometry(a, a.g.cover != nullptr, false, fsplit, pb);
    const int rows_zy = (a.C + 1) * V * gy;
    const int xc = (ncomp + rows_zy - 1) / rows_zy;                   // composite blocks appended to every row
    const int cap = (a.P + 15) & ~15;
    const size_t lds = DynList<CG>::bytes(cap, CG);
    dim3 grid(fsplit + xc, gy, (a.C + 1) * V);
    const bool nt = !(a.flags & SKS_NO_NT_STORES);
    if (a.W % 4 == 0) {
        if (nt) SKS_LAUNCH(prof, (k_render_fwd_sparse<CG, 4, true>), grid, dim3(256), lds, st, a, a.cp1_magic, gy, fsplit, pb, (const uint32_t*)a.g.cover);
        else SKS_LAUNCH(prof, (k_render_fwd_sparse<CG, 4, false>), grid, dim3(256), lds, st, a, a.cp1_magic, gy, fsplit, pb, (const uint32_t*)a.g.cover);
    } else if (fill_half_mode(a)) {
        SKS_LAUNCH(prof, (k_render_fwd_sparse<CG, 4, true, true>), grid, dim3(256), lds, st, a, a.cp1_magic, gy, fsplit, pb, (const uint32_t*)a.g.cover);
    } else {
        SKS_LAUNCH(prof, (k_render_fwd_sparse<CG, 1, false>), grid, dim3(256), lds, st, a, a.cp1_magic, gy, fsplit, pb, (const uint32_t*)a.g.cover);
    }
}

template <int CG>
void launch_fwd_binned(const FwdArgs& a, const BinView& bv, int V, int gx, int gy, const uint32_t* cover, hipStream_t st)
{
    int fsplit, pb;
    fill_geometry(a, true, true, fsplit, pb);
    const int xc = ((gx + TGROUP - 1) / TGROUP + a.C) / (a.C + 1);   // composite blocks (TGROUP tile columns each) spread over the rows
    dim3 grid(fsplit + xc, gy, (a.C + 1) * V);
    const bool nt = !(a.flags & SKS_NO_NT_STORES);
    if (a.W % 4 == 0) {
        if (nt) hipLaunchKernelGGL((k_render_fwd_binned<CG, 4, true>), grid, dim3(256), 0, st, a, bv, gx, gy, fsplit, pb, cover);
        else hipLaunchKernelGGL((k_render_fwd_binned<CG, 4, false>), grid, dim3(256), 0, st, a, bv, gx, gy, fsplit, pb, cover);
    } else if (fill_half_mode(a)) {
        hipLaunchKernelGGL((k_render_fwd_binned<CG, 4, true, true>), grid, dim3(256), 0, st, a, bv, gx, gy, fsplit, pb, cover);
    } else {
        hipLaunchKernelGGL((k_render_fwd_binned<CG, 1, false>), grid, dim3(256), 0, st, a, bv, gx, gy, fsplit, pb, cover);
    }
}

// workgroups of the binned backward: a fixed number per compute unit of the current device, each walking the tile list
inline int bwd_tile_blocks()
{
    static const int n = [] {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) {
            hipDeviceProp_t pr;
            if (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) cus = pr.multiProcessorCount;
        }
        const char* e = getenv("SKS_BWD_TILE_BLOCKS");   // tuning sweeps only
        return (e && atoi(e) > 0) ? atoi(e) : cus * BWD_TILE_BLOCKS_PER_CU;
    }();
    return n;
}

// workgroups per (view, Gaussian) of k_render_bwd_wave (they share the BWD_SPLITS partial-sum slots; see the kernel)
inline int bwd_wave_groups(int V, int P, unsigned flags)
{
    unsigned t = (flags >> SKS_BWD_WG_SHIFT) & 7u;
    static const int env = [] { const char* e = getenv("SKS_BWD_WG"); return e ? atoi(e) : 0; }();   // tuning sweeps only
    if (!t && env > 0) t = (unsigned)env;
    if (t) return t > 5 ? 1 : (BWD_SPLITS >> (t - 1));
    const int pairs = V * P;
    return pairs <= BWD_WG_PAIRS16 ? 16 : (pairs <= BWD_WG_PAIRS8 ? 8 : 4);
}

// the fused-loss variant of the wave-resident backward: heat-maps as planes or (a.hm_row) as separable factors
inline void launch_bwd_loss(const BwdArgs& a, const ViewTan& vt, const ViewOff& vo, dim3 grid, int cg, hipStream_t st)
{
    if (a.hm_row) {
        switch (cg) {
            case 4: hipLaunchKernelGGL((k_render_bwd_wave<4, false, true, true>), grid, dim3(256), 0, st, a, vt, vo); break;
            case 16: hipLaunchKernelGGL((k_render_bwd_wave<16, false, true, true>), grid, dim3(256), 0, st, a, vt, vo); break;
            case 20: hipLaunchKernelGGL((k_render_bwd_wave<20, false, true, true>), grid, dim3(256), 0, st, a, vt, vo); break;
            default: hipLaunchKernelGGL((k_render_bwd_wave<32, false, true, true>), grid, dim3(256), 0, st, a, vt, vo); break;
        }
        return;
    }
    switch (cg) {
        case 4: hipLaunchKernelGGL((k_render_bwd_wave<4, false, true>), grid, dim3(256), 0, st, a, vt, vo); break;
        case 16: hipLaunchKernelGGL((k_render_bwd_wave<16, false, true>), grid, dim3(256), 0, st, a, vt, vo); break;
        case 20: hipLaunchKernelGGL((k_render_bwd_wave<20, false, true>), grid, dim3(256), 0, st, a, vt, vo); break;
        default: hipLaunchKernelGGL((k_render_bwd_wave<32, false, true>), grid, dim3(256), 0, st, a, vt, vo); break;
    }
}

template <int CG>
void launch_bwd_small(const BwdArgs& a, const ViewTan& vt, const ViewOff& vo, int V, int gy, bool dfeat, hipStream_t st,
                      const ProfScope& prof)
{
    (void)gy;
    // slot index slowest: a (view, Gaussian) only has work for its first ceil(pixels / 256) slots, so the idle workgroups
    // (half of them on the skeleton scenes) come last in dispatch order instead of holding wave slots the working ones
    // wait for (the fused-loss kernel fits 3 workgroups per CU: 768 of H36M's 1 088 at once)
    if (a.P <= 64 && !(a.flags & (1u << 20))) {  // wave-resident variant (bit 20: force the LDS variant, tests)
        dim3 grid(a.P, V, bwd_wave_groups(V, a.P, a.flags));
        if (dfeat) SKS_LAUNCH(prof, (k_render_bwd_wave<CG, true, false>), grid, dim3(256), 0, st, a, vt, vo);
        else SKS_LAUNCH(prof, (k_render_bwd_wave<CG, false, false>), grid, dim3(256), 0, st, a, vt, vo);
        return;
    }
    dim3 grid(a.P, V, BWD_SPLITS);
    const size_t lds = GatherLds<CG>::bytes((a.P + 15) & ~15, CG, a.C);
    if (lds > 48 * 1024) {  // gfx950 has 160 KB of LDS per CU; raise the per-kernel dynamic limit when P is large
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_render_bwd_gather<CG, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_render_bwd_gather<CG, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    if (dfeat) SKS_LAUNCH(prof, (k_render_bwd_gather<CG, true>), grid, dim3(256), lds, st, a);
    else SKS_LAUNCH(prof, (k_render_bwd_gather<CG, false>), grid, dim3(256), lds, st, a);
}

// sks_prof_spin: one wavefront that does nothing for a given time (the 100 MHz wall clock) -- stands in for the wire time of a
// latency-bound collective when one GPU measures what a rank of many would see (bench.py)
__global__ void k_spin_us(unsigned long long ticks)
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

// What sks_forward_backward hands its forward (forward_impl; sks_forward passes none): the points of the forward's launch sequence
// behind which the backward's launches go to the second stream.
struct FwdHook {
    hipEvent_t geom_done;                              // small path: rides on k_geom_fwd's own dispatch (no marker packet), or null
    int (*after_geom)(void* ctx);                      // small path: right behind the geometry kernel
    int (*after_group)(void* ctx, int g, int v0, int nv);   // binned path: behind the forward launch of view group g (views v0 .. v0 + nv)
    void* ctx;
};
constexpr int FB_MAX_EVENTS = BIN_MAX_GROUPS;
constexpr int FB_OVERLAP_MAX_PAIRS = BWD_WG_PAIRS16;   // (view, Gaussian) pairs up to which sks_forward_backward runs the backward beside the forward
struct FbEvents {   // created by a thread's first combined call, reused by every later one on the same device
    hipEvent_t geom = nullptr, done = nullptr;
    hipEvent_t grp[FB_MAX_EVENTS] = {};
    int dev = -1;
};
thread_local FbEvents tl_fb_events;

}  // namespace

#ifndef SKS_GEOM_BINNED_THREADS
#define SKS_GEOM_BINNED_THREADS 256
#endif

extern "C" {

const char* sks_last_error(void) { return g_err; }
void sks_set_error_(const char* msg)  // used by the other translation units of the library
{
    snprintf(g_err, sizeof(g_err), "%s", msg);
}
int sks_version(void) { return 11; }

int sks_scratch_bytes(int V, int P, int C, int W, int H, size_t bin_capacity, size_t* geom, size_t* binning, size_t* accum)
{
    if (int rc = check_common(V, P, C, W, H)) return rc;
    const int NT = ((W + TILE - 1) / TILE) * ((H + TILE - 1) / TILE);
    if (geom) *geom = geom_bytes(V, P > 0 ? P : 1, W, H, C);
    if (binning) *binning = bin_bytes(V, P, NT, bin_capacity, C + 1, (size_t)((H + TILE - 1) / TILE) * cover_cw(W));
    if (accum) *accum = (size_t)V * (P > 0 ? P : 1) * BWD_SPLITS * (NACC + C) * sizeof(float);
    return 0;
}

}  // extern "C"

namespace {

int forward_impl(int V, int P, int C, int W, int H, const float* viewmatrix, const float* projmatrix,
                 const float* tanfovx, const float* tanfovy, const float* means3D, const float* features,
                 const float* opacities, const float* scales, const float* rotations, const float* cov3D_precomp,
                 float scale_modifier, unsigned flags, float* out_color, float* out_invdepth, int* radii, void* geom,
                 void* binning, size_t bin_capacity, int* num_rendered_dev, float* final_T, uint32_t* n_contrib,
                 void* stream, const FwdHook* hook)
{
    if (int rc = check_common(V, P, C, W, H)) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (!out_color || !out_invdepth) return fail(-2, "out_color / out_invdepth must be provided");
    if (((uintptr_t)out_color | (uintptr_t)out_invdepth) & 15u)
        return fail(-2, "out_color / out_invdepth must be 16-byte aligned (the planes are written with 16-byte stores)");
    const size_t HW = (size_t)H * W;
    if (P == 0) {  // rasterize_points.cu:88: nothing rendered, outputs stay zero
        HIP_TRY(hipMemsetAsync(out_color, 0, (size_t)V * C * HW * 4, st));
        HIP_TRY(hipMemsetAsync(out_invdepth, 0, (size_t)V * HW * 4, st));
        if (final_T) return fail(-2, "final_T not available for P == 0");
        return 0;
    }
    if (!viewmatrix || !projmatrix || !tanfovx || !tanfovy || !means3D || !features || !opacities || !radii || !geom)
        return fail(-2, "missing required pointer");
    if (!cov3D_precomp && (!scales || !rotations)) return fail(-2, "need scales+rotations or cov3D_precomp");
    ViewTan vt;
    ViewOff vo;
    fill_views(vt, &vo, V, C, W, H, tanfovx, tanfovy, nullptr, nullptr);
    Geom g = geom_from(geom, V, P, W, H);
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE, NT = gx * gy;

    const bool small = P <= SKS_SMALL_P && !(flags & SKS_FORCE_BINNED);
    Bin b{};
    if (!small) {
        if (!binning) return fail(-2, "binned path needs a binning buffer");
        if (bin_capacity < 1) return fail(-1, "binned path needs bin_capacity >= 1");
        b = bin_from(binning, V, P, NT, bin_capacity);
        // (nothing to clear: k_geom_fwd zeroes the header, k_bin_band_count writes every tile's count)
    }
    if (!small) g.cover = nullptr;   // (the binned path's cover rows are per plane: Bin::coverp, k_bin_scan + k_bin_sort_long)
    const int gthreads = small ? 256 : SKS_GEOM_BINNED_THREADS;
    const int gplanes = (small && g.cover && cover_per_plane(P, W, H, C)) ? C + 1 : 1;     // a block per plane's cover rows
    if (hook && small && hook->geom_done)
        hipExtLaunchKernelGGL(k_geom_fwd, dim3((P + gthreads - 1) / gthreads, V, gplanes), dim3(gthreads), 0, st, nullptr, hook->geom_done, 0,
                              P, W, H, vt, viewmatrix, projmatrix, means3D, opacities, scales, rotations, cov3D_precomp, scale_modifier,
                              flags, g, radii, 0, (uint32_t*)nullptr, (uint32_t*)nullptr, features, C, (uint2*)nullptr, (uint32_t*)nullptr,
                              cover_per_plane(P, W, H, C) ? 1 : 0);
    else
    hipLaunchKernelGGL(k_geom_fwd, dim3((P + gthreads - 1) / gthreads, V, gplanes), dim3(gthreads), 0, st, P, W, H, vt, viewmatrix, projmatrix,
                       means3D, opacities, scales, rotations, cov3D_precomp, scale_modifier, flags, g, radii, 0,
                       small ? (uint32_t*)nullptr : b.count, small ? (uint32_t*)nullptr : b.touched, features, C,
                       small ? (uint2*)nullptr : b.fmask, small ? (uint32_t*)nullptr : b.hdr, (small && cover_per_plane(P, W, H, C)) ? 1 : 0);
    STAGE_CHECK("geometry");
    if (hook && small && hook->after_geom) {   // (sks_forward_backward: the backward behind the geometry, on its own stream)
        if (int rc = hook->after_geom(hook->ctx)) return rc;
    }

    FwdArgs a{ P, C, W, H, flags, g, features, out_color, out_invdepth, final_T, n_contrib, composite_slots(flags, V, P),
               ((1u << 20) + (unsigned)C) / (unsigned)(C + 1), 0, (!small || (g.cover && cover_per_plane(P, W, H, C))) ? 1 : 0,
               (W >= 2 && W < 16384) ? (unsigned)(((1ull << 32) + (unsigned)W - 1) / (unsigned)W) : 0u };
    const int cg = pick_cg(C);
    if (small) {
        {
            ProfScope prof(0, st, true);
            switch (cg) {
                case 4: launch_fwd_small<4>(a, V, gy, st, prof); break;
                case 16: launch_fwd_small<16>(a, V, gy, st, prof); break;
                case 20: launch_fwd_small<20>(a, V, gy, st, prof); break;
                default: launch_fwd_small<32>(a, V, gy, st, prof); break;
            }
        }
        STAGE_CHECK("render(small)");
        return 0;
    }
    uint32_t* cover = b.coverp;   // a row per (view, plane, band)
    const int cw = cover_cw(W);
    hipLaunchKernelGGL(k_bin_band_count, dim3(gy, V), dim3(BAND_TC), (size_t)2 * gx * 4, st, P, gx, g, b);
    int vg, ng;
    bin_groups(flags, V, vg, ng);
    {
        const int bpc = gx >= SCAN_T * 8 ? 1 : (SCAN_T * 8) / gx;   // whole tile bands per scan block (<= SCAN_T * SCAN_IPT tiles)
        const int nchunk_t = (gy + bpc - 1) / bpc, nchunk_g = (P + SCAN_G - 1) / SCAN_G;
        hipLaunchKernelGGL(k_bin_scan, dim3(nchunk_t + nchunk_g, V), dim3(SCAN_T), (size_t)bpc * cw * 4, st, P, gx, gy, cw, bpc,
                           nchunk_t, bin_capacity, b, cover, num_rendered_dev, V, C + 1, vg);
    }
    hipLaunchKernelGGL(k_bin_band_scatter, dim3(gy, V), dim3(BAND_TS), (size_t)2 * gx * 4, st, P, gx, bin_capacity, g, b, C, cw);
    STAGE_CHECK("binning");
    const BinView bv_all = bin_view(b, NT, bin_capacity);
    {
        // one fill + composite launch per view group (one group = every view unless SKS_BIN_GROUPS asks for more): a group's launch
        // sees ITS views as views 0 .. nv - 1 -- every per-view array is handed over from the group's first view on; only the tile
        // descriptors (BinView::tlist, addressed through Bin::tidx) keep their call-wide positions
        ProfScope prof(0, st);
        for (int gi = 0; gi < ng; gi++) {
            const int v0 = gi * vg, nv = (V - v0 < vg) ? V - v0 : vg;
            FwdArgs ag = a;
            ag.out_color = a.out_color + (size_t)v0 * C * HW;
            ag.out_invdepth = a.out_invdepth + (size_t)v0 * HW;
            if (a.final_T) ag.final_T = a.final_T + (size_t)v0 * HW;
            if (a.n_contrib) ag.n_contrib = a.n_contrib + (size_t)v0 * HW;
            BinView bv = bv_all;
            bv.ranges += (size_t)v0 * NT;
            bv.tidx += (size_t)v0 * NT;
            bv.aux += (size_t)v0 * NT * TILE * TILE;
            bv.e_co += (size_t)v0 * bin_capacity;
            bv.e_xyd += (size_t)v0 * bin_capacity;
            bv.e_ids += (size_t)v0 * bin_capacity;
            bv.part += (size_t)v0 * bin_capacity * BIN_ROWS * NACC;
            const uint32_t* cov_g = cover + (size_t)v0 * (C + 1) * gy * cw;
            switch (cg) {
                case 4: launch_fwd_binned<4>(ag, bv, nv, gx, gy, cov_g, st); break;
                case 16: launch_fwd_binned<16>(ag, bv, nv, gx, gy, cov_g, st); break;
                case 20: launch_fwd_binned<20>(ag, bv, nv, gx, gy, cov_g, st); break;
                default: launch_fwd_binned<32>(ag, bv, nv, gx, gy, cov_g, st); break;
            }
            if (hook && hook->after_group)
                if (int rc = hook->after_group(hook->ctx, gi, v0, nv)) return rc;
        }
    }
    STAGE_CHECK("render(binned)");
    return 0;
}

}  // namespace

extern "C" {

int sks_forward(int V, int P, int C, int W, int H, const float* viewmatrix, const float* projmatrix,
                const float* tanfovx, const float* tanfovy, const float* means3D, const float* features,
                const float* opacities, const float* scales, const float* rotations, const float* cov3D_precomp,
                float scale_modifier, unsigned flags, float* out_color, float* out_invdepth, int* radii, void* geom,
                void* binning, size_t bin_capacity, int* num_rendered_dev, float* final_T, uint32_t* n_contrib,
                void* stream)
{
    return forward_impl(V, P, C, W, H, viewmatrix, projmatrix, tanfovx, tanfovy, means3D, features, opacities, scales, rotations,
                        cov3D_precomp, scale_modifier, flags, out_color, out_invdepth, radii, geom, binning, bin_capacity,
                        num_rendered_dev, final_T, n_contrib, stream, nullptr);
}

}  // extern "C"

namespace {

// sks_backward in two phases, so that sks_forward_backward can place them: BWD_RENDER = the compositing backward (binned path:
// of view group `group`, or of every group in turn when group < 0), BWD_GEOM = the geometry backward behind it (binned path: of
// view group `group`'s views, or of all views when group < 0).
constexpr unsigned BWD_RENDER = 1u, BWD_GEOM = 2u;
int backward_impl(int V, int P, int C, int W, int H, const float* viewmatrix, const float* projmatrix,
                  const float* tanfovx, const float* tanfovy, const float* bg, const float* means3D,
                  const float* features, const float* opacities, const float* scales, const float* rotations,
                  const float* cov3D_precomp, float scale_modifier, unsigned flags, const int* radii, const void* geom,
                  const void* binning, size_t bin_capacity, const float* dL_dout_color, const float* dL_dout_invdepth,
                  void* accum, float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dopacity, float* dL_dscales,
                  float* dL_drotations, float* dL_dcov3D, float* dL_dfeatures, float* dL_dmeans3D_mean, void* stream,
                  unsigned phases, int group)
{
    if (int rc = check_common(V, P, C, W, H)) return rc;
    if (P == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (!viewmatrix || !projmatrix || !tanfovx || !tanfovy || !means3D || !features || !opacities || !radii ||
        !geom || !dL_dout_color || !accum || !dL_dmeans3D || !dL_dmeans2D || !dL_dopacity)
        return fail(-2, "missing required pointer");
    if (!cov3D_precomp && (!scales || !rotations)) return fail(-2, "need scales+rotations or cov3D_precomp");
    ViewTan vt;
    ViewOff vo;
    fill_views(vt, &vo, V, C, W, H, tanfovx, tanfovy, nullptr, nullptr);
    Geom g = geom_from(const_cast<void*>(geom), V, P, W, H);
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE, NT = gx * gy;
    BwdArgs a{ P, C, W, H, flags, g, features, bg, dL_dout_color, dL_dout_invdepth, (float*)accum, nullptr, nullptr };
    const int cg = pick_cg(C);
    const bool dfeat = dL_dfeatures != nullptr;
    const bool small = P <= SKS_SMALL_P && !(flags & SKS_FORCE_BINNED);
    int vg = V, ng = 1;
    if (!small) bin_groups(flags, V, vg, ng);
    if (group >= ng) return fail(-1, "view group %d of %d", group, ng);
    if ((phases & BWD_RENDER) && small) {
        ProfScope prof(1, st, true);
        switch (cg) {
            case 4: launch_bwd_small<4>(a, vt, vo, V, gy, dfeat, st, prof); break;
            case 16: launch_bwd_small<16>(a, vt, vo, V, gy, dfeat, st, prof); break;
            case 20: launch_bwd_small<20>(a, vt, vo, V, gy, dfeat, st, prof); break;
            default: launch_bwd_small<32>(a, vt, vo, V, gy, dfeat, st, prof); break;
        }
        STAGE_CHECK("render-backward(small)");
    } else if (phases & BWD_RENDER) {
        if (!binning) return fail(-2, "binned path needs the forward's binning buffer");
        Bin b = bin_from(const_cast<void*>(binning), V, P, NT, bin_capacity);
        BinView bv = bin_view(b, NT, bin_capacity);
        // the per-Gaussian sums go through the slot rows of the binning scratch (plain stores, every row written); only the
        // feature gradient, when wanted, is accumulated with atomics
        if (dfeat && group <= 0) HIP_TRY(hipMemsetAsync(accum, 0, (size_t)V * P * (NACC + C) * sizeof(float), st));
        dim3 grid(bwd_tile_blocks());
        const unsigned magic_nt = (unsigned)(((1ull << 32) + (unsigned)NT - 1) / (unsigned)NT), magic_gx = (unsigned)(((1ull << 32) + (unsigned)gx - 1) / (unsigned)gx);
        const bool extra = bg != nullptr || dL_dout_invdepth != nullptr;
        for (int gi = group < 0 ? 0 : group; gi < (group < 0 ? ng : group + 1); gi++) {
            // a view group's tiles: its stretch of every class's descriptor region, its class counters (k_bin_scan)
            const uint4* tl = b.tlist + (size_t)gi * vg * NT;
            const uint32_t* hd = b.hdr + gi * BIN_CLASSES;
            ProfScope prof(1, st);
            if (dfeat) hipLaunchKernelGGL((k_render_bwd_tile<true, true>), grid, dim3(BWD_TILE_THREADS), 0, st, a, bv, gx, V, magic_nt, magic_gx, tl, hd);
            else if (extra) hipLaunchKernelGGL((k_render_bwd_tile<false, true>), grid, dim3(BWD_TILE_THREADS), 0, st, a, bv, gx, V, magic_nt, magic_gx, tl, hd);
            else hipLaunchKernelGGL((k_render_bwd_tile<false, false>), grid, dim3(BWD_TILE_THREADS), 0, st, a, bv, gx, V, magic_nt, magic_gx, tl, hd);
        }
        STAGE_CHECK("render-backward(binned)");
    }
    if (!(phases & BWD_GEOM)) return 0;
    GeomBwdArgs ga{ P, C, W, H, flags, viewmatrix, projmatrix, means3D, opacities, scales, rotations, cov3D_precomp,
                    scale_modifier, radii, (const float*)accum, small ? BWD_SPLITS : 0, nullptr, nullptr, nullptr, dL_dmeans3D, dL_dmeans2D, dL_dopacity, dL_dscales,
                    dL_drotations, dL_dcov3D, dL_dfeatures };
    if (!small) {
        Bin b = bin_from(const_cast<void*>(binning), V, P, NT, bin_capacity);
        ga.rect = g.rect;
        ga.goff = b.goff;
        ga.part = b.part;
        ga.cap = bin_capacity;
        ga.bin_hdr = b.hdr;
    }
    if (dL_dmeans3D_mean && V * P <= 256 && small) {   // (the binned path's k_geom_bwd also re-arms the work cursors)
        hipLaunchKernelGGL(k_geom_bwd_all, dim3(1), dim3(256), 0, st, ga, vt, V, dL_dmeans3D_mean);
    } else {
        if (small) hipLaunchKernelGGL(k_geom_bwd, dim3((P + 255) / 256, V), dim3(256), 0, st, ga, vt);
        else {
            const int v0 = group < 0 ? 0 : group * vg, nv = group < 0 ? V : ((V - v0 < vg) ? V - v0 : vg);
            ga.v0 = v0;
            hipLaunchKernelGGL(k_geom_bwd_binned, dim3((P + GEOMB_G - 1) / GEOMB_G, nv), dim3(256), 0, st, ga, vt);
        }
        // (the mean over the views wants every view's gradients: with view groups, behind the last group's geometry backward)
        if (dL_dmeans3D_mean && (small || group < 0 || group == ng - 1))
            hipLaunchKernelGGL(k_mean_views, dim3((3 * P + 255) / 256), dim3(256), 0, st, V, P, dL_dmeans3D, dL_dmeans3D_mean);
    }
    STAGE_CHECK("geometry-backward");
    return 0;
}

}  // namespace

extern "C" {

int sks_backward(int V, int P, int C, int W, int H, const float* viewmatrix, const float* projmatrix,
                 const float* tanfovx, const float* tanfovy, const float* bg, const float* means3D,
                 const float* features, const float* opacities, const float* scales, const float* rotations,
                 const float* cov3D_precomp, float scale_modifier, unsigned flags, const int* radii, const void* geom,
                 const void* binning, size_t bin_capacity, const float* dL_dout_color, const float* dL_dout_invdepth,
                 void* accum, float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dopacity, float* dL_dscales,
                 float* dL_drotations, float* dL_dcov3D, float* dL_dfeatures, float* dL_dmeans3D_mean, void* stream)
{
    return backward_impl(V, P, C, W, H, viewmatrix, projmatrix, tanfovx, tanfovy, bg, means3D, features, opacities, scales, rotations,
                         cov3D_precomp, scale_modifier, flags, radii, geom, binning, bin_capacity, dL_dout_color, dL_dout_invdepth, accum,
                         dL_dmeans3D, dL_dmeans2D, dL_dopacity, dL_dscales, dL_drotations, dL_dcov3D, dL_dfeatures, dL_dmeans3D_mean,
                         stream, BWD_RENDER | BWD_GEOM, -1);
}

int sks_forward_backward(int V, int P, int C, int W, int H, const float* viewmatrix, const float* projmatrix,
                         const float* tanfovx, const float* tanfovy, const float* means3D, const float* features,
                         const float* opacities, const float* scales, const float* rotations, const float* cov3D_precomp,
                         float scale_modifier, unsigned flags, float* out_color, float* out_invdepth, int* radii, void* geom,
                         void* binning, size_t bin_capacity, int* num_rendered_dev, const float* bg,
                         const float* dL_dout_color, const float* dL_dout_invdepth, void* accum, float* dL_dmeans3D,
                         float* dL_dmeans2D, float* dL_dopacity, float* dL_dscales, float* dL_drotations, float* dL_dcov3D,
                         float* dL_dfeatures, float* dL_dmeans3D_mean, void* stream, void* aux_stream, unsigned fb_flags)
{
    const bool small = P <= SKS_SMALL_P && !(flags & SKS_FORCE_BINNED);
    if (!small && aux_stream && aux_stream != stream && !((flags >> SKS_BIN_GROUPS_SHIFT) & 7u)) {
        // the binned path overlaps by VIEW GROUPS (the backward of a group starts from what the forward's compositor left per
        // pixel of ITS views): unless the caller chose a count, BIN_FB_GROUPS of them
        static const int env = [] { const char* e = getenv("SKS_BIN_GROUPS"); return e ? atoi(e) : 0; }();   // tuning sweeps
        const int want = env > 0 ? env : BIN_FB_GROUPS;
        flags |= SKS_BIN_GROUPS(want < 1 ? 1 : (want > BIN_MAX_GROUPS ? BIN_MAX_GROUPS : want));
    }
    int vg = V, ng = 1;
    if (!small && V >= 1) bin_groups(flags, V, vg, ng);
    struct Ctx {
        int V, P, C, W, H;
        const float *vm, *pm, *tx, *ty, *bg, *means, *feat, *opac, *scales, *rots, *cov;
        float smod;
        unsigned flags;
        const int* radii;
        const void *geom, *binning;
        size_t cap;
        const float *dL, *dLinv;
        void* accum;
        float *m3, *m2, *op, *sc, *rot, *dcov, *dfeat, *mean;
        hipStream_t s, aux;
        bool ext, handed;     // handed: the second stream holds work of this call
        int rc_aux;
        int backward(unsigned phases, int group) const
        {
            return backward_impl(V, P, C, W, H, vm, pm, tx, ty, bg, means, feat, opac, scales, rots, cov, smod, flags, radii, geom, binning, cap,
                                 dL, dLinv, accum, m3, m2, op, sc, rot, dcov, dfeat, mean, aux, phases, group);
        }
    } c{ V, P, C, W, H, viewmatrix, projmatrix, tanfovx, tanfovy, bg, means3D, features, opacities, scales, rotations, cov3D_precomp,
         scale_modifier, flags, radii, geom, binning, bin_capacity, dL_dout_color, dL_dout_invdepth, accum, dL_dmeans3D, dL_dmeans2D,
         dL_dopacity, dL_dscales, dL_drotations, dL_dcov3D, dL_dfeatures, dL_dmeans3D_mean, (hipStream_t)stream, (hipStream_t)aux_stream,
         true, false, 0 };
    {
        static const bool ext_off = [] { const char* e = getenv("SKS_FB_EXT"); return e && atoi(e) == 0; }();   // tuning
        if (ext_off) c.ext = false;
    }
    const bool no_join = (fb_flags & SKS_FB_NO_JOIN) != 0;
    // The step's form per workload: beside the forward the small path's backward pays while it is a handful of workgroups (H36M's 4
    // views: 1 088, step 66 -> 58 us; one rank's 4 Panoptic views) and costs more than it hides once its wavefronts crowd the fill
    // blocks out (all 31 Panoptic views, 589 (view, Gaussian) pairs: 0.908 ms beside, 0.893 behind -- the forward 817 -> 866 us, the
    // backward 56 -> 133).  Beyond FB_OVERLAP_MAX_PAIRS the call runs its two halves one after the other -- unless the caller
    // keeps the second stream's tail for itself (SKS_FB_NO_JOIN: a sharded step's collective wants the gradients THERE).
    const bool crowded = small && (long long)V * P > FB_OVERLAP_MAX_PAIRS && !no_join;
    if (!aux_stream || aux_stream == stream || P == 0 || (flags & SKS_DEBUG_SYNC) || (!small && ng < 2) || crowded) {
        // nothing to run side by side: one after the other on the caller's stream
        if (int rc = forward_impl(V, P, C, W, H, viewmatrix, projmatrix, tanfovx, tanfovy, means3D, features, opacities, scales, rotations,
                                  cov3D_precomp, scale_modifier, flags, out_color, out_invdepth, radii, geom, binning, bin_capacity,
                                  num_rendered_dev, nullptr, nullptr, stream, nullptr))
            return rc;
        return backward_impl(V, P, C, W, H, viewmatrix, projmatrix, tanfovx, tanfovy, bg, means3D, features, opacities, scales, rotations,
                             cov3D_precomp, scale_modifier, flags, radii, geom, binning, bin_capacity, dL_dout_color, dL_dout_invdepth,
                             accum, dL_dmeans3D, dL_dmeans2D, dL_dopacity, dL_dscales, dL_drotations, dL_dcov3D, dL_dfeatures,
                             dL_dmeans3D_mean, stream, BWD_RENDER | BWD_GEOM, -1);
    }
    FbEvents& ev = tl_fb_events;
    int cur_dev = 0;
    HIP_TRY(hipGetDevice(&cur_dev));
    if (ev.geom && ev.dev != cur_dev) {   // (events belong to a device: a thread that moved to another one gets new ones)
        (void)hipEventDestroy(ev.geom);
        (void)hipEventDestroy(ev.done);
        for (hipEvent_t& e : ev.grp)
            if (e) { (void)hipEventDestroy(e); e = nullptr; }
        ev.geom = ev.done = nullptr;
    }
    // hand-over between two queues of ONE device: a device-scope release is all the waiting side needs (the default,
    // a system-scope fence, is what a host reader of the event would want)
    unsigned evf = hipEventDisableTiming | hipEventDisableSystemFence;
    if (const char* e = getenv("SKS_FB_EVENT_FLAGS")) evf = (unsigned)strtoul(e, nullptr, 0);   // tuning
    if (!ev.geom) {
        ev.dev = cur_dev;
        HIP_TRY(hipEventCreateWithFlags(&ev.geom, evf));
        HIP_TRY(hipEventCreateWithFlags(&ev.done, evf));
    }
    if (!small)
        for (int gi = 0; gi < ng; gi++)
            if (!ev.grp[gi]) HIP_TRY(hipEventCreateWithFlags(&ev.grp[gi], evf));
    struct Run {
        // small path: behind the geometry kernel, i.e. behind everything the caller enqueued -- the whole backward
        static int after_geom(void* p)
        {
            Ctx& k = *(Ctx*)p;
            FbEvents& e = tl_fb_events;
            if (!k.ext) HIP_TRY(hipEventRecord(e.geom, k.s));   // (else it rode on k_geom_fwd's dispatch)
            HIP_TRY(hipStreamWaitEvent(k.aux, e.geom, 0));
            k.handed = true;
            return k.rc_aux = k.backward(BWD_RENDER | BWD_GEOM, -1);
        }
        // binned path: view group g's forward is enqueued -- its compositing and geometry backward go to the second stream, where
        // they run beside the forward of group g + 1
        static int after_group(void* p, int g, int, int)
        {
            Ctx& k = *(Ctx*)p;
            FbEvents& e = tl_fb_events;
            HIP_TRY(hipEventRecord(e.grp[g], k.s));
            HIP_TRY(hipStreamWaitEvent(k.aux, e.grp[g], 0));
            k.handed = true;
            return k.rc_aux = k.backward(BWD_RENDER | BWD_GEOM, g);
        }
    };
    const FwdHook hook{ (small && c.ext) ? ev.geom : nullptr, small ? &Run::after_geom : nullptr, small ? nullptr : &Run::after_group, &c };
    int rc = forward_impl(V, P, C, W, H, viewmatrix, projmatrix, tanfovx, tanfovy, means3D, features, opacities, scales, rotations,
                          cov3D_precomp, scale_modifier, flags, out_color, out_invdepth, radii, geom, binning, bin_capacity,
                          num_rendered_dev, nullptr, nullptr, stream, &hook);
    if (rc == 0 && !c.handed) rc = fail(-3, "sks_forward_backward: the forward did not reach the point the backward starts from");
    // The gradients are the caller's in stream order -- unless the caller has more to enqueue behind the backward on aux_stream
    // (a view-sharded step's collective: hidden under the forward as well) and joins the two streams itself.  After an ERROR
    // behind the hand-over the join is made whatever the caller asked for: what the second stream already holds reads `geom` and
    // writes the gradient tensors, and the caller's stream must not run ahead of it.
    if (c.handed && (!no_join || rc != 0)) {
        const hipError_t e1 = hipEventRecord(ev.done, c.aux);
        const hipError_t e2 = e1 == hipSuccess ? hipStreamWaitEvent(c.s, ev.done, 0) : e1;
        if (rc == 0 && e2 != hipSuccess) rc = fail((int)e2, "sks_forward_backward: joining the two streams: %s", hipGetErrorString(e2));
    }
    return rc;
}

int sks_mean_views(int V, int P, const float* dL_dmeans3D, int shard_world, float* mean_out, void* stream)
{
    if (V < 1 || P < 1 || shard_world < 1) return fail(-1, "mean_views: bad shape");
    if (!dL_dmeans3D || !mean_out) return fail(-2, "mean_views: missing pointer");
    hipLaunchKernelGGL(k_mean_views, dim3((3 * P + 255) / 256), dim3(256), 0, (hipStream_t)stream, V, P, dL_dmeans3D, mean_out,
                       shard_world);
    HIP_TRY(hipGetLastError());
    return 0;
}

int sks_gt_tile_stats(int V, int C, int W, int H, const float* gt, float* tile_S, float* tile_N, double* totals, void* stream)
{
    if (int rc = check_common(V, 1, C, W, H)) return rc;
    if (!gt || !totals) return fail(-2, "gt_tile_stats: missing pointer");
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(totals, 0, (size_t)V * 2 * sizeof(double), st));
    dim3 grid((H + TILE - 1) / TILE, C, V);
    hipLaunchKernelGGL(k_gt_tile_stats, grid, dim3(256), 0, st, C, W, H, gt, tile_S, tile_N, totals);
    HIP_TRY(hipGetLastError());
    return 0;
}

int sks_backward_fused_loss(int V, int P, int C, int W, int H, const float* viewmatrix, const float* projmatrix,
                            const float* tanfovx, const float* tanfovy, const float* bg, const float* means3D,
                            const float* features, const float* opacities, const float* scales, const float* rotations,
                            const float* cov3D_precomp, float scale_modifier, unsigned flags, const int* radii,
                            const void* geom, const float* gt, const float* tile_S, const float* tile_N,
                            const double* gt_totals, void* accum, float* dL_dmeans3D, float* dL_dmeans2D,
                            float* dL_dopacity, float* dL_dscales, float* dL_drotations, float* dL_dcov3D,
                            double* loss_sums, float* packed_raw_grads, const int* view_wh, const size_t* gt_offsets,
                            const float* const* hm_factors, void* stream)
{
    if (int rc = check_common(V, P, C, W, H)) return rc;
    if (P < 1 || P > 64) return fail(-1, "fused-loss backward needs 1 <= P <= 64 (got %d)", P);
    hipStream_t st = (hipStream_t)stream;
    if (!viewmatrix || !projmatrix || !tanfovx || !tanfovy || !means3D || !features || !opacities || !radii || !geom ||
        (!gt && !hm_factors) || !gt_totals || !accum || !dL_dmeans3D || !dL_dmeans2D || !dL_dopacity || !loss_sums)
        return fail(-2, "missing required pointer");
    if (hm_factors && (!hm_factors[0] || !hm_factors[1] || !hm_factors[2] || !hm_factors[3]))
        return fail(-2, "hm_factors: row, col, cmin, den must all be given");
    if (!cov3D_precomp && (!scales || !rotations)) return fail(-2, "need scales+rotations or cov3D_precomp");
    ViewTan vt;
    ViewOff vo;
    if (!fill_views(vt, &vo, V, C, W, H, tanfovx, tanfovy, view_wh, gt_offsets))
        return fail(-1, "view_wh: every view's size must be within [1, W] x [1, H] (pass the largest as W, H)");
    if (!hm_factors && (view_wh != nullptr) != (gt_offsets != nullptr)) return fail(-2, "view_wh and gt_offsets go together");
    Geom g = geom_from(const_cast<void*>(geom), V, P, W, H);
    BwdArgs a{ P, C, W, H, flags | SKS_CLAMP01, g, features, bg, gt, nullptr, (float*)accum, tile_S, tile_N };
    if (hm_factors) { a.hm_row = hm_factors[0]; a.hm_col = hm_factors[1]; a.hm_cmin = hm_factors[2]; a.hm_den = hm_factors[3]; }
    dim3 grid(P, V, bwd_wave_groups(V, P, flags));   // see launch_bwd_small
    {
        ProfScope prof(1, st);
        launch_bwd_loss(a, vt, vo, grid, pick_cg(C), st);
    }
    STAGE_CHECK("fused loss + render-backward");
    GeomBwdArgs ga{ P, C, W, H, flags, viewmatrix, projmatrix, means3D, opacities, scales, rotations, cov3D_precomp,
                    scale_modifier, radii, (const float*)accum, BWD_SPLITS, gt_totals, loss_sums, packed_raw_grads, dL_dmeans3D, dL_dmeans2D,
                    dL_dopacity, dL_dscales, dL_drotations, dL_dcov3D, nullptr };
    hipLaunchKernelGGL(k_geom_bwd, dim3((P + 255) / 256, V), dim3(256), 0, st, ga, vt);
    STAGE_CHECK("geometry-backward");
    return 0;
}

int sks_geometry(int V, int P, int C, int W, int H, const float* viewmatrix, const float* projmatrix, const float* tanfovx,
                 const float* tanfovy, const float* means3D, const float* opacities, const float* scales,
                 const float* rotations, const float* cov3D_precomp, float scale_modifier, unsigned flags, int* radii,
                 void* geom, const int* view_wh, int frames, void* stream)
{
    if (int rc = check_common(V, P, C, W, H)) return rc;
    if (P < 1) return fail(-1, "P must be positive");
    if (frames < 1 || V % frames) return fail(-1, "frames must divide the number of views (V = %d, frames = %d)", V, frames);
    if (!viewmatrix || !projmatrix || !tanfovx || !tanfovy || !means3D || !opacities || !radii || !geom)
        return fail(-2, "missing required pointer");
    if (!cov3D_precomp && (!scales || !rotations)) return fail(-2, "need scales+rotations or cov3D_precomp");
    hipStream_t st = (hipStream_t)stream;
    ViewTan vt;
    if (!fill_views(vt, nullptr, V, C, W, H, tanfovx, tanfovy, view_wh, nullptr))
        return fail(-1, "view_wh: every view's size must be within [1, W] x [1, H] (pass the largest as W, H)");
    Geom g = geom_from(geom, V, P, W, H);
    g.cover = nullptr;   // (the cover rows serve the dense forward: sks_forward makes its own)
    hipLaunchKernelGGL(k_geom_fwd, dim3((P + 255) / 256, V), dim3(256), 0, st, P, W, H, vt, viewmatrix, projmatrix,
                       means3D, opacities, scales, rotations, cov3D_precomp, scale_modifier, flags, g, radii,
                       frames > 1 ? V / frames : 0);
    STAGE_CHECK("geometry");
    return 0;
}

int sks_loop_fused_step(int V, int P, int C, int W, int H, const float* viewmatrix, const float* projmatrix,
                        const float* tanfovx, const float* tanfovy, const float* features, float scale_modifier,
                        unsigned flags, int* radii, void* geom, const float* gt, const double* gt_totals, void* accum,
                        double* loss_sums, float* packed, float* slots, unsigned long long group_mask, int last_view,
                        float* xyz, float* scaling, float* rotation, float* opacity, float* exp_avg, float* exp_avg_sq,
                        int* counters, int acc_steps, const double* lr_sched, const double* lrs, const double* adam,
                        float lambda_consistency, const int* limb, const int* view_wh, const size_t* gt_offsets, int frames,
                        const float* const* hm_factors, void* stream)
{
    if (int rc = check_common(V, P, C, W, H)) return rc;
    if (P < 1 || P > 64) return fail(-1, "fused step needs 1 <= P <= 64 (got %d)", P);
    if (frames < 1 || V % frames) return fail(-1, "frames must divide the number of views (V = %d, frames = %d)", V, frames);
    const int Vf = V / frames;   // views of one frame: the optimiser's V (slots, group mask, last_view are per frame)
    if (!viewmatrix || !projmatrix || !tanfovx || !tanfovy || !features || !radii || !geom || (!gt && !hm_factors) || !gt_totals ||
        !accum || !loss_sums || !packed)
        return fail(-2, "missing required pointer");
    if (hm_factors && (!hm_factors[0] || !hm_factors[1] || !hm_factors[2] || !hm_factors[3]))
        return fail(-2, "hm_factors: row, col, cmin, den must all be given");
    hipStream_t st = (hipStream_t)stream;
    ViewTan vt;
    ViewOff vo;
    if (!fill_views(vt, &vo, V, C, W, H, tanfovx, tanfovy, view_wh, gt_offsets))
        return fail(-1, "view_wh: every view's size must be within [1, W] x [1, H] (pass the largest as W, H)");
    if (!hm_factors && (view_wh != nullptr) != (gt_offsets != nullptr)) return fail(-2, "view_wh and gt_offsets go together");
    flags |= SKS_RAW_PARAMS | SKS_CLAMP01;
    sksloop::AdamArgs aa;
    if (const char* err = sksloop::fill_adam_args(aa, Vf, P, packed, slots, group_mask, last_view, xyz, scaling, rotation, opacity,
                                                  exp_avg, exp_avg_sq, counters, acc_steps, lr_sched, lrs, adam,
                                                  lambda_consistency, limb))
        return fail(-2, "%s", err);
    Geom g = geom_from(geom, V, P, W, H);
    g.cover = nullptr;   // no forward render on this path
    BwdArgs a{ P, C, W, H, flags, g, features, nullptr, gt, nullptr, (float*)accum, nullptr, nullptr };
    if (hm_factors) { a.hm_row = hm_factors[0]; a.hm_col = hm_factors[1]; a.hm_cmin = hm_factors[2]; a.hm_den = hm_factors[3]; }
    dim3 grid(P, V, bwd_wave_groups(V, P, flags));   // see launch_bwd_small
    {
        ProfScope prof(1, st);
        launch_bwd_loss(a, vt, vo, grid, pick_cg(C), st);
    }
    STAGE_CHECK("render-backward(fused loss)");
    GeomBwdArgs ga{ P, C, W, H, flags, viewmatrix, projmatrix, xyz, opacity, scaling, rotation, nullptr, scale_modifier, radii,
                    (const float*)accum, BWD_SPLITS, gt_totals, loss_sums, packed, nullptr, nullptr, nullptr, nullptr, nullptr,
                    nullptr, nullptr };
    hipLaunchKernelGGL(k_step_tail, dim3(frames), dim3(256), 0, st, ga, vt, aa, Vf, g, radii);
    STAGE_CHECK("step tail");
    return 0;
}

int sks_prof_spin(double microseconds, void* stream)
{
    if (!(microseconds >= 0.0) || microseconds > 1e6) return fail(-1, "prof_spin: 0 .. 1e6 us");
    hipLaunchKernelGGL(k_spin_us, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long)(microseconds * 100.0));
    HIP_TRY(hipGetLastError());
    return 0;
}

int sks_prof_enable(int on)
{
    if (!tl_prof) {
        if (!on) return 0;
        tl_prof = new ProfState();
    }
    tl_prof->on = on != 0;
    tl_prof->every = (on & 0xffff) > 1 ? (on & 0xffff) : 1;
    tl_prof->skip = (on >> 16) & 3;
    tl_prof->recorded = (on >> 18) & 1;
    tl_prof->seen[0] = tl_prof->seen[1] = 0;
    return 0;
}

int sks_prof_count(int kind, long long* launches)
{
    if (kind < 0 || kind > 1 || !launches) return fail(-2, "bad profile query");
    *launches = tl_prof ? tl_prof->kind[kind].n : 0;
    return 0;
}

int sks_prof_read(int kind, double* total_ms, long long* launches)
{
    if (kind < 0 || kind > 1 || !total_ms || !launches) return fail(-2, "bad profile query");
    if (!tl_prof) { *total_ms = 0.0; *launches = 0; return 0; }
    ProfKind& p = tl_prof->kind[kind];
    double tot = 0;
    for (int i = 0; i < p.n; i++) {
        HIP_TRY(hipEventSynchronize(p.e[i]));
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, p.b[i], p.e[i]));
        tot += ms;
    }
    *total_ms = tot;
    *launches = p.n;
    p.n = 0;
    return 0;
}

int sks_prof_read_quantiles(int kind, double* q_ms /* 3: p10, p50, p90 */, double* total_ms, long long* launches)
{
    if (kind < 0 || kind > 1 || !q_ms || !total_ms || !launches) return fail(-2, "bad profile query");
    if (!tl_prof) { q_ms[0] = q_ms[1] = q_ms[2] = 0.0; *total_ms = 0.0; *launches = 0; return 0; }
    ProfKind& p = tl_prof->kind[kind];
    static thread_local float t[PROF_MAX];
    double tot = 0;
    for (int i = 0; i < p.n; i++) {
        HIP_TRY(hipEventSynchronize(p.e[i]));
        HIP_TRY(hipEventElapsedTime(&t[i], p.b[i], p.e[i]));
        tot += t[i];
    }
    const int n = p.n;
    for (int i = 1; i < n; i++) {   // insertion sort: n is a few dozen samples
        const float v = t[i];
        int j = i - 1;
        for (; j >= 0 && t[j] > v; j--) t[j + 1] = t[j];
        t[j + 1] = v;
    }
    q_ms[0] = n ? t[(int)(0.1 * (n - 1) + 0.5)] : 0.0;
    q_ms[1] = n ? t[(n - 1) / 2] : 0.0;
    q_ms[2] = n ? t[(int)(0.9 * (n - 1) + 0.5)] : 0.0;
    *total_ms = tot;
    *launches = n;
    p.n = 0;
    return 0;
}

int sks_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present,
                     void* stream)
{
    (void)projmatrix;  // in_frustum only uses the view-space depth (auxiliary.h:166)
    if (P < 0) return fail(-1, "P negative");
    if (P == 0) return 0;
    if (!means3D || !viewmatrix || !present) return fail(-2, "missing required pointer");
    hipLaunchKernelGGL(k_mark_visible, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, means3D, viewmatrix, present);
    HIP_TRY(hipGetLastError());
    return 0;
}

int sks_export_lists(int V, int W, int H, const void* binning, size_t bin_capacity, uint32_t* point_list,
                     uint32_t* ranges, void* stream)
{
    if (!binning || !point_list || !ranges) return fail(-2, "missing required pointer");
    const int NT = ((W + TILE - 1) / TILE) * ((H + TILE - 1) / TILE);
    Bin b = bin_from(const_cast<void*>(binning), V, 1, NT, bin_capacity);   // (nothing per Gaussian is read: its arrays come last)
    const size_t m = bin_capacity > (size_t)NT ? bin_capacity : (size_t)NT;
    hipLaunchKernelGGL(k_export_lists, dim3((unsigned)((m + 255) / 256), V), dim3(256), 0, (hipStream_t)stream, NT,
                       bin_capacity, b.ranges, b.nrend, point_list, ranges);
    hipLaunchKernelGGL(k_export_sorted, dim3((NT + 3) / 4, V), dim3(256), 0, (hipStream_t)stream, bin_view(b, NT, bin_capacity),
                       point_list);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // extern "C"
