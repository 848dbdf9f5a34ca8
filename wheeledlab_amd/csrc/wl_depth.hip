// wl_depth.hip -- depth ray-cast of the visual task's camera against a heightfield (BASELINE.json config 5) for gfx950.
//
// distance_to_image_plane per pixel of the 60 x 80 pinhole camera (visual/mushr_visual_env_cfg.py:230-246; what
// mdp_sensors/observations.py:89-95 `camera_data_depth` / `raycast_depth` forward from IsaacLab's camera) against the
// terrain SOLID of wl_heightfield.h: bilinear patches on a regular grid, the plane z = outside_z beyond it.  Spec:
// oracle/depth.c (exact per-cell intersection in double, walking cell by cell).
//
// lane = ray.  A ray's ground track is walked through a MAX-PYRAMID of the field (level L cell = 2^L x 2^L grid cells,
// value = highest corner inside): where the ray stays above a cell's maximum over the cell's whole parameter interval the
// cell is skipped in one step and the walk climbs a level when it crosses into a new parent; otherwise it descends, and
// at level 0 the patch along the ray is the quadratic g(s) = A s^2 + B s + C whose first root in the cell is the hit --
// exact, no marching step.  Sky rays leave after ~log2(grid) steps, ground rays after a descent plus the few fine cells in
// front of the hit; near-horizontal rays skimming the surface are the long ones (the pyramid cannot skip what the ray
// nearly touches).
//
// Mapping: block = ONE wavefront = one 4 x 16 pixel tile of one env's image (75 tiles per image): neighbouring pixels take
// nearly the same walk, so the lanes stay together and their gathers hit the same cache lines; a tile's walk length varies
// 5 x over the image (sky / near ground short, the rows at the horizon long), and single-wavefront blocks let the dispatcher
// refill a slot the moment its tile is done (five-wavefront strip blocks held four finished wavefronts' slots until the
// slowest one ended: 716 against 611 us at 4096 cameras, same walk).  The field's copy (16-bit codes: 1.28 MB) and the pyramid
// (0.85 MB touched: 4-byte entries) fit an XCD's L2 twice over.  Output is the only HBM stream: 19 200 B per env, 64 B segments per tile row.
#include <hip/hip_runtime.h>

#include "../../include/wheeledlab_amd.h"
#include "wl_kernel_common.h"
#include "wl_depth_dev.h"

namespace {

constexpr int kStripRows = 4, kStrips = WL_VIS_IMG_H / kStripRows;
constexpr int kTileCols = 16, kTilesPerStrip = WL_VIS_IMG_W / kTileCols, kTiles = kStrips * kTilesPerStrip;   // 75 tiles per image
static_assert(WL_VIS_IMG_H % kStripRows == 0 && WL_VIS_IMG_W % kTileCols == 0 && kStripRows * kTileCols == 64, "one wavefront per tile");

// the header: the field's range and its steepest cell edge (one block: this runs once per heightfield); the maximum also goes to float 0
__global__ void __launch_bounds__(1024) pyramid_header_kernel(const WlHeightField f, float* __restrict__ buf, const int hdr) {
    __shared__ float red[3][16];
    __shared__ int bad;
    float hmin = INFINITY, hmax = -INFINITY, smax = 0.f;
    bool finite = true;
    if (threadIdx.x == 0) bad = 0;
    __syncthreads();
    const int64_t n = (int64_t)f.nx * f.ny;
    for (int64_t k = threadIdx.x; k < n; k += 1024) {
        const int j = (int)(k / f.nx);
        header_point(f, (int)(k - (int64_t)j * f.nx), j, hmin, hmax, smax, finite);
    }
    if (!finite) bad = 1;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        hmin = fminf(hmin, __shfl_xor(hmin, off, 64));
        hmax = fmaxf(hmax, __shfl_xor(hmax, off, 64));
        smax = fmaxf(smax, __shfl_xor(smax, off, 64));
    }
    if ((threadIdx.x & 63) == 0) red[0][threadIdx.x >> 6] = hmin, red[1][threadIdx.x >> 6] = hmax, red[2][threadIdx.x >> 6] = smax;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) hmin = fminf(hmin, red[0][w]), hmax = fmaxf(hmax, red[1][w]), smax = fmaxf(smax, red[2][w]);
        pyramid_header_values(hmin, hmax, smax, bad == 0, buf + hdr);
        buf[0] = hmax;
    }
}
// the bound pyramid, level by level straight from the heights (exact residuals: every grid point of a cell is visited).
// Small cells (L <= 4: at most 17 x 17 points): one thread per cell.
__global__ void __launch_bounds__(kBlock) pyramid_planes_small_kernel(const WlHeightField f, float* __restrict__ buf, const int lp, const int L, const int hdr) {
    const int W = (1 << lp) >> L;
    const int k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= W * W) return;
    reinterpret_cast<uint32_t*>(buf)[pyramid_level_offset(lp, L) + k] = plane_cell_serial(f, L, k % W, k / W, buf + hdr);
}
// Large cells: one block per cell, its points strided over the threads, maxima folded through LDS.
__global__ void __launch_bounds__(kBlock) pyramid_planes_large_kernel(const WlHeightField f, float* __restrict__ buf, const int lp, const int L, const int hdr) {
    __shared__ float red[2][kBlock / 64];
    const int W = (1 << lp) >> L;
    const int I = blockIdx.x % W, J = blockIdx.x / W;
    uint32_t* e = reinterpret_cast<uint32_t*>(buf) + pyramid_level_offset(lp, L) + blockIdx.x;
    int i0, i1, j0, j1;
    if (!plane_cell_range(f, L, I, J, i0, i1, j0, j1)) {     // block-uniform
        if (threadIdx.x == 0) *e = kEmptyEntry;
        return;
    }
    int a8, b8;
    float a, b, resid = -INFINITY, hmax = -INFINITY;
    plane_cell_slopes(f, i0, i1, j0, j1, buf[hdr + kPyrSlopeQ], a8, b8, a, b);
    const int wpts = i1 - i0 + 1, npts = wpts * (j1 - j0 + 1);
    for (int k = threadIdx.x; k < npts; k += kBlock) plane_point(f, i0, j0, i0 + k % wpts, j0 + k / wpts, a, b, resid, hmax);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        resid = fmaxf(resid, __shfl_xor(resid, off, 64));
        hmax = fmaxf(hmax, __shfl_xor(hmax, off, 64));
    }
    if ((threadIdx.x & 63) == 0) red[0][threadIdx.x >> 6] = resid, red[1][threadIdx.x >> 6] = hmax;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; ++w) resid = fmaxf(resid, red[0][w]), hmax = fmaxf(hmax, red[1][w]);
        *e = plane_entry(i0, i1, j0, j1, a8, b8, a, b, resid, hmax, buf + hdr);
    }
}
__global__ void __launch_bounds__(kBlock) pyramid_copy_heights_kernel(const int16_t* __restrict__ h, int16_t* __restrict__ dst, const int n) {
    const int k = blockIdx.x * kBlock + threadIdx.x;
    if (k < n) dst[k] = h[k];
}

// camera pose of env e: origin and the rotation of its body frame
struct DepthCam {
    V3 o;
    Mat3 R;
};
WL_DEV DepthCam depth_cam(const WlVisualParams& p, const WlEnvBuffers& b, int e) {
    const Rows S = make_rows(b.state, b.stride);
    const V3 pos = ld3(S, WL_S_PX, e);
    const Quat q{S.ld(WL_S_QW, e), S.ld(WL_S_QX, e), S.ld(WL_S_QY, e), S.ld(WL_S_QZ, e)};
    DepthCam c;
    c.R = mat_from_quat(q);
    c.o = pos + mul(c.R, v3(p.cam_pos[0], p.cam_pos[1], p.cam_pos[2]));
    return c;
}

// one ray per lane, one 4 x 16 tile per wavefront: the wavefront lives as long as its longest ray (lanes busy 0.6 of the steps on
// the bench poses) -- and is still the fastest form measured (the ray-pool form below: 1.3 - 2 x slower).
// ROWS names what the launch writes -- 0: the 60 x 80 image of wl_visual_depth, 1: the observation rows of the visual-depth task
// (stride 4808) -- same code; two instantiations so that a kernel-statistics summary keeps the two workloads apart.
template <int ROWS>
__global__ void __launch_bounds__(64) visual_depth_tile_kernel(const WlVisualParams p, const WlEnvBuffers b, const DepthGrid g,
                                                                const Pyramid py, const float* __restrict__ buf, const unsigned buf_bytes,
                                                                const float max_depth, float* __restrict__ depth, const int64_t row_stride) {
    const int e = blockIdx.x / kTiles, tile = blockIdx.x - e * kTiles;
    const int strip = tile / kTilesPerStrip;
    const DepthCam cam = depth_cam(p, b, e);
    const int lane = threadIdx.x;
    const int row = strip * kStripRows + (lane >> 4), col = (tile - strip * kTilesPerStrip) * kTileCols + (lane & 15);
    const V3 d = mul(cam.R, depth_pixel_ray_body(p, row, col));
    const FieldMem mem{__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(buf), 0, (int)buf_bytes, 0x00020000)};
    const float t = cast_ray(g, py, pyramid_head(g, py, mem), mem, cam.o, d, max_depth);
    // non-temporal: the image rows must not push the pyramid and the heights (4.3 MB for the 800 x 800 field, an XCD's L2 holds 4 MB)
    // out of L2 -- round 4, 4096 images: counter reads 176 -> 69 MB per render (the compulsory fill is 8 XCDs x 4.3 MB), 380 -> 372 us
    __builtin_nontemporal_store(t, depth + (int64_t)e * row_stride + row * WL_VIS_IMG_W + col);
}

// Ray POOL per wavefront (round 4; MEASURED SLOWER, not the default: -DWL_DEPTH_POOL_ROWS=4 | 12 | 20 | 60 builds it).  A tile's
// walk lengths differ 5 x between lanes (sky 2 - 3 steps, near ground 8 - 10, the rows below the horizon 15 - 20, grazing rays
// 100 +): with one ray per lane 40 % of the lane-steps are idle lanes waiting for the tile's longest ray.  Here a wavefront owns
// POOL_ROWS image rows of one env (in 4 x 16 tile order, so that the lanes start as neighbours) and a lane whose ray is done takes
// the next ray of the pool: whenever at least THRESH lanes are idle they are refilled in ONE pass of the set-up code (ballot +
// prefix count give each idle lane its pool index).  The host simulation of the same walk (tests/host_sim) promised 11.8
// wave-steps per 64 rays against 15.2 (lanes busy 0.80 instead of 0.60).  On the device, 4096 cameras, us per render: tile form
// 545; pools of 4 rows 718, 12 rows 839 (THRESH 12 / 20 / 32: 849 / 839 / 893; non-temporal stores 802), 20 rows 890, 60 rows
// 1083.  What the simulation does not see: lanes that hold rays from different tiles at different depths of their walks gather
// from 64 unrelated places per step (the tile form's neighbours share their cache lines), and the fewer, longer wavefronts
// balance worse over the chip.  Idle lanes are the cheaper evil.
#ifndef WL_DEPTH_POOL_ROWS
#define WL_DEPTH_POOL_ROWS 0     // 0: the tile form
#endif
#ifndef WL_DEPTH_POOL_THRESH
#define WL_DEPTH_POOL_THRESH 20
#endif
template <int POOL_ROWS, int THRESH>
__global__ void __launch_bounds__(64) visual_depth_pool_kernel(const WlVisualParams p, const WlEnvBuffers b, const DepthGrid g,
                                                                const Pyramid py, const float* __restrict__ buf, const unsigned buf_bytes,
                                                                const float max_depth, float* __restrict__ depth, const int64_t row_stride) {
    static_assert(POOL_ROWS % kStripRows == 0 && WL_VIS_IMG_H % POOL_ROWS == 0, "whole strips per pool, whole pools per image");
    constexpr int kPools = WL_VIS_IMG_H / POOL_ROWS, kPool = POOL_ROWS * WL_VIS_IMG_W;
    const int e = blockIdx.x / kPools, r0 = (blockIdx.x - e * kPools) * POOL_ROWS;
    const DepthCam cam = depth_cam(p, b, e);
    const FieldMem mem{__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(buf), 0, (int)buf_bytes, 0x00020000)};
    const PyrHead hd = pyramid_head(g, py, mem);
    const float zclear = hd.zclear;
    float* img = depth + (int64_t)e * row_stride + r0 * WL_VIS_IMG_W;
    const int lane = threadIdx.x;
    const int max_walk = max_walk_steps(g);
    // pool index -> pixel of the pool's rows: tile (q >> 6) = strip-major 4 x 16 tiles, (q & 63) = row-major inside the tile
    auto pixel = [](int q, int& row, int& col) {
        const int tile = q >> 6, in = q & 63;
        const int strip = (int)(__umul24((unsigned)tile, 13108u) >> 16);    // tile / 5 for tile < 2^14
        row = strip * kStripRows + (in >> 4);
        col = (tile - strip * kTilesPerStrip) * kTileCols + (in & 15);
    };
    // pixel -> ray through the reciprocal focal lengths (one division each per wavefront, not two per ray set-up)
    const float ifx = 1.f / p.fx, ify = 1.f / p.fy;
    auto ray_dir = [&](int prow, int pcol) {
        return mul(cam.R, v3(1.f, -(((float)pcol + 0.5f - p.cx) * ifx), -(((float)(r0 + prow) + 0.5f - p.cy) * ify)));
    };
    int q = lane, row, col, steps = 0;
    pixel(q, row, col);
    RayWalk w = ray_begin(g, py, zclear, cam.o, ray_dir(row, col), max_depth);
    bool have = true;
    int next = 64;          // wave-uniform: the first pool index nobody has taken
#pragma unroll 1
    for (;;) {
        if (have) {
            if (w.live && steps < max_walk) {
                ray_step(g, py, hd, mem, w);
                ++steps;
            } else {
#ifdef WL_DEPTH_NT
                __builtin_nontemporal_store(ray_result(g, w), img + row * WL_VIS_IMG_W + col);
#else
                img[row * WL_VIS_IMG_W + col] = ray_result(g, w);
#endif
                have = false;
            }
        }
        const uint64_t idle = __ballot(!have);
        const int n_idle = __popcll(idle);
        if (next < kPool) {
            if (n_idle >= THRESH) {
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(idle >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)idle, 0u));
                const int qn = next + rank;
                if (!have && qn < kPool) {
                    q = qn;
                    pixel(q, row, col);
                    w = ray_begin(g, py, zclear, cam.o, ray_dir(row, col), max_depth);
                    steps = 0;
                    have = true;
                }
                next += n_idle;
            }
        } else if (n_idle == 64) {
            break;
        }
    }
}

// the 8 proprioceptive columns of the visual-depth observation (base_lin_vel | base_ang_vel | last_action clipped) of the state as it
// stands: lane = env (the step launch writes them itself; this is the reset / first-observation path)
__global__ void __launch_bounds__(kBlock) visual_depth_prop_kernel(const WlEnvBuffers b, float* __restrict__ obs) {
    const int e = blockIdx.x * kBlock + threadIdx.x;
    if (e >= b.n_envs) return;
    const Rows S = make_rows(b.state, b.stride);
    const Quat q{S.ld(WL_S_QW, e), S.ld(WL_S_QX, e), S.ld(WL_S_QY, e), S.ld(WL_S_QZ, e)};
    const Mat3 R = mat_from_quat(q);
    const V3 vb = mul_t(R, ld3(S, WL_S_VX, e)), wb = mul_t(R, ld3(S, WL_S_WX, e));
    float* t = obs + (int64_t)e * WL_VISDEPTH_OBS_DIM + WL_VISDEPTH_NPIX;
    t[0] = vb.x, t[1] = vb.y, t[2] = vb.z, t[3] = wb.x, t[4] = wb.y, t[5] = wb.z;
    t[6] = clampf(S.ld(WL_S_ACT0, e), -1.f, 1.f), t[7] = clampf(S.ld(WL_S_ACT1, e), -1.f, 1.f);
}

}  // namespace

extern "C" {

int64_t wl_heightfield_pyramid_floats(int32_t nx, int32_t ny) {
    if (nx < 2 || ny < 2 || nx > 16385 || ny > 16385) return 0;
    return pyramid_total_floats(nx, ny);
}

int wl_heightfield_build_pyramid(const WlHeightField* hf, float* pyramid, void* stream) {
    if (!hf || !hf->height || !pyramid || hf->nx < 2 || hf->ny < 2 || hf->nx > 16385 || hf->ny > 16385 || !(hf->cell > 0.f) || !(hf->z_scale > 0.f && hf->z_scale < INFINITY)) return WL_EINVAL;
    const Pyramid py = make_pyramid(hf->nx, hf->ny);
    clear_error();
    const hipStream_t hs = (hipStream_t)stream;
    const int P = 1 << py.lp;
    pyramid_header_kernel<<<1, 1024, 0, hs>>>(*hf, pyramid, py.hdr);
    for (int L = 1; L <= py.lp; ++L) {
        const int cells = (P >> L) * (P >> L);
        if (L <= 4 && L < py.lp) pyramid_planes_small_kernel<<<grid_for(cells), kBlock, 0, hs>>>(*hf, pyramid, py.lp, L, py.hdr);
        else pyramid_planes_large_kernel<<<cells, kBlock, 0, hs>>>(*hf, pyramid, py.lp, L, py.hdr);
    }
    pyramid_copy_heights_kernel<<<grid_for(hf->nx * hf->ny), kBlock, 0, hs>>>(hf->height, reinterpret_cast<int16_t*>(pyramid + py.h0), hf->nx * hf->ny);
    return launch_status();
}

// every argument check of the depth launch, so that callers that launch something in front of it (wl_visual_depth_step: the step
// kernel advances state, episode_len and metrics) can refuse the WHOLE call before anything has run
static int depth_rows_args_ok(const WlVisualParams* p, const WlEnvBuffers* b, const WlHeightField* hf, const float* pyramid, float max_depth,
                              const float* rows, int64_t row_stride) {
    if (!p || !b || !hf || !b->state || !hf->height || !pyramid || !rows || b->n_envs <= 0 || !(max_depth > 0.f)) return 0;
    if (hf->nx < 2 || hf->ny < 2 || hf->nx > 16385 || hf->ny > 16385 || !(hf->cell > 0.f) || !(hf->z_scale > 0.f && hf->z_scale < INFINITY) || b->stride < b->n_envs || !(p->fx > 0.f) ||
        !(p->fy > 0.f) || row_stride < WL_VISDEPTH_NPIX)
        return 0;
    if (b->stride * 4 * WL_S_COUNT > 0x7fffffffLL || (int64_t)b->n_envs * kTiles > 0x7fffffffLL) return 0;
    return 1;
}

int wl_visual_depth_rows(const WlVisualParams* p, const WlEnvBuffers* b, const WlHeightField* hf, const float* pyramid, float max_depth,
                         float* rows, int64_t row_stride, void* stream) {
    if (!depth_rows_args_ok(p, b, hf, pyramid, max_depth, rows, row_stride)) return WL_EINVAL;
    const Pyramid py = make_pyramid(hf->nx, hf->ny);
    const unsigned bytes = (unsigned)(pyramid_total_floats(hf->nx, hf->ny) * 4);
    clear_error();
#if WL_DEPTH_POOL_ROWS > 0
    visual_depth_pool_kernel<WL_DEPTH_POOL_ROWS, WL_DEPTH_POOL_THRESH><<<b->n_envs * (WL_VIS_IMG_H / WL_DEPTH_POOL_ROWS), 64, 0, (hipStream_t)stream>>>(
        *p, *b, make_depth_grid(hf), py, pyramid, bytes, max_depth, rows, row_stride);
#else
    if (row_stride == WL_VISDEPTH_NPIX)
        visual_depth_tile_kernel<0><<<b->n_envs * kTiles, 64, 0, (hipStream_t)stream>>>(*p, *b, make_depth_grid(hf), py, pyramid, bytes, max_depth, rows, row_stride);
    else
        visual_depth_tile_kernel<1><<<b->n_envs * kTiles, 64, 0, (hipStream_t)stream>>>(*p, *b, make_depth_grid(hf), py, pyramid, bytes, max_depth, rows, row_stride);
#endif
    return launch_status();
}

int wl_visual_depth(const WlVisualParams* p, const WlEnvBuffers* b, const WlHeightField* hf, const float* pyramid, float max_depth,
                    float* depth, void* stream) {
    return wl_visual_depth_rows(p, b, hf, pyramid, max_depth, depth, WL_VISDEPTH_NPIX, stream);
}

int wl_visual_depth_step(const WlVisualParams* p, const WlEnvBuffers* b, const WlTravMap* m, const WlHeightField* hf, const float* pyramid,
                         float max_depth, const float* actions, const WlStepOut* out, uint64_t seed, uint64_t step, void* stream) {
    // all-or-nothing: the depth launch's own checks run BEFORE the step kernel advances the state (a refused call must leave the
    // batch and the caller's step counter where they were)
    if (!out || !depth_rows_args_ok(p, b, hf, pyramid, max_depth, out->obs, WL_VISDEPTH_OBS_DIM)) return WL_EINVAL;
    const int rc = wl_visual_step_hf(p, b, m, hf, actions, out, seed, step, stream);     // validates the rest
    if (rc != WL_OK) return rc;
    return wl_visual_depth_rows(p, b, hf, pyramid, max_depth, out->obs, WL_VISDEPTH_OBS_DIM, stream);
}

int wl_visual_depth_observe(const WlVisualParams* p, const WlEnvBuffers* b, const WlHeightField* hf, const float* pyramid, float max_depth,
                            float* obs, void* stream) {
    const int rc = wl_visual_depth_rows(p, b, hf, pyramid, max_depth, obs, WL_VISDEPTH_OBS_DIM, stream);
    if (rc != WL_OK) return rc;
    visual_depth_prop_kernel<<<grid_for(b->n_envs), kBlock, 0, (hipStream_t)stream>>>(*b, obs);
    return launch_status();
}

}  // extern "C"
