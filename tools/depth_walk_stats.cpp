// Developer probe (not product, not a test): the depth ray-cast's DEVICE walk (wheeledlab_amd/csrc/wl_depth_dev.h) compiled for the
// host with a step counter, over the 4 x 16 pixel tiles the kernel gives its wavefronts: walk steps per ray, wave-steps per tile
// (= the tile's longest ray), lanes busy (ray-steps / 64 x wave-steps), and how many of a tile's wave-steps have a lane in a fine
// cell (level 0: the divergent, expensive half of ray_step).  Driven by tools/depth_walk_stats.py.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <vector>
using std::max;
using std::min;

static thread_local int g_steps, g_fine_mask_step;
static thread_local std::vector<uint8_t>* g_trace;       // per step of the current ray: 1 = fine cell
#define WL_DEPTH_STEP_HOOK(L) (g_trace->push_back((L) == 0 ? 1 : 0))
static thread_local int g_clear_prefix;            // leading steps of the current ray whose cell was cleared (-1 - count once one was not)
#define WL_DEPTH_CLEAR_HOOK(clear, t, te) (g_clear_prefix = g_clear_prefix < 0 ? g_clear_prefix : ((clear) ? g_clear_prefix + 1 : -1 - g_clear_prefix))
#include "wl_depth_dev.h"

static std::vector<float> host_pyramid(const WlHeightField* hf) {
    const Pyramid py = make_pyramid(hf->nx, hf->ny);
    const int P = 1 << py.lp;
    std::vector<float> buf((size_t)pyramid_total_floats(hf->nx, hf->ny), 0.f);
    pyramid_header_serial(*hf, buf.data() + py.hdr);
    buf[0] = buf[py.hdr + kPyrMax];
    uint32_t* words = reinterpret_cast<uint32_t*>(buf.data());
    for (int L = 1; L <= py.lp; ++L)
        for (int J = 0; J < (P >> L); ++J)
            for (int I = 0; I < (P >> L); ++I)
                words[(size_t)pyramid_level_offset(py.lp, L) + (size_t)J * (P >> L) + I] = plane_cell_serial(*hf, L, I, J, buf.data() + py.hdr);
    std::copy(hf->height, hf->height + (size_t)hf->nx * hf->ny, reinterpret_cast<int16_t*>(buf.data() + py.h0));
    return buf;
}

extern "C" {
// the raw traces, for scheduling experiments: trace_out[((e * 75 + tile) * 64 + lane) * max_len + k] = 0 past the ray's end, 1 a coarse
// step, 2 a fine step; returns the longest ray
int dws_traces(const WlVisualParams* p, const WlHeightField* hf, int n, const float* pos, const float* quat, float max_depth, uint8_t* trace_out,
               int max_len) {
    const Pyramid py = make_pyramid(hf->nx, hf->ny);
    const std::vector<float> buf = host_pyramid(hf);
    const DepthGrid g = make_depth_grid(hf);
    const FieldMem mem{buf.data()};
    const PyrHead hd = pyramid_head(g, py, mem);
    std::vector<uint8_t> tr;
    size_t longest = 0;
    for (int e = 0; e < n; ++e) {
        const Quat q{quat[4 * e], quat[4 * e + 1], quat[4 * e + 2], quat[4 * e + 3]};
        const Mat3 R = mat_from_quat(q);
        const V3 o = v3(pos[3 * e], pos[3 * e + 1], pos[3 * e + 2]) + mul(R, v3(p->cam_pos[0], p->cam_pos[1], p->cam_pos[2]));
        for (int tile = 0; tile < 75; ++tile) {
            const int strip = tile / 5, tc = tile % 5;
            for (int lane = 0; lane < 64; ++lane) {
                tr.clear();
                g_trace = &tr;
                (void)cast_ray(g, py, hd, mem, o, mul(R, depth_pixel_ray_body(*p, strip * 4 + (lane >> 4), tc * 16 + (lane & 15))), max_depth);
                longest = std::max(longest, tr.size());
                uint8_t* dst = trace_out + (((size_t)e * 75 + tile) * 64 + lane) * max_len;
                for (size_t k = 0; k < tr.size() && k < (size_t)max_len; ++k) dst[k] = tr[k] ? 2 : 1;
            }
        }
    }
    return (int)longest;
}
// out[0..5]: rays, ray-steps, tiles, wave-steps (sum of per-tile maxima), wave-steps with >= 1 lane in a fine cell, fine ray-steps
int dws_stats(const WlVisualParams* p, const WlHeightField* hf, int n, const float* pos, const float* quat, float max_depth, double* out) {
    const Pyramid py = make_pyramid(hf->nx, hf->ny);
    const std::vector<float> buf = host_pyramid(hf);
    const DepthGrid g = make_depth_grid(hf);
    const FieldMem mem{buf.data()};
    const PyrHead hd = pyramid_head(g, py, mem);
    double rays = 0, ray_steps = 0, tiles = 0, wave_steps = 0, fine_wave_steps = 0, fine_ray_steps = 0;
    std::vector<uint8_t> trace[64];
    for (int e = 0; e < n; ++e) {
        const Quat q{quat[4 * e], quat[4 * e + 1], quat[4 * e + 2], quat[4 * e + 3]};
        const Mat3 R = mat_from_quat(q);
        const V3 o = v3(pos[3 * e], pos[3 * e + 1], pos[3 * e + 2]) + mul(R, v3(p->cam_pos[0], p->cam_pos[1], p->cam_pos[2]));
        for (int strip = 0; strip < WL_VIS_IMG_H / 4; ++strip)
            for (int tc = 0; tc < WL_VIS_IMG_W / 16; ++tc) {
                size_t longest = 0;
                for (int lane = 0; lane < 64; ++lane) {
                    trace[lane].clear();
                    g_trace = &trace[lane];
                    (void)cast_ray(g, py, hd, mem, o, mul(R, depth_pixel_ray_body(*p, strip * 4 + (lane >> 4), tc * 16 + (lane & 15))), max_depth);
                    rays += 1, ray_steps += (double)trace[lane].size();
                    longest = std::max(longest, trace[lane].size());
                }
                tiles += 1, wave_steps += (double)longest;
                for (size_t k = 0; k < longest; ++k) {
                    bool any = false;
                    for (int lane = 0; lane < 64; ++lane)
                        if (k < trace[lane].size() && trace[lane][k]) any = true, fine_ray_steps += 1;
                    if (any) fine_wave_steps += 1;
                }
            }
    }
    out[0] = rays, out[1] = ray_steps, out[2] = tiles, out[3] = wave_steps, out[4] = fine_wave_steps, out[5] = fine_ray_steps;
    return 0;
}
// round 6: what a tile-cooperative start could skip at most.  out[0..1]: the sum over the tiles of the number of leading CLEAR steps
// every ray of the tile shares (the minimum over its rays of the steps before the ray's first not-clear test), the tiles' wave-steps
int dws_shared_prefix(const WlVisualParams* p, const WlHeightField* hf, int n, const float* pos, const float* quat, float max_depth, double* out) {
    const Pyramid py = make_pyramid(hf->nx, hf->ny);
    const std::vector<float> buf = host_pyramid(hf);
    const DepthGrid g = make_depth_grid(hf);
    const FieldMem mem{buf.data()};
    const PyrHead hd = pyramid_head(g, py, mem);
    std::vector<uint8_t> tr;
    double shared = 0, wave_steps = 0;
    for (int e = 0; e < n; ++e) {
        const Quat q{quat[4 * e], quat[4 * e + 1], quat[4 * e + 2], quat[4 * e + 3]};
        const Mat3 R = mat_from_quat(q);
        const V3 o = v3(pos[3 * e], pos[3 * e + 1], pos[3 * e + 2]) + mul(R, v3(p->cam_pos[0], p->cam_pos[1], p->cam_pos[2]));
        for (int tile = 0; tile < 75; ++tile) {
            const int strip = tile / 5, tc = tile % 5;
            int longest = 0, prefix = 1 << 30;
            for (int lane = 0; lane < 64; ++lane) {
                tr.clear();
                g_trace = &tr;
                g_clear_prefix = 0;
                (void)cast_ray(g, py, hd, mem, o, mul(R, depth_pixel_ray_body(*p, strip * 4 + (lane >> 4), tc * 16 + (lane & 15))), max_depth);
                longest = std::max(longest, (int)tr.size());
                if (!tr.empty()) prefix = std::min(prefix, g_clear_prefix < 0 ? -1 - g_clear_prefix : g_clear_prefix);
            }
            wave_steps += longest;
            shared += longest ? std::min(prefix, longest) : 0;
        }
    }
    out[0] = shared, out[1] = wave_steps;
    return 0;
}
}
