// TEST INFRASTRUCTURE ONLY: the depth ray-cast's device walk (wheeledlab_amd/csrc/wl_depth_dev.h: max-pyramid + per-ray
// traversal, fp32) compiled for the host through the stand-in hip_runtime.h and driven over arrays, so that
// tests/test_oracle_depth.py can hold it against oracle/depth.c without a GPU.  Built by the test into a scratch directory.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <vector>
using std::max;
using std::min;

#include "wl_depth_dev.h"

// the pyramid buffer of a field exactly as wl_heightfield_build_pyramid lays it out (header, entries, copy of the heights)
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
// -> number of floats; fills `out` (capacity `cap` floats) when it is large enough
long long hs_pyramid(const WlHeightField* hf, float* out, long long cap) {
    const std::vector<float> buf = host_pyramid(hf);
    if (out && cap >= (long long)buf.size()) std::copy(buf.begin(), buf.end(), out);
    return (long long)buf.size();
}
// pos [n][3], quat [n][4]; depth [n][60][80]
int hs_depth(const WlVisualParams* p, const WlHeightField* hf, int n, const float* pos, const float* quat, float max_depth, float* depth) {
    const Pyramid py = make_pyramid(hf->nx, hf->ny);
    const std::vector<float> buf = host_pyramid(hf);
    const DepthGrid g = make_depth_grid(hf);
    const FieldMem mem{buf.data()};
    const PyrHead hd = pyramid_head(g, py, mem);
    for (int e = 0; e < n; ++e) {
        const Quat q{quat[4 * e], quat[4 * e + 1], quat[4 * e + 2], quat[4 * e + 3]};
        const Mat3 R = mat_from_quat(q);
        const V3 o = v3(pos[3 * e], pos[3 * e + 1], pos[3 * e + 2]) + mul(R, v3(p->cam_pos[0], p->cam_pos[1], p->cam_pos[2]));
        for (int r = 0; r < WL_VIS_IMG_H; ++r)
            for (int c = 0; c < WL_VIS_IMG_W; ++c)
                depth[((size_t)e * WL_VIS_IMG_H + r) * WL_VIS_IMG_W + c] = cast_ray(g, py, hd, mem, o, mul(R, depth_pixel_ray_body(*p, r, c)), max_depth);
    }
    return 0;
}
}
