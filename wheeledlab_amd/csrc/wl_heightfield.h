// wl_heightfield.h -- regular-grid heightfield terrain: bilinear height + normal (spec: oracle/heightfield.py::sample)
#pragma once
#include "../../include/wheeledlab_amd.h"
#include "wl_math.h"

namespace {

// bilinear heightfield sampler (spec: oracle/heightfield.py::sample)
struct HeightFieldGround {
    WlHeightField f;
    float inv_cell;
    WL_DEV bool sample_full(float x, float y, float& z, V3& n) const {
        const float u = (x - f.x0) * inv_cell, v = (y - f.y0) * inv_cell;
        const bool inside = u >= 0.f && v >= 0.f && u < (float)(f.nx - 1) && v < (float)(f.ny - 1);
        const float uc = fminf(fmaxf(u, 0.f), (float)(f.nx - 1) - 1e-3f), vc = fminf(fmaxf(v, 0.f), (float)(f.ny - 1) - 1e-3f);
        const float fi = floorf(uc), fj = floorf(vc);
        const int i = (int)fi, j = (int)fj;
        const float fu = uc - fi, fv = vc - fj;
        const float* row0 = f.height + (int64_t)j * f.nx + i;
        const float h00 = row0[0], h10 = row0[1], h01 = row0[f.nx], h11 = row0[f.nx + 1];
        const float a = fmaf(fu, h10 - h00, h00), b = fmaf(fu, h11 - h01, h01);
        const float zz = fmaf(fv, b - a, a);
        const float dzdx = fmaf(fv, (h11 - h01) - (h10 - h00), h10 - h00) * inv_cell;
        const float dzdy = (b - a) * inv_cell;
        const float inv_len = rsq(fmaf(dzdx, dzdx, fmaf(dzdy, dzdy, 1.f)));
        z = inside ? zz : f.outside_z;
        n = inside ? v3(-dzdx * inv_len, -dzdy * inv_len, inv_len) : v3(0.f, 0.f, 1.f);
        return inside;
    }
    WL_DEV void sample(float x, float y, float& z, V3& n) const { (void)sample_full(x, y, z, n); }
    // height only (ray casting: no normal needed) -- same arithmetic as sample_full for z
    WL_DEV bool sample_height(float x, float y, float& z) const {
        const float u = (x - f.x0) * inv_cell, v = (y - f.y0) * inv_cell;
        const bool inside = u >= 0.f && v >= 0.f && u < (float)(f.nx - 1) && v < (float)(f.ny - 1);
        const float uc = fminf(fmaxf(u, 0.f), (float)(f.nx - 1) - 1e-3f), vc = fminf(fmaxf(v, 0.f), (float)(f.ny - 1) - 1e-3f);
        const float fi = floorf(uc), fj = floorf(vc);
        const float fu = uc - fi, fv = vc - fj;
        const float* row0 = f.height + (int64_t)(int)fj * f.nx + (int)fi;
        const float h00 = row0[0], h10 = row0[1], h01 = row0[f.nx], h11 = row0[f.nx + 1];
        const float a = fmaf(fu, h10 - h00, h00), b = fmaf(fu, h11 - h01, h01);
        z = inside ? fmaf(fv, b - a, a) : f.outside_z;
        return inside;
    }
};

inline HeightFieldGround make_ground(const WlHeightField* hf) { return HeightFieldGround{*hf, 1.f / hf->cell}; }

}  // namespace
