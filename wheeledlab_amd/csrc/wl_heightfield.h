// wl_heightfield.h -- regular-grid heightfield terrain: bilinear height + normal (spec: oracle/heightfield.py::sample)
#pragma once
#include "../../include/wheeledlab_amd.h"
#include "wl_math.h"

namespace {

// (Round 3: a pair layout -- every grid point stored next to its +y neighbour, so that a cell's four corners are ONE 16-byte
// gather -- was measured: elevation step at 4096 envs 29.2 vs 26.2 us (the 5.1 MB table no longer fits an XCD's 4 MB L2), at
// 262 144 envs 668 vs 685, at 1 M envs 2572 vs 2503: no gain where nothing fits, a loss where the plain field does.  Reverted.)
// two horizontally adjacent cells as ONE 8-byte gather (the address is only 4-byte aligned: fine for global loads on
// gfx9+); halves the number of gather instructions per bilinear sample
typedef float wl_float2_u __attribute__((ext_vector_type(2), aligned(4)));

// bilinear heightfield sampler (spec: oracle/heightfield.py::sample)
struct HeightFieldGround {
    static constexpr bool kFlat = false;
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
    template <int W>
    WL_DEV void sample_wheel(float x, float y, float& z, V3& n) const { (void)sample_full(x, y, z, n); }
    // split form of sample_height for software pipelining: `corners` issues the two 8-byte gathers, `blend` consumes them
    struct Corners {
        wl_float2_u lo, hi;
        float fu, fv;
        bool inside;
    };
    // the cell a point falls into and its position inside it (the arithmetic every sampler of this file shares)
    struct CellOf {
        int i, j;
        float fu, fv;
        bool inside;
    };
    WL_DEV CellOf cell_of(float x, float y) const {
        CellOf c;
        const float u = (x - f.x0) * inv_cell, v = (y - f.y0) * inv_cell;
        c.inside = u >= 0.f && v >= 0.f && u < (float)(f.nx - 1) && v < (float)(f.ny - 1);
        const float uc = fminf(fmaxf(u, 0.f), (float)(f.nx - 1) - 1e-3f), vc = fminf(fmaxf(v, 0.f), (float)(f.ny - 1) - 1e-3f);
        const float fi = floorf(uc), fj = floorf(vc);
        c.fu = uc - fi;
        c.fv = vc - fj;
        c.i = (int)fi;
        c.j = (int)fj;
        return c;
    }
    WL_DEV Corners corners(float x, float y) const {
        const CellOf k = cell_of(x, y);
        Corners c;
        c.inside = k.inside;
        c.fu = k.fu;
        c.fv = k.fv;
        const float* row0 = f.height + k.j * f.nx + k.i;
        c.lo = *reinterpret_cast<const wl_float2_u*>(row0);
        c.hi = *reinterpret_cast<const wl_float2_u*>(row0 + f.nx);
        return c;
    }
    WL_DEV float blend(const Corners& c) const {
        const float a = fmaf(c.fu, c.lo.y - c.lo.x, c.lo.x), b = fmaf(c.fu, c.hi.y - c.hi.x, c.hi.x);
        return c.inside ? fmaf(c.fv, b - a, a) : f.outside_z;
    }
    // height only (ray casting: no normal needed) -- same arithmetic as sample_full for z
    WL_DEV bool sample_height(float x, float y, float& z) const {
        const float u = (x - f.x0) * inv_cell, v = (y - f.y0) * inv_cell;
        const bool inside = u >= 0.f && v >= 0.f && u < (float)(f.nx - 1) && v < (float)(f.ny - 1);
        const float uc = fminf(fmaxf(u, 0.f), (float)(f.nx - 1) - 1e-3f), vc = fminf(fmaxf(v, 0.f), (float)(f.ny - 1) - 1e-3f);
        const float fi = floorf(uc), fj = floorf(vc);
        const float fu = uc - fi, fv = vc - fj;
        const float* row0 = f.height + (int64_t)(int)fj * f.nx + (int)fi;
        const wl_float2_u lo = *reinterpret_cast<const wl_float2_u*>(row0);
        const wl_float2_u hi = *reinterpret_cast<const wl_float2_u*>(row0 + f.nx);
        const float a = fmaf(fu, lo.y - lo.x, lo.x), b = fmaf(fu, hi.y - hi.x, hi.x);
        z = inside ? fmaf(fv, b - a, a) : f.outside_z;
        return inside;
    }
};

// The same sampler with the four corner heights of each WHEEL's current cell kept in registers (lane form of the step kernels: one
// lane = one env = four wheels).  A wheel moves <= 1.5 cm per 5 ms sub-step over 5 cm cells: its cell changes every few sub-steps,
// so the two 8-byte gathers are re-issued only for the lanes whose wheel crossed a cell line -- a quarter of the lane addresses.
// At large batches the lane-form step is bound by exactly those addresses (160 divergent gather instructions per env-step, each
// 64 envs = 64 unrelated cache lines).  Same arithmetic on the same corners: bit-identical to HeightFieldGround.
struct HeightFieldGroundCached {
    static constexpr bool kFlat = false;
    HeightFieldGround g;
    mutable int cell[4];
    mutable float h00[4], h10[4], h01[4], h11[4];
    WL_DEV explicit HeightFieldGroundCached(const HeightFieldGround& g_) : g(g_) {
#pragma unroll
        for (int w = 0; w < 4; ++w) cell[w] = -1, h00[w] = h10[w] = h01[w] = h11[w] = 0.f;
    }
    template <int W>
    WL_DEV void sample_wheel(float x, float y, float& z, V3& n) const {
        const WlHeightField& f = g.f;
        const float u = (x - f.x0) * g.inv_cell, v = (y - f.y0) * g.inv_cell;
        const bool inside = u >= 0.f && v >= 0.f && u < (float)(f.nx - 1) && v < (float)(f.ny - 1);
        const float uc = fminf(fmaxf(u, 0.f), (float)(f.nx - 1) - 1e-3f), vc = fminf(fmaxf(v, 0.f), (float)(f.ny - 1) - 1e-3f);
        const float fi = floorf(uc), fj = floorf(vc);
        const int k = (int)fj * f.nx + (int)fi;
        const float fu = uc - fi, fv = vc - fj;
        if (k != cell[W]) {       // per lane: only the lanes whose wheel changed cells gather
            const float* row0 = f.height + k;
            const wl_float2_u lo = *reinterpret_cast<const wl_float2_u*>(row0);
            const wl_float2_u hi = *reinterpret_cast<const wl_float2_u*>(row0 + f.nx);
            h00[W] = lo.x, h10[W] = lo.y, h01[W] = hi.x, h11[W] = hi.y;
            cell[W] = k;
        }
        const float a00 = h00[W], a10 = h10[W], a01 = h01[W], a11 = h11[W];
        const float a = fmaf(fu, a10 - a00, a00), b = fmaf(fu, a11 - a01, a01);
        const float zz = fmaf(fv, b - a, a);
        const float dzdx = fmaf(fv, (a11 - a01) - (a10 - a00), a10 - a00) * g.inv_cell;
        const float dzdy = (b - a) * g.inv_cell;
        const float inv_len = rsq(fmaf(dzdx, dzdx, fmaf(dzdy, dzdy, 1.f)));
        z = inside ? zz : f.outside_z;
        n = inside ? v3(-dzdx * inv_len, -dzdy * inv_len, inv_len) : v3(0.f, 0.f, 1.f);
    }
};

inline HeightFieldGround make_ground(const WlHeightField* hf) { return HeightFieldGround{*hf, 1.f / hf->cell}; }

}  // namespace
