// wl_depth_dev.h -- the depth ray-cast's device functions: max-pyramid layout + the per-ray walk (see wl_depth.hip).
// A header of its own so that tests/host_sim can compile the walk for the host and hold it against oracle/depth.c.
#pragma once
#include "../../include/wheeledlab_amd.h"
#include "wl_heightfield.h"
#include "wl_math.h"

namespace {

// Max-pyramid layout: level L (1 <= L <= lmax) is a (P >> L) x (P >> L) array, P = the power of two >= the cell count of the
// longer side; levels are stored back to back, so level L starts at (P^2 - (P >> (L - 1))^2) / 3 floats.  Entries
// that cover no cell hold -inf.  Level 0 is not stored: a cell's maximum is the largest of the four corners the
// intersection needs anyway.
struct Pyramid {
    const float* mip;
    int P, lmax;
};
__host__ __device__ inline int pyramid_offset(int P, int L) {   // P <= 16 384: 32-bit arithmetic (a division by 3 per walk step)
    const unsigned a = (unsigned)P * (unsigned)P, q = (unsigned)(P >> (L - 1));
    return (int)((a - q * q) / 3u);
}
inline int pyramid_pow2(int nx, int ny) {
    int P = 2;
    while (P < nx - 1 || P < ny - 1) P <<= 1;
    return P;
}
inline int pyramid_levels(int P) {
    int l = 0;
    while ((P >> l) > 1) ++l;
    return l;
}

// level 1 from the heights: cell (I, J) covers grid cells (2I .. 2I+1, 2J .. 2J+1), i.e. corners (2I .. 2I+2, 2J .. 2J+2)
WL_DEV float pyramid_level1_value(const WlHeightField& f, int I, int J) {
    float m = -INFINITY;
    if (2 * I < f.nx - 1 && 2 * J < f.ny - 1) {
        const int i1 = min(2 * I + 2, f.nx - 1), j1 = min(2 * J + 2, f.ny - 1);
        for (int j = 2 * J; j <= j1; ++j)
            for (int i = 2 * I; i <= i1; ++i) m = fmaxf(m, f.height[(int64_t)j * f.nx + i]);
    }
    return m;
}
// level L >= 2 from level L - 1
WL_DEV float pyramid_reduce_value(const float* mip, int P, int L, int I, int J) {
    const int W = P >> L;
    const float* src = mip + pyramid_offset(P, L - 1) + (2 * J) * (2 * W) + 2 * I;
    return fmaxf(fmaxf(src[0], src[1]), fmaxf(src[2 * W], src[2 * W + 1]));
}

// first t in [ta, tb] at which the ray is on or below the plane z = zp; < 0: none
WL_DEV float plane_hit(float oz, float dz, float zp, float ta, float tb) {
    if (!(tb >= ta)) return -1.f;
    if (fmaf(ta, dz, oz) <= zp) return ta;
    if (dz < 0.f) {
        const float tp = (zp - oz) * rcp(dz);
        if (tp <= tb) return fmaxf(tp, ta);
    }
    return -1.f;
}

#ifndef WL_DEPTH_START_LEVEL
#define WL_DEPTH_START_LEVEL 2
#endif
constexpr int kMaxWalk = 8192;   // safety bound on walk steps (a ray crosses < 2 * 1024 cells; each costs <= 3 visits)

WL_DEV float cast_ray(const HeightFieldGround& g, const Pyramid& py, const V3 o, const V3 d, const float tmax) {
    const WlHeightField& f = g.f;
    const int NX = f.nx - 1, NY = f.ny - 1;   // cells
    const float ou = (o.x - f.x0) * g.inv_cell, ov = (o.y - f.y0) * g.inv_cell;
    const float du = d.x * g.inv_cell, dv = d.y * g.inv_cell;
    const float oz = o.z, dz = d.z;
    const float idu = du != 0.f ? 1.f / du : 0.f, idv = dv != 0.f ? 1.f / dv : 0.f;
    // parameter interval of the ground track inside the grid domain [0, NX] x [0, NY]
    float t_in = -INFINITY, t_out = INFINITY;
    if (du != 0.f) {
        const float a = (0.f - ou) * idu, b = ((float)NX - ou) * idu;
        t_in = fmaxf(t_in, fminf(a, b));
        t_out = fminf(t_out, fmaxf(a, b));
    } else if (ou < 0.f || ou >= (float)NX) {
        t_in = INFINITY;
    }
    if (dv != 0.f) {
        const float a = (0.f - ov) * idv, b = ((float)NY - ov) * idv;
        t_in = fmaxf(t_in, fminf(a, b));
        t_out = fminf(t_out, fmaxf(a, b));
    } else if (ov < 0.f || ov >= (float)NY) {
        t_in = INFINITY;
    }
    if (!(t_in <= t_out) || t_out < 0.f || t_in > tmax) {   // never over the grid within range
        const float t = plane_hit(oz, dz, f.outside_z, 0.f, tmax);
        return t >= 0.f ? t : tmax;
    }
    if (t_in > 0.f) {
        const float t = plane_hit(oz, dz, f.outside_z, 0.f, t_in);
        if (t >= 0.f) return t;
    } else {
        t_in = 0.f;
    }
    const float t_stop = fminf(t_out, tmax);
    float t = t_in;
    int i, j;
    {
        const float u = fmaf(t, du, ou), v = fmaf(t, dv, ov);
        const float fu = floorf(u), fv = floorf(v);
        i = (int)fu - ((fu == u && du < 0.f) ? 1 : 0);
        j = (int)fv - ((fv == v && dv < 0.f) ? 1 : 0);
        i = min(max(i, 0), NX - 1);
        j = min(max(j, 0), NY - 1);
    }
    const bool up_u = du > 0.f, up_v = dv > 0.f;
    int L = min(WL_DEPTH_START_LEVEL, py.lmax);
    float res = -1.f;
#pragma unroll 1
    for (int it = 0; it < kMaxWalk; ++it) {
        const int iL = i >> L, jL = j >> L;
        const int bx = up_u ? (iL + 1) << L : iL << L, by = up_v ? (jL + 1) << L : jL << L;
        const float tx = du != 0.f ? ((float)bx - ou) * idu : INFINITY;
        const float ty = dv != 0.f ? ((float)by - ov) * idv : INFINITY;
        const float te = fmaxf(fminf(fminf(tx, ty), t_stop), t);
        const float z_t = fmaf(t, dz, oz);
        const float zmin = dz < 0.f ? fmaf(te, dz, oz) : z_t;
        bool advance;
        if (L > 0) {
            const float m = py.mip[pyramid_offset(py.P, L) + jL * (py.P >> L) + iL];
            advance = zmin > m + 1e-6f;
            if (!advance) {
                --L;
                continue;
            }
        } else {
            const float* r0 = f.height + (int64_t)j * f.nx + i;
            const wl_float2_u lo = *reinterpret_cast<const wl_float2_u*>(r0);
            const wl_float2_u hi = *reinterpret_cast<const wl_float2_u*>(r0 + f.nx);
            const float m = fmaxf(fmaxf(lo.x, lo.y), fmaxf(hi.x, hi.y));
            if (!(zmin > m + 1e-6f)) {
                const float hx = lo.y - lo.x, hy = hi.x - lo.x, hxy = (hi.y - lo.y) - hy;
                const float fu = clampf(fmaf(t, du, ou) - (float)i, 0.f, 1.f), fv = clampf(fmaf(t, dv, ov) - (float)j, 0.f, 1.f);
                const float C = z_t - fmaf(fu * fv, hxy, fmaf(fv, hy, fmaf(fu, hx, lo.x)));
                if (C <= 0.f) {
                    res = t;
                    break;
                }
                const float A = -du * dv * hxy;
                const float B = dz - fmaf(fmaf(fu, dv, fv * du), hxy, fmaf(dv, hy, du * hx));
                const float disc = fmaf(B, B, -4.f * A * C);
                if (disc >= 0.f) {
                    const float sq = fsqrt(disc);
                    const float q = -0.5f * (B + copysignf(sq, B));
                    float s = INFINITY;
                    const float r1 = q / A, r2 = C / q;   // A == 0 / q == 0: inf or NaN, neither passes the tests below
                    if (r1 > 0.f && r1 < s) s = r1;
                    if (r2 > 0.f && r2 < s) s = r2;
                    if (s <= te - t) {
                        res = t + s;
                        break;
                    }
                }
            }
            advance = true;
        }
        // leave the level-L cell through its nearer boundary
        if (te >= t_stop) break;
        t = te;
        const int lo_i = iL << L, lo_j = jL << L, span = (1 << L) - 1;
        bool new_parent;
        if (tx <= ty) {
            i = up_u ? bx : bx - 1;
            j = min(max((int)floorf(fmaf(t, dv, ov)), lo_j), min(lo_j + span, NY - 1));
            new_parent = (i >> (L + 1)) != (iL >> 1);
            if (i < 0 || i >= NX) break;
        } else {
            j = up_v ? by : by - 1;
            i = min(max((int)floorf(fmaf(t, du, ou)), lo_i), min(lo_i + span, NX - 1));
            new_parent = (j >> (L + 1)) != (jL >> 1);
            if (j < 0 || j >= NY) break;
        }
        if (new_parent && L < py.lmax) ++L;
    }
    if (res >= 0.f) return fminf(res, tmax);
    if (t_out < tmax) {
        const float th = plane_hit(oz, dz, f.outside_z, t_out, tmax);
        if (th >= 0.f) return th;
    }
    return tmax;
}

// camera ray of pixel (row, col) of the FULL 60 x 80 image in the body frame: optical axis = body +x, image right = body -y,
// image down = body -z (the visual camera's convention, wl_visual.hip).  Body x component 1: the ray parameter IS the
// image-plane distance.
WL_DEV V3 depth_pixel_ray_body(const WlVisualParams& p, int row, int col) {
    return v3(1.f, -(((float)col + 0.5f - p.cx) / p.fx), -(((float)row + 0.5f - p.cy) / p.fy));
}

}  // namespace
