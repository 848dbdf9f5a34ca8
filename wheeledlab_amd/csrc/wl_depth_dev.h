// wl_depth_dev.h -- the depth ray-cast's device functions: max-pyramid layout + the per-ray walk (see wl_depth.hip).
// A header of its own so that tests/host_sim can compile the walk for the host and hold it against oracle/depth.c.
#pragma once
#include "../../include/wheeledlab_amd.h"
#include "wl_math.h"
#include "wl_heightfield.h"      // hf_at / hf_decode_pair: the one decoding of the 16-bit height codes

namespace {

// ONE device buffer holds everything the walk gathers, so that every access is a 32-bit offset from one base:
//   float  [0]                       the field's maximum (also in the header)
//   words  [PP >> 2L, 2 (PP >> 2L))  level L of the BOUND pyramid, 1 <= L <= lp: a (P >> L) x (P >> L) array (P = 2^lp = the
//                                    power of two >= the cell count of the longer side, PP = P^2) of 4-byte entries
//                                    { uint16 c' | int8 a' | int8 b' }: inside the 2^L x 2^L block of cells (J, I) the terrain stays
//                                    below the PLANE  a (u - I 2^L) + b (v - J 2^L) + c  (u, v in grid units) with a = a' qs,
//                                    b = b' qs, c = c0 + c' qc; blocks that cover no cell: the word 0 (never read by the walk)
//   words  [h0, h0 + ceil(ny nx / 2)) a copy of the 16-bit height CODES [ny][nx] (round 5; fp32 heights before: 2.56 MB of the bench
//                                    field's 3.4 MB), h0 = max(PP / 2, 4) (level 0 is not stored: the intersection needs the
//                                    four corners of a cell anyway -- two 4-byte gathers of two codes each)
//   floats [hdr, hdr + 4)            the header: field maximum, slope quantum qs, offset base c0, offset quantum qc
// The level offsets are shifts (no table, no division); the price is PP / 6 floats of padding.
//
// Round 4: bounding PLANES instead of the round-3 maxima.  Under a maximum a cell on a slope is "as high as its highest corner":
// a ray skimming 10 - 20 cm above a hillside cannot skip cells larger than ~0.2 m and walks them one by one.  A plane fitted to
// the cell (slopes from its corner heights, quantised; c = the largest residual of ANY grid point of the cell against the
// QUANTISED slopes, rounded UP, so the bound is exact whatever the fit is worth) is loose only by the cell's curvature; the ray
// clears it over its whole stay in the cell iff it clears it at both ends (ray and plane are both linear).  Same answers bit for
// bit (a skip is only ever conservative); host simulation on the bench poses: 8.15 -> 6.49 steps per ray, 13.9 -> 9.1 wave-steps
// per tile.  Where the flat bound (a = b = 0, c = the cell's maximum) is the lower one at the cell's centre it is stored instead.
// Entries of FOUR bytes (first version: fp16 slopes + float c = eight): the 800 x 800 bench field's entries + heights are 3.4 MB
// instead of 4.3 MB -- under the 4 MB of L2 an XCD has.  What the quantisation costs the bound: int8 slopes with ONE quantum per
// field (its steepest cell edge / 127) are off by at most qs / 2 per cell of the block's extent (bench field: 0.1 mm per cell);
// 16-bit fixed-point offsets over twice the field's relief are off by at most qc (bench field: 0.03 mm -- fp16 offsets, 0.25 - 0.5 mm
// there, cost 2 % more wave-steps: the rays that matter pass the last cells before their hit within a millimetre of the surface).
struct Pyramid {
    int lp;        // log2 P = the top level (one entry)
    int h0;        // word (4-byte) offset of the copy of the height codes
    int hdr;       // word offset of the header
};
constexpr int kPyrMax = 0, kPyrSlopeQ = 1, kPyrBase = 2, kPyrOffsetQ = 3, kPyrHeader = 4;   // header floats
inline int pyramid_log2(int nx, int ny) {
    int lp = 1;
    while ((1 << lp) < nx - 1 || (1 << lp) < ny - 1) ++lp;
    return lp;
}
__host__ __device__ inline int pyramid_level_offset(int lp, int L) { return (1 << (2 * lp)) >> (2 * L); }   // in ENTRIES = words
inline Pyramid make_pyramid(int nx, int ny) {
    const int lp = pyramid_log2(nx, ny);
    const int h0 = max((1 << (2 * lp)) >> 1, 4);
    return Pyramid{lp, h0, h0 + (nx * ny + 1) / 2};
}
inline int64_t pyramid_total_floats(int nx, int ny) { return (int64_t)make_pyramid(nx, ny).hdr + kPyrHeader; }

// what every launch reads once from the header
struct PyrHead {
    float zclear;       // the highest height of the field or the outside plane, whichever is higher (rising rays end above it)
    float qs, c0, qc;   // slope quantum, offset base, offset quantum
};
// the header of a field with heights in [hmin, hmax] (finite: `finite`) whose steepest cell edge rises smax per cell.  Offsets cover
// [hmin, hmin + 2 relief]: every block maximum (the flat bound) fits, a fitted plane whose corner offset does not falls back to it.
// A field with a non-finite height gets qc = +inf: every bound is +inf, nothing is ever skipped, the walk visits every cell.
__host__ __device__ inline void pyramid_header_values(float hmin, float hmax, float smax, bool finite, float* hd) {
    hd[kPyrMax] = hmax;
    hd[kPyrSlopeQ] = (finite && smax > 0.f) ? smax * (1.f / 127.f) : 1.f;
    const float span = fmaxf(2.f * (hmax - hmin), 1e-3f * (1.f + fabsf(hmax)));
    hd[kPyrBase] = finite ? hmin : 0.f;
    hd[kPyrOffsetQ] = finite ? span * (1.f / 65534.f) : INFINITY;
}
// a slope as a multiple of the quantum
WL_DEV int slope_steps(float s, float q) {
    const float r = rintf(s / q);
    return (r == r) ? (int)fminf(fmaxf(r, -127.f), 127.f) : 0;      // NaN heights: flat
}
// the smallest offset code whose value c0 + code qc is >= c (may exceed 65535: the caller falls back to the flat bound)
WL_DEV int offset_code(float c, float c0, float qc) {
    if (!(qc < 1e30f)) return 1;                        // qc = +inf: any non-zero code decodes to +inf
    float u = ceilf((c - c0) / qc);
    if (!(u == u) || u > 1e6f) return 1 << 20;
    u = fmaxf(u, 1.f);
    if (fmaf(u, qc, c0) < c) u += 1.f;                  // the division rounded down across an integer
    return (int)u;
}
WL_DEV uint32_t pack_entry(int a8, int b8, int code) { return ((uint32_t)code << 16) | ((uint32_t)(a8 & 0xff) << 8) | (uint32_t)(b8 & 0xff); }
WL_DEV void unpack_entry(uint32_t w, const PyrHead& hd, float& a, float& b, float& c) {
    a = (float)(int)(int8_t)(w >> 8) * hd.qs;
    b = (float)(int)(int8_t)w * hd.qs;
    c = fmaf((float)(w >> 16), hd.qc, hd.c0);
}
// the grid points of cell (I, J) of level L: [i0, i1] x [j0, j1] (clipped to the field); false: the block covers no cell
WL_DEV bool plane_cell_range(const WlHeightField& f, int L, int I, int J, int& i0, int& i1, int& j0, int& j1) {
    i0 = I << L, j0 = J << L;
    if (i0 >= f.nx - 1 || j0 >= f.ny - 1) return false;
    i1 = min(i0 + (1 << L), f.nx - 1), j1 = min(j0 + (1 << L), f.ny - 1);
    return true;
}
// the fitted slopes of a cell: mean slope between its opposite edges' corner heights, as the entry will hold them (whole steps of the
// quantum: any slopes give a valid bound, the residual below makes it exact)
WL_DEV void plane_cell_slopes(const WlHeightField& f, int i0, int i1, int j0, int j1, float q, int& a8, int& b8, float& a, float& b) {
    const float h00 = hf_at(f, (int64_t)j0 * f.nx + i0), h10 = hf_at(f, (int64_t)j0 * f.nx + i1);
    const float h01 = hf_at(f, (int64_t)j1 * f.nx + i0), h11 = hf_at(f, (int64_t)j1 * f.nx + i1);
    const float fa = ((h10 + h11) - (h00 + h01)) / (2.f * (float)(i1 - i0)), fb = ((h01 + h11) - (h00 + h10)) / (2.f * (float)(j1 - j0));
    a8 = slope_steps(fa, q), b8 = slope_steps(fb, q);
    a = (float)a8 * q, b = (float)b8 * q;        // exactly what unpack_entry returns
}
// residual and height of ONE grid point against the cell's slopes (the reductions over a cell's points take the max of both)
WL_DEV void plane_point(const WlHeightField& f, int i0, int j0, int i, int j, float a, float b, float& resid, float& hmax) {
    const float h = hf_at(f, (int64_t)j * f.nx + i);
    resid = fmaxf(resid, h - fmaf(a, (float)(i - i0), b * (float)(j - j0)));
    hmax = fmaxf(hmax, h);
}
// the entry to store: the fitted plane, or the flat bound where that is the lower one at the cell's centre; a hair of slack
// covers the rounding of the walk's own evaluation of the plane (and of base + c')
WL_DEV uint32_t plane_entry(int i0, int i1, int j0, int j1, int a8, int b8, float a, float b, float resid, float hmax, const float* hd) {
    const float centre = resid + 0.5f * fmaf(a, (float)(i1 - i0), b * (float)(j1 - j0));
    const float c0 = hd[kPyrBase], qc = hd[kPyrOffsetQ];
    // + the rounding of the walk's t g1 term: |t du| is the distance travelled from the ray's origin, at most |u0| + the block's
    // extent -- the walk's own margin covers the |a u0| + |b v0| part, the extent part is paid here once per entry instead of per
    // walk step (a ray that starts at a block's corner and crosses a large steep block: 1.2e-7 |a| 2^L, beyond the walk's fixed
    // 1e-5 from |a| ~ 0.7 m per cell at level 7)
    const float travel = 4e-7f * fmaf(fabsf(a), (float)(i1 - i0), fabsf(b) * (float)(j1 - j0));
    const int fit = offset_code(resid + 2e-6f * (1.f + fabsf(resid)) + travel, c0, qc);
    if (centre < hmax && fit <= 65535) return pack_entry(a8, b8, fit);
    return pack_entry(0, 0, min(offset_code(hmax + 2e-6f * (1.f + fabsf(hmax)), c0, qc), 65535));
}
constexpr uint32_t kEmptyEntry = 0u;     // the block covers no cell (the walk never asks for it)
// one cell, serially (small cells on the device; every cell in the host simulation)
WL_DEV uint32_t plane_cell_serial(const WlHeightField& f, int L, int I, int J, const float* hd) {
    int i0, i1, j0, j1;
    if (!plane_cell_range(f, L, I, J, i0, i1, j0, j1)) return kEmptyEntry;
    int a8, b8;
    float a, b, resid = -INFINITY, hmax = -INFINITY;
    plane_cell_slopes(f, i0, i1, j0, j1, hd[kPyrSlopeQ], a8, b8, a, b);
    for (int j = j0; j <= j1; ++j)
        for (int i = i0; i <= i1; ++i) plane_point(f, i0, j0, i, j, a, b, resid, hmax);
    return plane_entry(i0, i1, j0, j1, a8, b8, a, b, resid, hmax, hd);
}
// one height's share of the header's reductions
WL_DEV void header_point(const WlHeightField& f, int i, int j, float& hmin, float& hmax, float& smax, bool& finite) {
    const float h = hf_at(f, (int64_t)j * f.nx + i);
    hmin = fminf(hmin, h), hmax = fmaxf(hmax, h);
    finite = finite && (h - h == 0.f);      // codes are finite; a non-finite z_scale is refused at the ABI -- kept for the bound's contract
    if (i + 1 < f.nx) smax = fmaxf(smax, fabsf(hf_at(f, (int64_t)j * f.nx + i + 1) - h));
    if (j + 1 < f.ny) smax = fmaxf(smax, fabsf(hf_at(f, (int64_t)(j + 1) * f.nx + i) - h));
}
// the header of a field, serially (the host simulation; the device has pyramid_header_kernel)
#ifdef WL_HOST_SIM
inline void pyramid_header_serial(const WlHeightField& f, float* hd) {
    float hmin = INFINITY, hmax = -INFINITY, smax = 0.f;
    bool finite = true;
    for (int j = 0; j < f.ny; ++j)
        for (int i = 0; i < f.nx; ++i) header_point(f, i, j, hmin, hmax, smax, finite);
    pyramid_header_values(hmin, hmax, smax, finite, hd);
}
#endif

// the walk's view of that buffer: 4-byte gathers (an entry, or two height codes) at word / halfword offsets
struct FieldMem {
#ifdef WL_HOST_SIM
    const float* base;
    WL_DEV float ld(int idx) const { return base[idx]; }
    WL_DEV uint32_t ldw(int idx) const { return __builtin_bit_cast(uint32_t, base[idx]); }
    // two adjacent height codes at HALFWORD index hidx of the buffer, decoded
    WL_DEV void ldh2(int hidx, float zs, float& a, float& b) const {
        const int16_t* c = reinterpret_cast<const int16_t*>(base) + hidx;
        hf_decode_pair((uint32_t)(uint16_t)c[0] | ((uint32_t)(uint16_t)c[1] << 16), zs, a, b);
    }
#else
    __amdgpu_buffer_rsrc_t rsrc;   // buffer loads: ONE 32-bit VGPR offset per gather, no 64-bit address arithmetic
    WL_DEV float ld(int idx) const { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, idx * 4, 0, 0)); }
    WL_DEV uint32_t ldw(int idx) const { return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, idx * 4, 0, 0); }
    // two adjacent height codes at HALFWORD index hidx of the buffer (one 4-byte gather, 2-byte aligned), decoded
    WL_DEV void ldh2(int hidx, float zs, float& a, float& b) const {
        hf_decode_pair((uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, hidx * 2, 0, 0), zs, a, b);
    }
#endif
};

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

// where a ray enters the pyramid: rays that rise can only meet terrain above the camera and start high (a few large cells and
// out); rays that fall start near the level at which the ground in front of the car stops being skippable.  Round 3 (block maxima),
// same-box A/B at 4096 cameras (us per render at 100 / 20 / 5 m range): 4 / 4: 694 / 665 / 566, 3 / 7: 694 / 673 / 565, 2 / 6:
// 685 / 683 / 563, 4 / 8: 712 / 689 / 577 -- flat.  Round 4 (bounding planes skip further): host simulation on the bench poses,
// wave-steps per tile: 3 / 7: 9.48, 4 / 6: 9.25, 4 / 5: 9.44, 5 / 6: 9.36; on the device 3 / 7: 374.8, 4 / 6: 369.8 us per render.
#ifndef WL_DEPTH_STEP_HOOK
#define WL_DEPTH_STEP_HOOK(L)       // host instrumentation (walk steps per ray): nothing in the device build
#endif
#ifndef WL_DEPTH_CLEAR_HOOK
#define WL_DEPTH_CLEAR_HOOK(clear, t, te)   // host instrumentation: nothing in the device build
#endif
#ifndef WL_DEPTH_START_LEVEL
#define WL_DEPTH_START_LEVEL 4      // falling rays
#endif
#ifndef WL_DEPTH_START_LEVEL_UP
#define WL_DEPTH_START_LEVEL_UP 6   // rising rays
#endif

// the grid the walk runs on: cells, not metres (u = (x - x0) / cell, integer cell lines)
struct DepthGrid {
    int nx, NX, NY;            // row pitch of the heights; cells per side
    float x0, y0, inv_cell, outside_z;
    float zs;                  // metres per height code
};
inline DepthGrid make_depth_grid(const WlHeightField* hf) {
    return DepthGrid{hf->nx, hf->nx - 1, hf->ny - 1, hf->x0, hf->y0, 1.f / hf->cell, hf->outside_z, hf->z_scale};
}

// The walk of ONE ray as a state machine, so that a wavefront can keep its lanes busy: ray_begin() sets a ray up (or answers it
// at once: never over the grid, underground, ...), ray_step() advances it by one pyramid cell, ray_result() closes it.  The
// kernel (wl_depth.hip) hands a lane the next ray of its pool the moment the lane's ray is done.  Spec: oracle/depth.c::cast_ray.
struct RayWalk {
    // constants of the ray
    float ou, ov, du, dv, idu, idv, oz, dz;
    float t_stop, t_out, tmax;
    int su, sv;
    // the walk
    float t, res;
    int i, j, L;
    bool live;      // still walking (false: `res` is final, or the walk left the grid / the range: see ray_result)
    bool cleared;   // t_stop is the parameter beyond which a RISING ray is above every height of the field and the outside plane
};

// Distance along the optical axis (|d_body.x| = 1) to the first point of the ray o + t d on or below the terrain solid,
// clipped at tmax: set-up.  `zclear`: the highest height of the field (the pyramid's top entry) or the outside plane, whichever
// is higher -- a rising ray that has climbed past it can meet nothing any more, its walk ends there instead of at the grid's edge
// (round 4: sky rays 6 -> 2.5 steps, every ray 10.3 -> 8.6 on the bench poses; same results bit for bit).
WL_DEV RayWalk ray_begin(const DepthGrid& g, const Pyramid& py, const float zclear, const V3 o, const V3 d, const float tmax) {
    const int NX = g.NX, NY = g.NY;
    RayWalk w;
    w.ou = (o.x - g.x0) * g.inv_cell, w.ov = (o.y - g.y0) * g.inv_cell;
    w.du = d.x * g.inv_cell, w.dv = d.y * g.inv_cell;
    w.oz = o.z, w.dz = d.z;
    w.tmax = tmax;
    w.live = false;
    w.cleared = false;
    w.res = -1.f;
    w.t = 0.f, w.i = w.j = 0, w.L = 0, w.t_stop = 0.f;
    const float ou = w.ou, ov = w.ov, du = w.du, dv = w.dv, oz = w.oz, dz = w.dz;
    // a direction component of exactly zero never reaches a cell line: the huge reciprocal with `up` set sends that line's
    // parameter to +1e30 (the line above the entry cell is strictly above the entry point)
    const bool up_u = du >= 0.f, up_v = dv >= 0.f;
    w.idu = du != 0.f ? rcp(du) : 1e30f, w.idv = dv != 0.f ? rcp(dv) : 1e30f;
    w.su = up_u ? 1 : 0, w.sv = up_v ? 1 : 0;
    // parameter interval of the ground track inside the grid domain [0, NX] x [0, NY]
    float t_in = -INFINITY, t_out = INFINITY;
    if (du != 0.f) {
        const float a = (0.f - ou) * w.idu, b = ((float)NX - ou) * w.idu;
        t_in = fmaxf(t_in, fminf(a, b));
        t_out = fminf(t_out, fmaxf(a, b));
    } else if (ou < 0.f || ou >= (float)NX) {
        t_in = INFINITY;
    }
    if (dv != 0.f) {
        const float a = (0.f - ov) * w.idv, b = ((float)NY - ov) * w.idv;
        t_in = fmaxf(t_in, fminf(a, b));
        t_out = fminf(t_out, fmaxf(a, b));
    } else if (ov < 0.f || ov >= (float)NY) {
        t_in = INFINITY;
    }
    w.t_out = t_out;
    if (!(t_in <= t_out) || t_out < 0.f || t_in > tmax) {   // never over the grid within range
        const float t = plane_hit(oz, dz, g.outside_z, 0.f, tmax);
        w.res = t >= 0.f ? t : tmax;
        return w;
    }
    if (t_in > 0.f) {
        const float t = plane_hit(oz, dz, g.outside_z, 0.f, t_in);
        if (t >= 0.f) {
            w.res = t;
            return w;
        }
    } else {
        t_in = 0.f;
    }
    w.t_stop = fminf(t_out, tmax);
    if (dz > 0.f) {     // rising: above everything from t_clear on
        const float t_clear = (zclear + 1e-5f - oz) * rcp(dz);
        if (t_clear < w.t_stop) {
            w.t_stop = t_clear;
            w.cleared = true;
            if (t_clear <= t_in) {   // already above everything where it enters the grid
                w.res = tmax;
                return w;
            }
        }
    }
    w.t = t_in;
    {
        const float u = fmaf(w.t, du, ou), v = fmaf(w.t, dv, ov);
        const float fu = floorf(u), fv = floorf(v);
        int i = (int)fu - ((fu == u && du < 0.f) ? 1 : 0);
        int j = (int)fv - ((fv == v && dv < 0.f) ? 1 : 0);
        w.i = min(max(i, 0), NX - 1);
        w.j = min(max(j, 0), NY - 1);
    }
    w.L = min(dz >= 0.f ? WL_DEPTH_START_LEVEL_UP : WL_DEPTH_START_LEVEL, py.lp);
    w.live = true;
    return w;
}

// one step of the walk: test the level-L cell of (i, j); hit, descend, or leave it through its nearer line
WL_DEV void ray_step(const DepthGrid& g, const Pyramid& py, const PyrHead& hd, const FieldMem& mem, RayWalk& w) {
    const int NX = g.NX, NY = g.NY;
    const float ou = w.ou, ov = w.ov, du = w.du, dv = w.dv, oz = w.oz, dz = w.dz, idu = w.idu, idv = w.idv, t = w.t, t_stop = w.t_stop;
    const int su = w.su, sv = w.sv, i = w.i, j = w.j, L = w.L;
    WL_DEPTH_STEP_HOOK(L);
    // the level-L cell of (i, j), the parameter at which the ray leaves it, the ray's lowest point inside it
    const int iL = i >> L, jL = j >> L;
    const int bx = (iL + su) << L, by = (jL + sv) << L;
    // (line - origin) / direction -- not fma(line, 1 / d, -origin / d), two instructions shorter: for rays nearly parallel to
    // a grid axis the two products are ~1e5 and the exit parameter would carry their rounding (up to centimetres).  The walk
    // survives that (a hit the shortened interval misses is caught by the next cell's entry test; the parity sets pass either
    // way), but an exit parameter that is exact to rounding is worth two instructions
    const float tx = ((float)bx - ou) * idu, ty = ((float)by - ov) * idv;
    const float te = fmaxf(fminf(fminf(tx, ty), t_stop), t);
    const float z_t = fmaf(t, dz, oz);
    const float zmin = dz < 0.f ? fmaf(te, dz, oz) : z_t;
    const bool fine = L == 0;
    float h00, h10, h01, h11;
    bool clear;      // the ray stays above everything in this cell over [t, te]
    if (fine) {
        const int k = 2 * py.h0 + (int)__umul24((unsigned)j, (unsigned)g.nx) + i;   // halfword index; j, nx < 2^24: the full-rate multiply
        mem.ldh2(k, g.zs, h00, h10);
        mem.ldh2(k + g.nx, g.zs, h01, h11);
        clear = zmin > fmaxf(fmaxf(h00, h10), fmaxf(h01, h11)) + 1e-6f;
    } else {
        // the cell's bounding plane, one 4-byte gather: above it at both ends of the stay = above it throughout
        float a, b, c;
        unpack_entry(mem.ldw(pyramid_level_offset(py.lp, L) + (jL << (py.lp - L)) + iL), hd, a, b, c);
        const float u0 = ou - (float)(iL << L), v0 = ov - (float)(jL << L);
        const float au = a * u0, bv = b * v0;
        const float g0 = (oz - c) - (au + bv), g1 = dz - fmaf(a, du, b * dv);      // height above the plane: g(t) = g0 + t g1
        // margin: the rounding of g itself (the products can be hundreds of times the ray's clearance on steep, far cells)
        clear = fminf(fmaf(t, g1, g0), fmaf(te, g1, g0)) > fmaf(4e-7f, fabsf(au) + fabsf(bv), 1e-5f);
    }
    WL_DEPTH_CLEAR_HOOK(clear, t, te);
    if (!clear) {      // the ray may touch something in this cell
        if (!fine) {
            w.L = L - 1;
            return;
        }
        const float hx = h10 - h00, hy = h01 - h00, hxy = (h11 - h10) - hy;
        const float fu = clampf(fmaf(t, du, ou) - (float)i, 0.f, 1.f), fv = clampf(fmaf(t, dv, ov) - (float)j, 0.f, 1.f);
        const float C = z_t - fmaf(fu * fv, hxy, fmaf(fv, hy, fmaf(fu, hx, h00)));
        if (C <= 0.f) {
            w.res = t;
            w.live = false;
            return;
        }
        const float A = -du * dv * hxy;
        const float B = dz - fmaf(fmaf(fu, dv, fv * du), hxy, fmaf(dv, hy, du * hx));
        const float disc = fmaf(B, B, -4.f * A * C);
        if (disc >= 0.f) {
            const float q = -0.5f * (B + copysignf(fsqrt(disc), B));
            const float r1 = q * rcp(A), r2 = C * rcp(q);   // A == 0 / q == 0: inf or NaN, neither passes the tests below
            float s = INFINITY;
            if (r1 > 0.f && r1 < s) s = r1;
            if (r2 > 0.f && r2 < s) s = r2;
            if (s <= te - t) {
                w.res = t + s;
                w.live = false;
                return;
            }
        }
    }
    // leave the level-L cell through its nearer line; climb when that line is also the parent's
    if (te >= t_stop) {
        w.live = false;
        return;
    }
    w.t = te;
    const bool exit_x = tx <= ty;
    // the coordinate ALONG the line crossed, recomputed from t and kept inside the cell just left (rounding must not move it
    // to a cell the ray has not reached); the coordinate ACROSS it steps by one cell of level L
    const float wl = floorf(fmaf(te, exit_x ? dv : du, exit_x ? ov : ou));
    const int lo = (exit_x ? jL : iL) << L;
    const int c = min(max((int)wl, lo), min(lo + (1 << L) - 1, (exit_x ? NY : NX) - 1));
    const int ni = exit_x ? bx + su - 1 : c, nj = exit_x ? c : by + sv - 1;
    w.i = ni, w.j = nj;
    if ((unsigned)ni >= (unsigned)NX || (unsigned)nj >= (unsigned)NY) {
        w.live = false;
        return;
    }
    const int edge = exit_x ? iL ^ su : jL ^ sv;    // moving up out of an odd cell / down out of an even one: a new parent
    w.L = L + (((edge & 1) == 0 && L < py.lp) ? 1 : 0);
}

// the answer of a ray whose walk has ended
WL_DEV float ray_result(const DepthGrid& g, const RayWalk& w) {
    if (w.res >= 0.f) return fminf(w.res, w.tmax);
    if (!w.cleared && w.t_out < w.tmax) {
        const float th = plane_hit(w.oz, w.dz, g.outside_z, w.t_out, w.tmax);
        if (th >= 0.f) return th;
    }
    return w.tmax;
}

// bound on walk steps: a ground track crosses at most NX + NY cell lines, every cell costs at most a climb, a descent and a
// visit per level change -- generous, and finite whatever rounding does (a GPU must never spin)
WL_DEV int max_walk_steps(const DepthGrid& g) { return 4 * (g.NX + g.NY) + 64; }

// one ray from start to end (the host simulation and the one-ray-per-lane kernel form)
WL_DEV float cast_ray(const DepthGrid& g, const Pyramid& py, const PyrHead& hd, const FieldMem& mem, const V3 o, const V3 d, const float tmax) {
    RayWalk w = ray_begin(g, py, hd.zclear, o, d, tmax);
    const int max_walk = max_walk_steps(g);
#pragma unroll 1
    for (int it = 0; it < max_walk && w.live; ++it) ray_step(g, py, hd, mem, w);
    return ray_result(g, w);
}
// the pyramid's header: the height nothing of the terrain solid rises above (the field's maximum or the outside plane), the quanta and
// the base of the entries
WL_DEV PyrHead pyramid_head(const DepthGrid& g, const Pyramid& py, const FieldMem& mem) {
    return PyrHead{fmaxf(mem.ld(py.hdr + kPyrMax), g.outside_z), mem.ld(py.hdr + kPyrSlopeQ), mem.ld(py.hdr + kPyrBase), mem.ld(py.hdr + kPyrOffsetQ)};
}

// camera ray of pixel (row, col) of the FULL 60 x 80 image in the body frame: optical axis = body +x, image right = body -y,
// image down = body -z (the visual camera's convention, wl_visual.hip).  Body x component 1: the ray parameter IS the
// image-plane distance.
WL_DEV V3 depth_pixel_ray_body(const WlVisualParams& p, int row, int col) {
    return v3(1.f, -(((float)col + 0.5f - p.cx) / p.fx), -(((float)row + 0.5f - p.cy) / p.fy));
}

}  // namespace
