// wl_heightfield.h -- regular-grid heightfield terrain: bilinear height + normal (spec: oracle/heightfield.py::sample)
#pragma once
#include "../../include/wheeledlab_amd.h"
#include "wl_math.h"

namespace {

// (Round 3: a pair layout -- every grid point stored next to its +y neighbour, so that a cell's four corners are ONE 16-byte
// gather -- was measured: elevation step at 4096 envs 29.2 vs 26.2 us (the 5.1 MB table no longer fits an XCD's 4 MB L2), at
// 262 144 envs 668 vs 685, at 1 M envs 2572 vs 2503: no gain where nothing fits, a loss where the plain field does.  Reverted.)
// Round 5: the grid holds 16-bit height CODES (z = code * z_scale, WlHeightField): two horizontally adjacent grid points are ONE
// 4-byte gather (the address is only 2-byte aligned: gfx950 takes unaligned dword loads from global memory, buffers and LDS --
// the compiler emits global_load_dword / ds_read_b32 for them), a cell's four corners two of them; decoding is one conversion
// (SDWA: sign-extended half -> float) and one multiply per corner, the same on every CONTACT and DEPTH path, so that those samplers
// of a field see exactly the floats the oracle's decoded grid holds.  The one exception is the height SCAN (wl_elev.hip::scan_value,
// shared by all four scan forms): it blends the four corner codes and scales once -- identical when z_scale is a power of two
// (terrain.py's default 2^-13; scaling by 2^k commutes with every rounding), different in the last bit otherwise (IsaacLab's
// vertical_scale 0.005): the scan's parity bound is 2e-5 m (tests/test_gpu_elev_parity.py), not bit equality with the decoded grid.
typedef uint32_t wl_u32_u2 __attribute__((aligned(2)));
// a dword of two adjacent codes (low half = the first) -> their heights.  The multiply is kept a multiply (contract off): under
// -ffp-contract=fast the compiler would otherwise fuse it into whichever add consumes the height at each inlining site, and two
// samplers of one field would disagree in the last bit for scales that are not a power of two.
WL_DEV void hf_decode_pair(uint32_t w, float z_scale, float& a, float& b) {
#pragma clang fp contract(off)
    a = (float)(int)(int16_t)(w & 0xffffu) * z_scale;
    b = (float)((int)w >> 16) * z_scale;
}
// (Rounds 4 - 5 read two adjacent codes as ONE 2-byte aligned dword gather from the code field, `hf_pair`, two per cell; the aligned
// 8 bytes around the pair + v_alignbit instead -- what the LDS patch reads need, wl_elev.hip -- was slower from global memory: the
// gather-form scan at 262 144 envs 504 against 482 us per launch, the fused 4096-env step 25.2 against 25.0 us.  The depth walk still
// reads its code pairs that way, from its own copy of the codes: wl_depth_dev.h.)
// a cell's four corner heights from the ROW-PAIR table (WlHeightField.pair, ABI 23: pair[j][i] = code[j][i] | code[j + 1][i] << 16):
// ONE 8-byte gather at 4-byte alignment instead of two 4-byte ones from rows j and j + 1 of the code field -- half the lane addresses
// the texture unit is charged for, and the contact samplers share the height scan's working set in L2 (with the scan on the pair
// table and the contacts on the code field the fused elevation launch's ten sub-steps took 8.56 us instead of 8.08).  Same codes,
// same decode: bit-identical to the two-gather form.
typedef uint32_t wl_u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
WL_DEV void hf_corners(const WlHeightField& f, int64_t k, float& h00, float& h10, float& h01, float& h11) {
    const wl_u32x2_a4 w = *reinterpret_cast<const wl_u32x2_a4*>(f.pair + k);
    hf_decode_pair(w.x, f.z_scale, h00, h01);
    hf_decode_pair(w.y, f.z_scale, h10, h11);
}
// one grid point, decoded (table builders: the depth pyramid)
WL_DEV float hf_at(const WlHeightField& f, int64_t k) {
#pragma clang fp contract(off)
    return (float)(int)f.height[k] * f.z_scale;
}

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
        const int64_t k = (int64_t)j * f.nx + i;
        float h00, h10, h01, h11;
        hf_corners(f, k, h00, h10, h01, h11);
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
    // split form of sample_height for software pipelining: `corners` issues the gather, `blend` consumes it
    struct Corners {
        float h00, h10, h01, h11;
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
        const int idx = k.j * f.nx + k.i;
        hf_corners(f, idx, c.h00, c.h10, c.h01, c.h11);
        return c;
    }
    WL_DEV float blend(const Corners& c) const {
        const float a = fmaf(c.fu, c.h10 - c.h00, c.h00), b = fmaf(c.fu, c.h11 - c.h01, c.h01);
        return c.inside ? fmaf(c.fv, b - a, a) : f.outside_z;
    }
    // height only (ray casting: no normal needed) -- same arithmetic as sample_full for z
    WL_DEV bool sample_height(float x, float y, float& z) const {
        const Corners c = corners(x, y);
        z = blend(c);
        return c.inside;
    }
};

// The same sampler with the four corner heights of each WHEEL's current cell kept in registers (lane form of the step kernels: one
// lane = one env = four wheels).  A wheel moves <= 1.5 cm per 5 ms sub-step over 5 cm cells: its cell changes every few sub-steps,
// so the gather is re-issued only for the lanes whose wheel crossed a cell line -- a quarter of the lane addresses.
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
        if (k != cell[W]) {       // per lane: only the lanes whose wheel changed cells gather (and decode)
            hf_corners(f, k, h00[W], h10[W], h01[W], h11[W]);
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

// (Round 5, quad form: a sampler that requested the NEXT sub-step's codes one sub-step ahead -- a 4 x 3 block of codes around the contact
// point, three 8-byte gathers, the two code pairs picked out of the landed rows with v_cndmask / v_alignbit -- took the gathers off the
// dependent chain and made the launch SLOWER: fused elevation step at 4096 envs 29.6 against 25.1 us, the visual-depth task's step 411
// against 399 us.  The lone wavefront of the latency forms pays ~3 ns per instruction whatever it waits for; the ~35 instructions of
// block selection and address arithmetic per sub-step cost more than the round trip to L2 they hide.  profiles/r05_probes/hf_look_ahead.diff,
// hf_look_ahead_scan_stagers.jsonl.)
inline HeightFieldGround make_ground(const WlHeightField* hf) { return HeightFieldGround{*hf, 1.f / hf->cell}; }

}  // namespace
