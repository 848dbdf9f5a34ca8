/* oracle/depth.c -- TEST INFRASTRUCTURE (CPU restatement; never linked into or called by the product).
 *
 * distance_to_image_plane of the visual task's pinhole camera against a heightfield terrain: the BASELINE.json config 5
 * kernel ("depth raycast against heightfield").  The reference only forwards what IsaacLab's camera sensor renders
 * (wheeledlab_tasks/visual/mdp_sensors/observations.py:89-95 `camera_data_depth` / `raycast_depth` ->
 * `sensor.data.output["distance_to_image_plane"]`; camera: visual/mushr_visual_env_cfg.py:230-246, 60 x 80 pinhole,
 * focal 1.93 / apertures 3.896 x 2.453, clipping range (0.01, 100)); the renderer (RTX / Warp) is closed or un-vendored and
 * the terrain mesh is missing, so the SURFACE is the designed one of oracle/heightfield.py (bilinear patches on a regular
 * grid; outside the grid the plane z = outside_z) and this file is the executable definition of "depth": parity unpinned
 * against IsaacLab, pinned against a brute-force marcher over oracle/heightfield.py::sample (tests/test_oracle_depth.py).
 *
 * Method (double precision, exact per cell): Amanatides-Woo walk over the grid cells the ray's ground track crosses; in
 * a cell the bilinear patch along the ray is a quadratic g(s) = A s^2 + B s + C (height of the ray above the patch), the
 * first root inside the cell's parameter interval is the hit.  The terrain is a SOLID: a ray that enters a cell (or the grid
 * through its side wall, or starts) below the surface hits at the entry parameter.  The ray parameter t is the distance
 * along the optical axis (the body-frame direction is (1, dy, dz)), i.e. distance_to_image_plane.
 *
 * No hierarchy, no fp32: deliberately a different traversal from the HIP kernel's (max-pyramid, fp32).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

typedef struct {
    const float* h;
    int nx, ny;
    double x0, y0, inv_cell, outside_z;
} Field;

/* first t in [ta, tb] at which the ray is on or below the plane z = outside_z; < 0: none */
static double plane_hit(double oz, double dz, double outside_z, double ta, double tb) {
    if (tb < ta) return -1.0;
    if (oz + ta * dz <= outside_z) return ta;
    if (dz < 0.0) {
        const double tp = (outside_z - oz) / dz;
        if (tp <= tb) return tp > ta ? tp : ta;
    }
    return -1.0;
}

/* smallest s in [0, smax] with A s^2 + B s + C <= 0, given C > 0; < 0: none */
static double first_root(double A, double B, double C, double smax) {
    const double disc = B * B - 4.0 * A * C;
    if (disc < 0.0) return -1.0;
    const double sq = sqrt(disc);
    const double q = -0.5 * (B + (B >= 0.0 ? sq : -sq));
    double s = INFINITY;
    if (A != 0.0) {
        const double r1 = q / A;
        if (r1 > 0.0 && r1 < s) s = r1;
    }
    if (q != 0.0) {
        const double r2 = C / q;
        if (r2 > 0.0 && r2 < s) s = r2;
    }
    return s <= smax ? s : -1.0;
}

static double cast_ray(const Field* f, double ox, double oy, double oz, double dx, double dy, double dz, double tmax) {
    const int NX = f->nx - 1, NY = f->ny - 1; /* cells */
    const double ou = (ox - f->x0) * f->inv_cell, ov = (oy - f->y0) * f->inv_cell;
    const double du = dx * f->inv_cell, dv = dy * f->inv_cell;
    /* parameter interval [t_in, t_out] of the ray's ground track inside the grid domain [0, NX] x [0, NY] */
    double t_in = -INFINITY, t_out = INFINITY;
    if (du != 0.0) {
        double a = (0.0 - ou) / du, b = ((double)NX - ou) / du;
        if (a > b) { double c = a; a = b; b = c; }
        if (a > t_in) t_in = a;
        if (b < t_out) t_out = b;
    } else if (ou < 0.0 || ou >= (double)NX) {
        t_in = INFINITY;
    }
    if (dv != 0.0) {
        double a = (0.0 - ov) / dv, b = ((double)NY - ov) / dv;
        if (a > b) { double c = a; a = b; b = c; }
        if (a > t_in) t_in = a;
        if (b < t_out) t_out = b;
    } else if (ov < 0.0 || ov >= (double)NY) {
        t_in = INFINITY;
    }
    if (!(t_in <= t_out) || t_out < 0.0 || t_in > tmax) { /* never over the grid within range */
        const double t = plane_hit(oz, dz, f->outside_z, 0.0, tmax);
        return t >= 0.0 ? t : tmax;
    }
    if (t_in > 0.0) { /* starts outside: the outside plane first */
        const double t = plane_hit(oz, dz, f->outside_z, 0.0, t_in);
        if (t >= 0.0) return t;
    } else {
        t_in = 0.0;
    }
    const double t_stop = t_out < tmax ? t_out : tmax;
    double t = t_in;
    const double u = ou + t * du, v = ov + t * dv;
    int i = (int)floor(u), j = (int)floor(v);
    if ((double)i == u && du < 0.0) --i; /* on a cell line heading down: the cell behind the line */
    if ((double)j == v && dv < 0.0) --j;
    if (i < 0) i = 0;
    if (i > NX - 1) i = NX - 1;
    if (j < 0) j = 0;
    if (j > NY - 1) j = NY - 1;
    const int si = du > 0.0 ? 1 : -1, sj = dv > 0.0 ? 1 : -1;
    for (;;) {
        const double tx = du != 0.0 ? ((double)(du > 0.0 ? i + 1 : i) - ou) / du : INFINITY;
        const double ty = dv != 0.0 ? ((double)(dv > 0.0 ? j + 1 : j) - ov) / dv : INFINITY;
        double te = tx < ty ? tx : ty;
        if (te > t_stop) te = t_stop;
        if (te < t) te = t;
        {
            const float* r0 = f->h + (int64_t)j * f->nx + i;
            const double h00 = r0[0], h10 = r0[1], h01 = r0[f->nx], h11 = r0[f->nx + 1];
            const double hx = h10 - h00, hy = h01 - h00, hxy = h11 - h10 - h01 + h00;
            double fu = ou + t * du - (double)i, fv = ov + t * dv - (double)j;
            fu = fu < 0.0 ? 0.0 : (fu > 1.0 ? 1.0 : fu);
            fv = fv < 0.0 ? 0.0 : (fv > 1.0 ? 1.0 : fv);
            const double C = oz + t * dz - (h00 + fu * hx + fv * hy + fu * fv * hxy);
            if (C <= 0.0) return t;
            const double A = -du * dv * hxy;
            const double B = dz - du * hx - dv * hy - (fu * dv + fv * du) * hxy;
            const double s = first_root(A, B, C, te - t);
            if (s >= 0.0) return t + s;
        }
        if (te >= t_stop) break;
        if (tx <= ty) i += si; else j += sj;
        if (i < 0 || i >= NX || j < 0 || j >= NY) break;
        t = te;
    }
    if (t_out < tmax) { /* left the grid: the outside plane beyond */
        const double th = plane_hit(oz, dz, f->outside_z, t_out, tmax);
        if (th >= 0.0) return th;
    }
    return tmax;
}

/* pos [n][3], quat [n][4] (w, x, y, z) of the ROOT; camera at cam_pos in the body frame, optical axis body +x, image right
 * = body -y, image down = body -z (the kernel's pixel_ray_body); depth [n][img_h][img_w].  Returns 0. */
int wl_oracle_depth(int n, const float* pos, const float* quat, const float* cam_pos, float fx, float fy, float cx, float cy,
                    int img_h, int img_w, const float* height, int nx, int ny, float x0, float y0, float cell, float outside_z,
                    float max_depth, float* depth) {
    Field f;
    f.h = height;
    f.nx = nx;
    f.ny = ny;
    f.x0 = x0;
    f.y0 = y0;
    f.inv_cell = 1.0 / (double)cell;
    f.outside_z = outside_z;
#pragma omp parallel for schedule(dynamic, 1)
    for (int e = 0; e < n; ++e) {
        const double w = quat[4 * e], x = quat[4 * e + 1], y = quat[4 * e + 2], z = quat[4 * e + 3];
        /* matrix_from_quat (oracle/mathlib.py) */
        const double R[3][3] = {{1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)},
                                {2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)},
                                {2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)}};
        double o[3];
        for (int k = 0; k < 3; ++k)
            o[k] = (double)pos[3 * e + k] + R[k][0] * cam_pos[0] + R[k][1] * cam_pos[1] + R[k][2] * cam_pos[2];
        for (int r = 0; r < img_h; ++r) {
            const double bz = -(((double)r + 0.5 - (double)cy) / (double)fy);
            for (int c = 0; c < img_w; ++c) {
                const double by = -(((double)c + 0.5 - (double)cx) / (double)fx);
                const double dx = R[0][0] + R[0][1] * by + R[0][2] * bz;
                const double dy = R[1][0] + R[1][1] * by + R[1][2] * bz;
                const double dz = R[2][0] + R[2][1] * by + R[2][2] * bz;
                depth[((int64_t)e * img_h + r) * img_w + c] = (float)cast_ray(&f, o[0], o[1], o[2], dx, dy, dz, (double)max_depth);
            }
        }
    }
    return 0;
}
