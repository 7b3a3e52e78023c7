// ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked into the product path.
// C entry points (ctypes-friendly) over the CPU restatement.  Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may load liboracle.so; the product library (libisdf_accel.so) never does.
#include "cost.hpp"
#include "minco.hpp"
#include "frontend.hpp"
#include <cstring>
#include <memory>
#include <string>

using namespace orc;

namespace {
struct OrcCtx {
    isdf_config cfg;
    CostParams P;
    Grid grid;
    MeshSdf mesh;
    SweptCtx sv;
    std::vector<double> points;
    std::vector<double> mesh_v;
    std::vector<int32_t> mesh_f;
    bool has_shape = false;
    FrontEnd fe;
    void sync() {
        P.flat.mass = cfg.vehicle_mass; P.flat.grav = cfg.grav_acc; P.flat.dh = cfg.horiz_drag;
        P.flat.dv = cfg.vert_drag; P.flat.cp = cfg.paras_drag; P.flat.veps = cfg.speed_eps;
        P.grid = &grid; P.mesh = &mesh;
        P.variant = cfg.variant; P.kernel_size = cfg.kernel_size; P.K = cfg.integral_intervs;
        P.enable_dyn = cfg.enable_dyn; P.enable_pos = cfg.enable_pos; P.enable_cull = cfg.enable_cull;
        P.safety_hor = cfg.safety_hor; P.weight_p = cfg.weight_p; P.weight_v = cfg.weight_v;
        P.weight_omg = cfg.weight_omg; P.weight_theta = cfg.weight_theta; P.vmax = cfg.vmax;
        P.omgmax = cfg.omgmax; P.thetamax = cfg.thetamax; P.smoothing_eps = cfg.smoothing_eps;
        P.occ_thresh = cfg.occ_thresh;
        sv.flat = P.flat; sv.shape = P.shape; sv.mesh = &mesh; sv.safety_hor = cfg.safety_hor;
    }
};
} // namespace

extern "C" {

void *orc_create(const isdf_config *cfg) {
    auto *c = new OrcCtx();
    c->cfg = *cfg;
    c->sync();
    return c;
}
void orc_destroy(void *h) { delete (OrcCtx *)h; }
void orc_set_threads(void *h, int n) { ((OrcCtx *)h)->P.threads = n > 0 ? n : 1; }
int orc_max_threads(void) { return omp_get_max_threads(); }
void orc_set_shard(void *h, int rank, int world) { ((OrcCtx *)h)->P.shard_rank = rank; ((OrcCtx *)h)->P.shard_world = world; }

int orc_set_grid(void *h, const void *vox, int dtype, int nx, int ny, int nz, const double origin[3],
                 const double bmax[3], double res, int kind) {
    OrcCtx &c = *(OrcCtx *)h;
    Grid &g = c.grid;
    g.X = nx; g.Y = ny; g.Z = nz; g.res = res;
    g.bmin = V3(origin[0], origin[1], origin[2]);
    g.bmax = bmax ? V3(bmax[0], bmax[1], bmax[2]) : V3(origin[0] + nx * res, origin[1] + ny * res, origin[2] + nz * res);
    const size_t n = (size_t)nx * ny * nz;
    auto get = [&](size_t i) -> double {
        if (dtype == ISDF_U8) return ((const uint8_t *)vox)[i];
        if (dtype == ISDF_F32) return ((const float *)vox)[i];
        return ((const double *)vox)[i];
    };
    if (kind == ISDF_GRID_OCCUPANCY) {
        g.occ.resize(n);
        for (size_t i = 0; i < n; i++) g.occ[i] = get(i) != 0 ? 1 : 0;
        g.has_occ = true;
    } else {
        g.esdf.resize(n);
        for (size_t i = 0; i < n; i++) g.esdf[i] = (float)get(i);
        g.has_esdf = true;
    }
    return 0;
}

int orc_set_shape(void *h, const isdf_shape *s) {
    OrcCtx &c = *(OrcCtx *)h;
    c.P.shape = *s;
    if (s->kind == ISDF_SHAPE_MESH) {
        c.mesh.set(s->mesh_vertices, s->n_vertices, s->mesh_faces, s->n_faces);
        c.P.shape.mesh_vertices = nullptr; c.P.shape.mesh_faces = nullptr;
    }
    c.has_shape = true;
    c.sync();
    return 0;
}

// ISDF_SHAPE_GRID: the lattice of isdf_set_shape_grid (cells: n x 4 doubles)
int orc_set_shape_grid(void *h, const double *cells, int nx, int ny, int nz, const double *gmin, double nres, double bound_radius) {
    OrcCtx &c = *(OrcCtx *)h;
    isdf_shape s{};
    s.kind = ISDF_SHAPE_GRID; s.grad_mode = ISDF_GRAD_GRID; s.bound_radius = bound_radius;
    for (int i = 0; i < 9; i++) s.rotate[i] = (i % 4 == 0) ? 1.0 : 0.0;
    c.P.shape = s;
    c.mesh.grid_cells.assign(cells, cells + (size_t)4 * nx * ny * nz);
    c.mesh.gn[0] = nx; c.mesh.gn[1] = ny; c.mesh.gn[2] = nz; c.mesh.gres = nres;
    for (int a = 0; a < 3; a++) c.mesh.gmin[a] = gmin[a];
    c.has_shape = true;
    c.sync();
    return 0;
}

// ---- front end (kernel convolution), oracle/frontend.hpp
int orc_frontend_build(void *h, const isdf_frontend_config *cfg) {
    OrcCtx &c = *(OrcCtx *)h;
    if (!c.has_shape || !c.grid.has_occ) return -1;
    c.fe.configure(*cfg, c.grid.res);
    c.fe.build_shape_kernels(c.P.shape, &c.mesh);
    c.fe.build_map_kernel(c.grid);
    return 0;
}
void orc_frontend_dims(void *h, int *out) {     // xk, yk, bytes per attitude, inflated X, Y, bytes per z-row
    OrcCtx &c = *(OrcCtx *)h;
    const int hh = (c.fe.k - 1) / 2;
    out[0] = c.fe.xk; out[1] = c.fe.yk; out[2] = c.fe.k * c.fe.k * c.fe.bl;
    out[3] = c.fe.mX + 2 * hh; out[4] = c.fe.mY + 2 * hh; out[5] = (c.fe.mZ + 2 * hh + 7) / 8;
}
void orc_frontend_shape_kernels(void *h, uint8_t *out) { OrcCtx &c = *(OrcCtx *)h; std::memcpy(out, c.fe.shape_bytes.data(), c.fe.shape_bytes.size()); }
void orc_frontend_map_kernel(void *h, uint8_t *out) { OrcCtx &c = *(OrcCtx *)h; std::memcpy(out, c.fe.map_kernel.data(), c.fe.map_kernel.size() - 1); }
void orc_frontend_rotations(void *h, double *out) { OrcCtx &c = *(OrcCtx *)h; std::memcpy(out, c.fe.rot.data(), c.fe.rot.size() * sizeof(double)); }
void orc_frontend_check(void *h, int n, const int32_t *index, const double *fr, const double *fp, uint8_t *ok, double *cr, double *cp, int32_t *kidx) {
    OrcCtx &c = *(OrcCtx *)h;
#pragma omp parallel for schedule(dynamic, 64) num_threads(c.P.threads)
    for (int q = 0; q < n; q++) {
        const int ind[3] = {index[3 * q], index[3 * q + 1], index[3 * q + 2]};
        double r = 0, p = 0;
        int ki = -1;
        const bool good = c.fe.successor_ok(c.grid, fr[q], fp[q], r, p, ind, &ki);
        ok[q] = good ? 1 : 0;
        if (good) { cr[q] = r; cp[q] = p; }
        if (kidx) kidx[q] = ki;
    }
}
// every voxel x every attitude: kernel_conv; 4 * ceil(attitudes / 128) dwords per voxel, occupied voxels 0
void orc_frontend_cspace(void *h, uint32_t *out) {
    OrcCtx &c = *(OrcCtx *)h;
    const Grid &g = c.grid;
    const int na = c.fe.xk * c.fe.yk;
    const int nw = 4 * ((na + 127) / 128);
#pragma omp parallel for schedule(dynamic, 16) num_threads(c.P.threads)
    for (int x = 0; x < g.X; x++)
        for (int y = 0; y < g.Y; y++)
            for (int z = 0; z < g.Z; z++) {
                uint32_t *m = out + (size_t)nw * g.addr(x, y, z);
                for (int w = 0; w < nw; w++) m[w] = 0;
                if (g.occ[g.addr(x, y, z)] == 1) continue;
                const int ind[3] = {x, y, z};
                for (int a = 0; a < na; a++)
                    if (c.fe.kernel_conv(a / c.fe.yk, a % c.fe.yk, ind)) m[a >> 5] |= 1u << (a & 31);
            }
}
// front_end_Astar.hpp:238-403 driven like PlannerManager::generatePath: -1 = no path, else the number of path nodes (written
// up to cap); stats = [neighbour tests, nodes taken off the open set]
int orc_frontend_astar(void *h, const double *start, const double *goal, int cap, double *path_xyz, double *path_rp, int64_t *stats) {
    OrcCtx &c = *(OrcCtx *)h;
    const orc::FrontEnd::AstarOut o = c.fe.astar(c.grid, orc::V3(start[0], start[1], start[2]), orc::V3(goal[0], goal[1], goal[2]));
    if (stats) { stats[0] = o.checks; stats[1] = o.expansions; }
    if (!o.success) return -1;
    const int n = (int)(o.rp.size() / 2);
    for (int i = 0; i < n && i < cap; i++) {
        for (int k = 0; k < 3; k++) path_xyz[3 * i + k] = o.xyz[3 * (size_t)i + k];
        path_rp[2 * i] = o.rp[2 * (size_t)i]; path_rp[2 * i + 1] = o.rp[2 * (size_t)i + 1];
    }
    return n;
}
int orc_frontend_kernel_conv(void *h, int ki, int kj, const int32_t *ind) { OrcCtx &c = *(OrcCtx *)h; const int i3[3] = {ind[0], ind[1], ind[2]}; return c.fe.kernel_conv(ki, kj, i3) ? 1 : 0; }

// 1: the mesh winding number comes from the vendored HDK library (oracle/_ref/libref_fwn.so); 0: exact-sum fallback
int orc_mesh_backend(void *h) { return ((OrcCtx *)h)->mesh.backend(); }

int orc_set_points(void *h, const double *xyz, int M) {
    OrcCtx &c = *(OrcCtx *)h;
    c.points.assign(xyz, xyz + 3 * (size_t)M);
    return 0;
}

// One trajectory; accumulates.  stats_out[4] = units, culled, pairs, grad_pairs (may be NULL).
int orc_eval(void *h, int N, const double *T, const double *coeffs, double *cost_inout, double *gradT_inout,
             double *gradC_inout, double *tstar_inout, long long *stats_out) {
    OrcCtx &c = *(OrcCtx *)h;
    c.sync();
    Traj tr; tr.N = N; tr.T = T; tr.C = coeffs;
    SweepStats st;
    if (c.cfg.variant == ISDF_V1_SWEPT) {
        const int M = (int)(c.points.size() / 3);
        std::vector<double> tmp;
        double *ts = tstar_inout;
        if (!ts) { tmp.assign(M, 0.0); ts = tmp.data(); }
        sweep_swept(c.P, c.sv, tr, c.points.data(), M, ts, *cost_inout, gradT_inout, gradC_inout, &st);
    } else {
        sweep_integral(c.P, tr, *cost_inout, gradT_inout, gradC_inout, &st);
    }
    if (stats_out) { stats_out[0] = st.units; stats_out[1] = st.culled; stats_out[2] = st.pairs; stats_out[3] = st.grad_pairs; }
    return 0;
}

// ---- unit-level probes used by the oracle self-tests and the GPU parity tests ----
// grad_cost_p for one pose (position, quaternion w x y z); returns costp
double orc_grad_cost_p(void *h, const double pos[3], const double quat[4], double gradp[3], double gradq[4], int *active) {
    OrcCtx &c = *(OrcCtx *)h;
    c.sync();
    V4 q; q.w = quat[0]; q.x = quat[1]; q.y = quat[2]; q.z = quat[3];
    const M3 R = quat_to_rot(q);
    V3 gp; V4 gq; double costp = 0.0;
    const bool a = grad_cost_p(c.P, V3(pos[0], pos[1], pos[2]), R, q, gp, gq, costp, nullptr);
    *active = a ? 1 : 0;
    gradp[0] = gp.x; gradp[1] = gp.y; gradp[2] = gp.z;
    gradq[0] = gq.w; gradq[1] = gq.x; gradq[2] = gq.y; gradq[3] = gq.z;
    return costp;
}
void orc_flat_forward(const double fp[6], const double vel[3], const double acc[3], const double jer[3],
                      double quat[4], double omg[3]) {
    FlatParams P{fp[0], fp[1], fp[2], fp[3], fp[4], fp[5]};
    V4 q; V3 w;
    flat_forward(P, V3(vel[0], vel[1], vel[2]), V3(acc[0], acc[1], acc[2]), V3(jer[0], jer[1], jer[2]), q, w);
    V4 q2;
    flat_forward(P, V3(vel[0], vel[1], vel[2]), V3(acc[0], acc[1], acc[2]), V3(jer[0], jer[1], jer[2]), q2);
    (void)q2;
    quat[0] = q.w; quat[1] = q.x; quat[2] = q.y; quat[3] = q.z;
    omg[0] = w.x; omg[1] = w.y; omg[2] = w.z;
}
void orc_flat_backward(const double fp[6], const double vel[3], const double acc[3], const double jer[3],
                       const double pos_grad[3], const double vel_grad[3], const double quat_grad[4],
                       const double omg_grad[3], double out12[12]) {
    FlatParams P{fp[0], fp[1], fp[2], fp[3], fp[4], fp[5]};
    V4 qg; qg.w = quat_grad[0]; qg.x = quat_grad[1]; qg.y = quat_grad[2]; qg.z = quat_grad[3];
    V3 a, b, c, d;
    flat_backward(P, V3(vel[0], vel[1], vel[2]), V3(acc[0], acc[1], acc[2]), V3(jer[0], jer[1], jer[2]),
                  V3(pos_grad[0], pos_grad[1], pos_grad[2]), V3(vel_grad[0], vel_grad[1], vel_grad[2]), qg,
                  V3(omg_grad[0], omg_grad[1], omg_grad[2]), a, b, c, d);
    const V3 *o[4] = {&a, &b, &c, &d};
    for (int i = 0; i < 4; i++) { out12[3 * i] = o[i]->x; out12[3 * i + 1] = o[i]->y; out12[3 * i + 2] = o[i]->z; }
}
void orc_quat_to_rot(const double q[4], double R9[9]) {
    V4 v; v.w = q[0]; v.x = q[1]; v.y = q[2]; v.z = q[3];
    const M3 R = quat_to_rot(v);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R9[3 * i + j] = R.m[i][j];
}
// sdf_out[n], grad_out[3n] (either may be NULL)
void orc_shape_eval(void *h, const double *pts, int n, double *sdf_out, double *grad_out) {
    OrcCtx &c = *(OrcCtx *)h;
    for (int i = 0; i < n; i++) {
        const V3 p(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
        if (sdf_out) sdf_out[i] = shape_sdf(c.P.shape, &c.mesh, p);
        if (grad_out) {
            const V3 g = shape_grad(c.P.shape, &c.mesh, p);
            grad_out[3 * i] = g.x; grad_out[3 * i + 1] = g.y; grad_out[3 * i + 2] = g.z;
        }
    }
}
double orc_smoothed_l1(double x, double mu, double *df) {
    double f = 0, d = 0;
    if (!smoothed_l1(x, mu, f, d)) { *df = 0; return 0; }
    *df = d;
    return f;
}
// occupancy (u8) -> ESDF in metres (double), the reference's generateESDF3d positive pass
void orc_build_esdf(const uint8_t *occ, int nx, int ny, int nz, double res, double *out) {
    Grid g; g.X = nx; g.Y = ny; g.Z = nz; g.res = res;
    g.occ.assign(occ, occ + (size_t)nx * ny * nz);
    std::vector<double> o;
    build_esdf(g, o);
    std::memcpy(out, o.data(), o.size() * sizeof(double));
}
double orc_esdf_trilinear(void *h, const double pos[3], double grad[3]) {
    OrcCtx &c = *(OrcCtx *)h;
    V3 g;
    const double d = c.grid.esdf_trilinear(V3(pos[0], pos[1], pos[2]), g);
    grad[0] = g.x; grad[1] = g.y; grad[2] = g.z;
    return d;
}
void orc_aabb_index_box(void *h, const double pos[3], double half, int lo[3], int hi[3]) {
    ((OrcCtx *)h)->grid.aabb_index_box(V3(pos[0], pos[1], pos[2]), half, lo, hi);
}
// trajectory probes
void orc_traj_eval(int N, const double *T, const double *coeffs, double t, double out12[12], int *piece, double *tlocal) {
    Traj tr; tr.N = N; tr.T = T; tr.C = coeffs;
    double tl = t;
    const int i = locate_piece(tr, tl);
    V3 p, v, a, j;
    eval_pvaj_local(tr, i, tl, p, v, a, j);
    const V3 *o[4] = {&p, &v, &a, &j};
    for (int k = 0; k < 4; k++) { out12[3 * k] = o[k]->x; out12[3 * k + 1] = o[k]->y; out12[3 * k + 2] = o[k]->z; }
    *piece = i; *tlocal = tl;
}
// swept-volume probe for one point: returns min sdf; tstar in/out; grad_prel[3]; n_ranges
double orc_swept_sdf(void *h, int N, const double *T, const double *coeffs, const double p[3], double *tstar,
                     double grad_prel[3], int *n_ranges) {
    OrcCtx &c = *(OrcCtx *)h;
    c.sync();
    Traj tr; tr.N = N; tr.T = T; tr.C = coeffs;
    c.sv.update_traj(tr);
    V3 g;
    const double s = c.sv.sdf_of_swept_volume(V3(p[0], p[1], p[2]), *tstar, g, n_ranges);
    grad_prel[0] = g.x; grad_prel[1] = g.y; grad_prel[2] = g.z;
    return s;
}

// developer probe (tools/v1_trace.py): per interval of one point's swept-volume query, the number of sign-descent passes
// (outer iterations of gradientDescent), the iteration counter it ends with, and how many passes accepted the FULL step
// (div == 1) in the same direction as the pass before - the marching part of a descent.  out: 3 ints per interval.
int orc_swept_trace(void *h, int N, const double *T, const double *coeffs, const double p[3], int cap, int *out) {
    OrcCtx &c = *(OrcCtx *)h;
    c.sync();
    Traj tr; tr.N = N; tr.T = T; tr.C = coeffs;
    c.sv.update_traj(tr);
    const V3 P3(p[0], p[1], p[2]);
    std::vector<double> rl, rr, rts;
    c.sv.choice_t_init(P3, 0.02, rl, rr, rts);
    const int n = (int)rl.size();
    for (int i = 0; i < n && i < cap; i++) {
        const double t_min = std::max(0.0, rl[i]), t_max = std::min(rr[i], c.sv.traj_duration);
        double x = rts[i], prev_x = 10000000.0, fx = 0.0;
        int iter = 0, passes = 0, march = 0, last_sgn = 0;
        bool stop = false;
        while (iter < 300 && !stop && std::fabs(x - prev_x) > 1e-5) {
            if (iter == 0) fx = c.sv.sdf_at_time(P3, x);
            const double g = c.sv.sdf_dot_at_time(P3, x);
            const int sgn = (int)(g > 0) - (int)(g < 0);
            double tau = 0.02;
            prev_x = x;
            passes++;
            for (int div = 1; div < 10; div++) {
                iter++;
                double xc = std::max(std::min(x - tau * sgn, t_max), t_min);
                const double fc = c.sv.sdf_at_time(P3, xc);
                if ((fc - fx) < 0) { x = xc; fx = fc; if (div == 1 && sgn == last_sgn) march++; break; }
                tau *= 0.5;
                if (div == 9) stop = true;
            }
            last_sgn = sgn;
        }
        out[3 * i] = passes; out[3 * i + 1] = iter; out[3 * i + 2] = march;
    }
    return n;
}

// ---- MINCO (N1 "next" row; also the synthetic-trajectory generator of the tests) ----
// head/tail: 3x3 column-major state matrices [pos vel acc] (Eigen::Matrix3d, col(0)=pos ...).
// inPs: 3 x (N-1) column-major waypoints.  coeffs_out: 6N x 3 column-major.
int orc_minco_coeffs(int N, const double *head, const double *tail, const double *inPs, const double *T, double *coeffs_out) {
    Minco m;
    m.set_conditions(head, tail, N);
    m.set_parameters(inPs, T);
    std::memcpy(coeffs_out, m.b.data(), sizeof(double) * 18 * (size_t)N);
    return 0;
}
// energy and its partials (minco.hpp:530-582) for a coefficient matrix
double orc_minco_energy(int N, const double *coeffs, const double *T, double *gdC, double *gdT) {
    return minco_energy(N, coeffs, T, gdC, gdT);
}
// adjoint propagation: (dCost/dc, dCost/dT) -> (dCost/dq (3 x (N-1) colmajor), dCost/dT)
int orc_minco_propagate(int N, const double *head, const double *tail, const double *inPs, const double *T,
                        const double *gradC, const double *gradT_partial, double *gradP_out, double *gradT_out) {
    Minco m;
    m.set_conditions(head, tail, N);
    m.set_parameters(inPs, T);
    m.propagate_grad(gradC, gradT_partial, gradP_out, gradT_out);
    return 0;
}

} // extern "C"
