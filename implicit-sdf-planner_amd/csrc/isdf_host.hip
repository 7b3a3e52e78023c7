// Host side of the C ABI (include/isdf_accel.h): context, device memory, launches.
// Replaces, for the two sweeps only, what TrajOptimizer holds around them in the reference
// (back_end_optimizer.hpp:59-62 parallel_points/lastTstar, :667-725 setParam/setEnvironment/setGridMap) and the
// shape registry lookup of sw_manager.hpp:74-123,:255-275.  No CPU compute path exists here.
#include "isdf_ctx.hpp"
#include "minco_dev.hpp"
#include "lbfgs_host.hpp"
#include <array>
#include <atomic>
#include <map>
#include <dlfcn.h>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <cstdlib>
#include <string>
#include <vector>
#include <new>
#include "fwn_host.hpp"

using namespace isdf;

namespace {
thread_local std::string g_create_error;
}

int isdf_fail(isdf_ctx *c, int code, const char *msg) {
    if (c) c->err = msg; else g_create_error = msg;
    return code;
}
static int fail(isdf_ctx *c, int code, const char *msg) { return isdf_fail(c, code, msg); }

// --------------------------------------------------------------------------------------------------------------
// defaults and the analytic-shape registry
// --------------------------------------------------------------------------------------------------------------
extern "C" int isdf_abi_version(void) { return ISDF_ABI_VERSION; }
extern "C" size_t isdf_out_stride(int N) { return (size_t)1 + 19 * (size_t)N; }

extern "C" void isdf_config_default(isdf_config *c) {   // config_CappedCone.yaml (demo1)
    std::memset(c, 0, sizeof(*c));
    c->device = 0;
    c->variant = ISDF_V3_ESDF_TILE;
    c->kernel_size = 13;
    c->integral_intervs = 64;
    c->enable_dyn = 1;
    c->enable_pos = 1;
    c->enable_cull = 0;
    c->safety_hor = 0.866;
    c->weight_p = 4000.0;
    c->weight_v = c->weight_omg = c->weight_theta = 1000.0;
    c->vmax = 10.0; c->omgmax = 10.0; c->thetamax = 100.0;
    c->smoothing_eps = 1.0e-2;
    c->occ_thresh = 0.0;
    c->vehicle_mass = 0.61; c->grav_acc = 9.8; c->horiz_drag = 0.10; c->vert_drag = 0.10;
    c->paras_drag = 0.01; c->speed_eps = 1.0e-4;
}

static void shape_identity(isdf_shape *s, int kind) {
    std::memset(s, 0, sizeof(*s));
    s->kind = kind;
    s->grad_mode = ISDF_GRAD_DEFAULT;
    s->rotate[0] = s->rotate[4] = s->rotate[8] = 1.0;
}
static void setp(isdf_shape *s, std::initializer_list<double> v) {
    int i = 0;
    for (double x : v) s->params[i++] = x;
}

extern "C" int isdf_shape_default(isdf_shape *s, int kind) {
    if (!s || kind < 0 || kind >= ISDF_SHAPE_KIND_COUNT) return ISDF_ERR_INVALID_ARG;
    shape_identity(s, kind);
    switch (kind) {
    case ISDF_SHAPE_TORUS: setp(s, {2.5, 0.3}); break;
    case ISDF_SHAPE_CAPPEDTORUS: setp(s, {std::sin(40), std::cos(40), 3.5, 0.3}); break;
    case ISDF_SHAPE_CAPPEDCONE: setp(s, {2.0, 0.8, 0, 0, -1, 0, 0, 1}); break;
    case ISDF_SHAPE_ROUNDEDCONE: setp(s, {1.5, 0.6, 4.5}); break;
    case ISDF_SHAPE_WIREFRAMEBOX: setp(s, {1.8, 2.5, 3.5, 0.1}); break;
    case ISDF_SHAPE_BENDLINEAR: setp(s, {2.0, 0.25}); break;
    case ISDF_SHAPE_TWISTBOX: setp(s, {2.0, 2.0, 2.0, 3.14159265358979323846 / 6}); break;
    case ISDF_SHAPE_BENDBOX: setp(s, {2.0, 2.0, 2.0, 0.5}); break;
    case ISDF_SHAPE_TABLE: setp(s, {0.0, 0.0, 0.0, 3.5, 1.75, 0.7, 2.8, 1.05, 0.0, 3.5, 1.75, 2.8}); break;
    case ISDF_SHAPE_TREFOIL: setp(s, {3.5, 0.2, 0.2, 0.05, 0.4}); break;
    case ISDF_SHAPE_SMOOTHDIFFERENCE: setp(s, {3.0, 3.0, 0.5, 1.0, 0.25}); break;
    case ISDF_SHAPE_SMOOTHINTERSECTION: setp(s, {3.0, 3.0, 0.5, 1.0, 0.25}); break;
    case ISDF_SHAPE_CSG: setp(s, {3.0, 4.5, 1.5}); break;
    case ISDF_SHAPE_BOX: setp(s, {3.0, 0.3, 0.3}); break;
    case ISDF_SHAPE_BALL: setp(s, {1.0}); break;
    default: break;
    }
    return ISDF_OK;
}

extern "C" int isdf_shape_from_name(isdf_shape *s, const char *stem) {
    if (!s || !stem) return ISDF_ERR_INVALID_ARG;
    struct Ent { const char *name; int kind; };
    static const Ent reg[] = {   // sw_manager.hpp:74-123
        {"CSG", ISDF_SHAPE_CSG}, {"Torus", ISDF_SHAPE_TORUS}, {"Torus_big", ISDF_SHAPE_TORUS},
        {"Cappedtorus", ISDF_SHAPE_CAPPEDTORUS}, {"Trefoil", ISDF_SHAPE_TREFOIL}, {"Table", ISDF_SHAPE_TABLE},
        {"CappedCone", ISDF_SHAPE_CAPPEDCONE}, {"RoundedCone", ISDF_SHAPE_ROUNDEDCONE},
        {"WireframeBox", ISDF_SHAPE_WIREFRAMEBOX}, {"BendLinear", ISDF_SHAPE_BENDLINEAR},
        {"BendLinear_big", ISDF_SHAPE_BENDLINEAR}, {"TwistBox", ISDF_SHAPE_TWISTBOX}, {"BendBox", ISDF_SHAPE_BENDBOX},
        {"SmoothDifference", ISDF_SHAPE_SMOOTHDIFFERENCE}, {"SmoothIntersection", ISDF_SHAPE_SMOOTHINTERSECTION},
        {"SmoothIntersection_big", ISDF_SHAPE_SMOOTHINTERSECTION}};
    for (const Ent &e : reg) {
        if (std::strcmp(e.name, stem) == 0) {
            isdf_shape_default(s, e.kind);
            if (!std::strcmp(stem, "Torus_big")) setp(s, {3.5, 0.3});
            if (!std::strcmp(stem, "BendLinear_big")) setp(s, {3.2, 0.45});
            if (!std::strcmp(stem, "SmoothIntersection_big")) setp(s, {9.0, 9.0, 1.5, 3.0, 0.25});
            return ISDF_OK;
        }
    }
    return ISDF_ERR_UNSUPPORTED;   // not analytic: the reference falls back to the mesh Generalshape (:263-274)
}

// --------------------------------------------------------------------------------------------------------------
// lifetime
// --------------------------------------------------------------------------------------------------------------
extern "C" const char *isdf_last_error(const isdf_ctx *c) { return c ? c->err.c_str() : g_create_error.c_str(); }

extern "C" int isdf_create(isdf_ctx **out, const isdf_config *cfg) {
    if (!out || !cfg) return fail(nullptr, ISDF_ERR_INVALID_ARG, "null argument");
    *out = nullptr;
    if (cfg->variant < ISDF_V1_SWEPT || cfg->variant > ISDF_V3_ESDF_TILE) return fail(nullptr, ISDF_ERR_INVALID_ARG, "bad variant");
    if (cfg->kernel_size < 1 || cfg->kernel_size > 512) return fail(nullptr, ISDF_ERR_INVALID_ARG, "kernel_size must be in [1,512]");
    if (cfg->integral_intervs < 1) return fail(nullptr, ISDF_ERR_INVALID_ARG, "integral_intervs must be >= 1");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, ISDF_ERR_NO_DEVICE, "no HIP device available (the product path has no CPU fallback)");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, ISDF_ERR_INVALID_ARG, "device ordinal out of range");
    isdf_ctx *c = new (std::nothrow) isdf_ctx();
    if (!c) return fail(nullptr, ISDF_ERR_HIP, "out of host memory");
    c->cfg = *cfg;
    c->device = cfg->device;
    {   // every environment switch of the host paths is read HERE, once per ctx
        auto on = [](const char *name) { const char *e = getenv(name); return e && e[0] == '1'; };
        c->fuse_small = !on("ISDF_NO_FUSE");
        c->env_no_direct = on("ISDF_NO_HOST_DIRECT");
        c->env_multi_no_hostout = on("ISDF_MULTI_NO_HOST_OUT");
        c->env_no_bar = on("ISDF_NO_BAR_WRITES");
        c->env_no_lpt = on("ISDF_NO_LPT");
        c->minco_mode = on("ISDF_HOST_MINCO") ? 1 : (on("ISDF_DEVICE_MINCO") ? 2 : 0);
    }
    if (hipSetDevice(c->device) != hipSuccess || hipMalloc(&c->d_stats, 8 * sizeof(unsigned long long)) != hipSuccess ||
        hipMalloc(&c->d_traj_duration, sizeof(double)) != hipSuccess || hipMalloc(&c->d_n_coarse, sizeof(int)) != hipSuccess ||
        hipMemset(c->d_traj_duration, 0, sizeof(double)) != hipSuccess || hipMemset(c->d_stats, 0, 8 * sizeof(unsigned long long)) != hipSuccess || hipStreamCreate(&c->stream) != hipSuccess) {
        delete c;
        return fail(nullptr, ISDF_ERR_HIP, "device initialisation failed");
    }
    *out = c;
    return ISDF_OK;
}

static void free_mesh(isdf_ctx *c) {
    if (c->d_mesh) (void)hipFree(c->d_mesh);
    if (c->d_mesh_tri) (void)hipFree(c->d_mesh_tri);
    if (c->d_mesh_trif) (void)hipFree(c->d_mesh_trif);
    if (c->d_fwn_child) (void)hipFree(c->d_fwn_child);
    if (c->d_fwn_box) (void)hipFree(c->d_fwn_box);
    if (c->d_fwn_boxq) (void)hipFree(c->d_fwn_boxq);
    if (c->d_fwn_triq) (void)hipFree(c->d_fwn_triq);
    if (c->d_mesh_dl) (void)hipFree(c->d_mesh_dl);
    c->d_mesh_dl = nullptr;
    if (c->d_mesh_flat) (void)hipFree(c->d_mesh_flat);
    c->d_mesh_flat = nullptr;
    c->d_mesh = nullptr; c->d_mesh_tri = nullptr; c->d_mesh_trif = nullptr; c->d_fwn_child = nullptr; c->d_fwn_box = nullptr; c->d_fwn_boxq = nullptr; c->d_fwn_triq = nullptr;
}

static void multi_release(isdf_ctx *c);
extern "C" int isdf_destroy(isdf_ctx *c) {
    if (!c) return ISDF_OK;
    for (isdf_ctx *p : c->peers) { p->is_peer = false; (void)isdf_destroy(p); }
    c->peers.clear();
    (void)hipSetDevice(c->device);
    multi_release(c);
    (void)hipDeviceSynchronize();
    for (auto &p : c->prof_events) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); (void)hipEventDestroy(p.c); (void)hipEventDestroy(p.d); }
    if (c->h_pin) (void)hipHostFree(c->h_pin);
    if (c->h_eval_pin) (void)hipHostFree(c->h_eval_pin);
    if (c->h_dir) (void)hipHostFree(c->h_dir);
    if (c->h_cbres) (void)hipHostFree(c->h_cbres);
    if (c->h_v1_pin) (void)hipHostFree(c->h_v1_pin);
    isdf_frontend_release(c);
    isdf_xchg_release(c);
    void *ptrs[] = {c->d_cb, c->d_cbdev, c->d_esdf_stage, c->d_esdf_bricks, c->d_esdf, c->d_occ, c->d_points, c->d_tstar, c->d_acc, c->d_sample_info, c->d_bits, c->d_piece_cost, c->d_in, c->d_out, c->d_stats,
                    c->d_traj_duration, c->d_coarse_t, c->d_coarse_pose, c->d_n_coarse, c->d_point_partial, c->d_point_piece, c->d_point_stat, c->d_point_nr, c->d_task_buf,
                    c->d_task_map, c->d_v1_words, c->d_point_lmask, c->d_scan_ticks, c->d_scan_order, c->d_hist, c->d_shape_grid, c->d_pose, c->d_mq_entries, c->d_mq_items, c->d_mq_res, c->d_mq_sample_items, c->d_mq_sample_n, c->d_mq_count, c->d_msum_blocks, c->d_dbg, c->d_stage, c->d_stage_flags, c->d_sample_map, c->d_plan_cls, c->d_plan_map, c->d_plan_lr, c->d_plan_hist};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    free_mesh(c);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return ISDF_OK;
}

// --------------------------------------------------------------------------------------------------------------
// once-per-plan state
// --------------------------------------------------------------------------------------------------------------
extern "C" int isdf_set_grid(isdf_ctx *c, const void *vox, int dtype, int nx, int ny, int nz, const double origin[3],
                             const double bmax[3], double res, int kind) {
    if (!c) return ISDF_ERR_INVALID_ARG;
    if (!vox || !origin || nx < 1 || ny < 1 || nz < 1 || !(res > 0)) return fail(c, ISDF_ERR_INVALID_ARG, "bad grid arguments");
    if (dtype < ISDF_U8 || dtype > ISDF_F64) return fail(c, ISDF_ERR_INVALID_ARG, "bad dtype");
    if (kind != ISDF_GRID_OCCUPANCY && kind != ISDF_GRID_ESDF) return fail(c, ISDF_ERR_INVALID_ARG, "bad grid kind");
    HIPCHK(c, hipSetDevice(c->device));
    isdf_frontend_release(c);       // the inflated bit-packed map was built from the previous grid
    const size_t n = (size_t)nx * ny * nz;
    if (c->have_geom && (c->grid.X != nx || c->grid.Y != ny || c->grid.Z != nz)) {
        // new geometry: drop the other grid kind, it no longer matches
        if (c->d_esdf) { (void)hipFree(c->d_esdf); c->d_esdf = nullptr; }
        if (c->d_occ) { (void)hipFree(c->d_occ); c->d_occ = nullptr; }
    }
    c->grid.X = nx; c->grid.Y = ny; c->grid.Z = nz; c->grid.res = res;
    for (int a = 0; a < 3; a++) {
        c->grid.bmin[a] = origin[a];
        const int dim = a == 0 ? nx : (a == 1 ? ny : nz);
        c->grid.bmax[a] = bmax ? bmax[a] : origin[a] + dim * res;
    }
    c->have_geom = true;
    if (kind == ISDF_GRID_ESDF) {
        std::vector<float> tmp;
        const float *src = nullptr;
        if (dtype == ISDF_F32) src = (const float *)vox;
        else {
            tmp.resize(n);
            if (dtype == ISDF_F64) for (size_t i = 0; i < n; i++) tmp[i] = (float)((const double *)vox)[i];
            else for (size_t i = 0; i < n; i++) tmp[i] = (float)((const uint8_t *)vox)[i];
            src = tmp.data();
        }
        if (c->d_esdf) { (void)hipFree(c->d_esdf); c->d_esdf = nullptr; }
        HIPCHK(c, hipMalloc(&c->d_esdf, n * sizeof(float)));
        HIPCHK(c, hipMemcpy(c->d_esdf, src, n * sizeof(float), hipMemcpyHostToDevice));
    } else {
        std::vector<uint8_t> tmp(n);
        if (dtype == ISDF_U8) for (size_t i = 0; i < n; i++) tmp[i] = ((const uint8_t *)vox)[i] != 0;
        else if (dtype == ISDF_F32) for (size_t i = 0; i < n; i++) tmp[i] = ((const float *)vox)[i] != 0;
        else for (size_t i = 0; i < n; i++) tmp[i] = ((const double *)vox)[i] != 0;
        if (c->d_occ) { (void)hipFree(c->d_occ); c->d_occ = nullptr; }
        HIPCHK(c, hipMalloc(&c->d_occ, n));
        HIPCHK(c, hipMemcpy(c->d_occ, tmp.data(), n, hipMemcpyHostToDevice));
    }
    c->grid.esdf = c->d_esdf;
    c->bricks_stale = true;
    c->grid.occ = c->d_occ;
    c->bits_dirty = true;
    ISDF_REPLICATE(c, isdf_set_grid(p_, vox, dtype, nx, ny, nz, origin, bmax, res, kind));
    return ISDF_OK;
}

// rotate()/rotate_to() of the CSG class (Shape.hpp:2016-2053), evaluated once on the host
static void csg_rotate_to(const double a_in[3], const double b_in[3], double R[9]) {
    auto nrm = [](const double v[3], double o[3]) {
        const double z = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
        const double s = z > 0 ? std::sqrt(z) : 1.0;
        for (int i = 0; i < 3; i++) o[i] = z > 0 ? v[i] / s : v[i];
    };
    double a[3], b[3];
    nrm(a_in, a); nrm(b_in, b);
    const double d = b[0] * a[0] + b[1] * a[1] + b[2] * a[2];
    for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
    if (std::fabs(d - 1) < 1.1920929e-07f) return;
    const double angle = std::acos(d);
    const double v[3] = {b[1] * a[2] - b[2] * a[1], b[2] * a[0] - b[0] * a[2], b[0] * a[1] - b[1] * a[0]};
    double n[3];
    nrm(v, n);
    const double x = n[0], y = n[1], z = n[2], s = std::sin(angle), co = std::cos(angle), m = 1 - co;
    R[0] = m * x * x + co;    R[1] = m * x * y + z * s; R[2] = m * z * x - y * s;
    R[3] = m * x * y - z * s; R[4] = m * y * y + co;    R[5] = m * y * z + x * s;
    R[6] = m * z * x + y * s; R[7] = m * y * z - x * s; R[8] = m * z * z + co;
}

extern "C" int isdf_set_shape(isdf_ctx *c, const isdf_shape *s) {
    if (!c) return ISDF_ERR_INVALID_ARG;
    if (!s || s->kind < 0 || s->kind >= ISDF_SHAPE_KIND_COUNT) return fail(c, ISDF_ERR_INVALID_ARG, "bad shape kind");
    if (s->grad_mode < ISDF_GRAD_DEFAULT || s->grad_mode > ISDF_GRAD_ANALYTIC_BALL) return fail(c, ISDF_ERR_INVALID_ARG, "bad grad_mode");
    if (s->kind == ISDF_SHAPE_GRID) return fail(c, ISDF_ERR_INVALID_ARG, "ISDF_SHAPE_GRID is installed with isdf_set_shape_grid / isdf_set_shape_sampled");
    HIPCHK(c, hipSetDevice(c->device));
    DevShape d{};
    d.kind = s->kind;
    d.grad_mode = s->grad_mode;
    if (d.grad_mode == ISDF_GRAD_DEFAULT)
        d.grad_mode = s->kind == ISDF_SHAPE_BOX ? ISDF_GRAD_BOX_FORWARD : (s->kind == ISDF_SHAPE_BALL ? ISDF_GRAD_ANALYTIC_BALL : ISDF_GRAD_CENTRAL);
    std::memcpy(d.d.p, s->params, sizeof(d.d.p));
    std::memcpy(d.d.trans, s->trans, sizeof(d.d.trans));
    std::memcpy(d.d.rot, s->rotate, sizeof(d.d.rot));
    d.bound_radius = s->bound_radius;
    const double X[3] = {1, 0, 0}, Y[3] = {0, 1, 0}, Z[3] = {0, 0, 1};
    csg_rotate_to(X, Y, d.d.csg_r2);
    csg_rotate_to(X, Z, d.d.csg_r3);
    for (int i = 0; i < 16; i++) d.f.p[i] = (float)d.d.p[i];
    for (int i = 0; i < 3; i++) d.f.trans[i] = (float)d.d.trans[i];
    {
        bool ident = d.d.trans[0] == 0.0 && d.d.trans[1] == 0.0 && d.d.trans[2] == 0.0;
        for (int i = 0; i < 9; i++) ident = ident && d.d.rot[i] == ((i % 4 == 0) ? 1.0 : 0.0);
        d.d.ident = d.f.ident = ident ? 1 : 0;
    }
    for (int i = 0; i < 9; i++) { d.f.rot[i] = (float)d.d.rot[i]; d.f.csg_r2[i] = (float)d.d.csg_r2[i]; d.f.csg_r3[i] = (float)d.d.csg_r3[i]; }
    double bb_c[3] = {s->bbox_center[0], s->bbox_center[1], s->bbox_center[2]}, bb_h[3] = {s->bbox_half[0], s->bbox_half[1], s->bbox_half[2]};
    if (s->kind == ISDF_SHAPE_MESH && s->mesh_vertices && s->n_vertices > 0 && !(bb_h[0] > 0 && bb_h[1] > 0 && bb_h[2] > 0)) {
        // a mesh lies inside the box of its vertices: use it for row pruning when the caller gave none
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        for (int v = 0; v < s->n_vertices; v++)
            for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], s->mesh_vertices[3 * v + a]); hi[a] = std::max(hi[a], s->mesh_vertices[3 * v + a]); }
        for (int a = 0; a < 3; a++) { bb_c[a] = 0.5 * (lo[a] + hi[a]); bb_h[a] = 0.5 * (hi[a] - lo[a]) + 1e-6 * (std::fabs(lo[a]) + std::fabs(hi[a]) + 1.0); }
    }
    d.prune_rows = (bb_h[0] > 0 && bb_h[1] > 0 && bb_h[2] > 0) && !(getenv("ISDF_NO_ROW_PRUNE") && getenv("ISDF_NO_ROW_PRUNE")[0] == '1');
    for (int i = 0; i < 3; i++) { d.bbox_lo[i] = (float)(bb_c[i] - bb_h[i]); d.bbox_hi[i] = (float)(bb_c[i] + bb_h[i]); }
    // the fp32 pre-filter of tile_kernel needs a continuous analytic SDF; ISDF_NO_F32_FILTER=1 disables it (A/B runs)
    d.filter_f32 = (s->kind != ISDF_SHAPE_MESH) && !(getenv("ISDF_NO_F32_FILTER") && getenv("ISDF_NO_F32_FILTER")[0] == '1');
    d.mesh = nullptr;
    d.mesh_wg = 0;
    d.mesh_flat = 0; d.mesh_flat_words = 0; d.mesh_flat_slots = 0;
    d.mesh_levels = isdf::MESH_Q_LEVELS;
    if (s->kind == ISDF_SHAPE_MESH) {
        if (!s->mesh_vertices || !s->mesh_faces || s->n_faces < 1 || s->n_vertices < 3) return fail(c, ISDF_ERR_INVALID_ARG, "mesh shape needs vertices and faces");
        std::vector<double> tri((size_t)9 * s->n_faces);
        std::vector<float> trif((size_t)9 * s->n_faces);
        for (int f = 0; f < s->n_faces; f++)
            for (int k = 0; k < 3; k++) {
                const int vi = s->mesh_faces[3 * f + k];
                if (vi < 0 || vi >= s->n_vertices) return fail(c, ISDF_ERR_INVALID_ARG, "mesh face index out of range");
                for (int a = 0; a < 3; a++) {
                    tri[(size_t)9 * f + 3 * k + a] = s->mesh_vertices[3 * vi + a];
                    trif[(size_t)9 * f + 3 * k + a] = (float)s->mesh_vertices[3 * vi + a];
                }
            }
        // the reference's winding-number hierarchy (igl::fast_winding_number(V, F, 2, fwn_bvh), Shape.cpp:86)
        isdf_host::FwnTree tree;
        isdf_host::fwn_build(s->mesh_vertices, s->n_vertices, s->mesh_faces, s->n_faces, tree);
        if (3 * isdf_host::fwn_depth(tree) + 1 > isdf::MESH_STACK) return fail(c, ISDF_ERR_UNSUPPORTED, "mesh hierarchy too deep for the device traversal stack");
        free_mesh(c);
        HIPCHK(c, hipMalloc(&c->d_fwn_child, tree.child.size() * sizeof(int32_t)));
        HIPCHK(c, hipMalloc(&c->d_fwn_box, tree.box.size() * sizeof(float)));
        HIPCHK(c, hipMemcpy(c->d_fwn_child, tree.child.data(), tree.child.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(c->d_fwn_box, tree.box.data(), tree.box.size() * sizeof(float), hipMemcpyHostToDevice));
        HIPCHK(c, hipMalloc(&c->d_mesh_tri, tri.size() * sizeof(double)));
        HIPCHK(c, hipMalloc(&c->d_mesh_trif, trif.size() * sizeof(float)));
        HIPCHK(c, hipMalloc(&c->d_mesh, sizeof(DevMesh)));
        HIPCHK(c, hipMemcpy(c->d_mesh_tri, tri.data(), tri.size() * sizeof(double), hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(c->d_mesh_trif, trif.data(), trif.size() * sizeof(float), hipMemcpyHostToDevice));
        // child-major copies for the quad-cooperative walks (csrc/dev_mesh.hpp): lane l of a quad reads child l's record, child
        // word and triangle in one round of loads
        const int depth = isdf_host::fwn_depth(tree);
        std::vector<float> boxq((size_t)4 * isdf::MESH_Q_REC * tree.n_nodes(), 0.f);
        std::vector<double> triq((size_t)4 * isdf::MESH_Q_TRI * tree.n_nodes(), 0.0);
        // bounding box of every (node, child) from the fp64 vertices, rounded OUTWARDS to float (the closest-point walk's bound);
        // children have higher node numbers than their parents in this layout or not - a memoised recursion does not care
        std::vector<double> aabb((size_t)4 * 6 * tree.n_nodes());
        std::vector<char> aabb_done((size_t)tree.n_nodes(), 0);
        std::function<void(int)> node_boxes = [&](int nd) {
            if (aabb_done[nd]) return;
            aabb_done[nd] = 1;
            for (int ch = 0; ch < 4; ch++) {
                double *bb = aabb.data() + ((size_t)4 * nd + ch) * 6;
                for (int a = 0; a < 3; a++) { bb[a] = 1.0e300; bb[3 + a] = -1.0e300; }
                const int32_t ci = tree.child[(size_t)4 * nd + ch];
                if (ci == -1) continue;
                if (ci >= 0) {
                    for (int k = 0; k < 3; k++) for (int a = 0; a < 3; a++) { const double v = tri[(size_t)9 * ci + 3 * k + a]; bb[a] = std::min(bb[a], v); bb[3 + a] = std::max(bb[3 + a], v); }
                } else {
                    const int sub = ci & 0x7fffffff;
                    node_boxes(sub);
                    for (int c2 = 0; c2 < 4; c2++) {
                        const double *sb = aabb.data() + ((size_t)4 * sub + c2) * 6;
                        for (int a = 0; a < 3; a++) { bb[a] = std::min(bb[a], sb[a]); bb[3 + a] = std::max(bb[3 + a], sb[3 + a]); }
                    }
                }
            }
        };
        node_boxes(0);
        auto f_down = [](double v) { float f = (float)v; if ((double)f > v) f = std::nextafterf(f, -INFINITY); return f; };
        auto f_up = [](double v) { float f = (float)v; if ((double)f < v) f = std::nextafterf(f, INFINITY); return f; };
        for (int nd = 0; nd < tree.n_nodes(); nd++)
            for (int ch = 0; ch < 4; ch++) {
                float *rq = boxq.data() + ((size_t)4 * nd + ch) * isdf::MESH_Q_REC;
                {
                    const double *bb = aabb.data() + ((size_t)4 * nd + ch) * 6;
                    const bool any = bb[0] <= bb[3];
                    for (int a = 0; a < 3; a++) { rq[34 + a] = any ? f_down(bb[a]) : 3.0e38f; rq[37 + a] = any ? f_up(bb[3 + a]) : -3.0e38f; }
                }
                for (int k = 0; k < 23; k++) rq[k] = tree.box[(size_t)92 * nd + 4 * k + ch];
                const int32_t ci = tree.child[(size_t)4 * nd + ch];
                std::memcpy(&rq[23], &ci, 4);
                if (ci >= 0) {
                    for (int k = 0; k < 9; k++) rq[24 + k] = trif[(size_t)9 * ci + k];
                    for (int k = 0; k < 9; k++) triq[((size_t)4 * nd + ch) * isdf::MESH_Q_TRI + k] = tri[(size_t)9 * ci + k];
                }
            }
        HIPCHK(c, hipMalloc(&c->d_fwn_boxq, boxq.size() * sizeof(float)));
        HIPCHK(c, hipMemcpy(c->d_fwn_boxq, boxq.data(), boxq.size() * sizeof(float), hipMemcpyHostToDevice));
        HIPCHK(c, hipMalloc(&c->d_fwn_triq, triq.size() * sizeof(double)));
        HIPCHK(c, hipMemcpy(c->d_fwn_triq, triq.data(), triq.size() * sizeof(double), hipMemcpyHostToDevice));
        DevMesh hm{c->d_mesh_tri, c->d_mesh_trif, s->n_faces, c->d_fwn_child, c->d_fwn_box, tree.n_nodes(), c->d_fwn_boxq, c->d_fwn_triq, depth, nullptr, {0, 0, 0}, {0.f, 0.f, 0.f}, 0.f, 0.f, 0.f, 0.f,
                   nullptr, 0, 0, 0, 0, 0, 0};
        // The flat slot table of a (sub)tree rooted at `root` (DevMesh::flat / DevMesh::sub) - levels by breadth-first search from it.
        // ONE self-describing blob, copied into LDS as it is: a header of 24 ints ([0..8] the slot index where level l begins, [9..17]
        // the same for the combine steps over the nodes, deepest level first, [18] triangles, [19] slots, [20] levels, [21] the blob's
        // size in 4-byte words, [22] / [23] where the records / the fp64 triangles begin), per slot 4 ints (record = 4 node + child,
        // parent slot or -1, triangle index or -1, level), per triangle its slot, per node (deepest first) 5 ints (own slot or -1 for the
        // root, its four child slots or -1), then per slot its fwn_boxq record (MESH_Q_REC floats) and its fwn_triq triangle
        // (MESH_Q_TRI doubles).  Empty: the subtree does not qualify (more than max_slots slots, or too deep).
        const int nn_all = tree.n_nodes();
        auto build_blob = [&](const int root, const int max_slots) -> std::vector<int> {
            std::vector<int> node_level((size_t)nn_all, -1), node_slot((size_t)nn_all, -1), order;
            node_level[root] = 0; order.push_back(root);
            int n_slots = 0;
            for (size_t h = 0; h < order.size(); h++) {
                const int nd = order[h];
                for (int ch = 0; ch < 4; ch++) {
                    const int32_t ci = tree.child[(size_t)4 * nd + ch];
                    if (ci == -1) continue;
                    if (++n_slots > max_slots) return {};
                    if (ci < 0) { const int sub = ci & 0x7fffffff; if (node_level[sub] < 0) { node_level[sub] = node_level[nd] + 1; order.push_back(sub); } }
                }
            }
            const int nn = (int)order.size();
            // slots in (level of their node, node in search order, child) order
            std::vector<int> slots, tris, nodes, lvl_begin(9, 0), step_begin(9, 0);
            int cur_level = -1;
            for (int nd : order) {
                if (node_level[nd] != cur_level) { cur_level = node_level[nd]; if (cur_level >= isdf::MESH_FLAT_LEVELS) return {}; lvl_begin[cur_level] = (int)slots.size() / 4; }
                for (int ch = 0; ch < 4; ch++) {
                    const int32_t ci = tree.child[(size_t)4 * nd + ch];
                    if (ci == -1) continue;
                    const int sl = (int)slots.size() / 4;
                    if (ci < 0) node_slot[ci & 0x7fffffff] = sl;
                    else tris.push_back(sl);
                    slots.push_back(4 * nd + ch); slots.push_back(node_slot[nd]); slots.push_back(ci >= 0 ? ci : -1); slots.push_back(node_level[nd]);
                }
            }
            const int n_levels = cur_level + 1;
            for (int l = n_levels; l < 9; l++) lvl_begin[l] = n_slots;
            // combine steps: the nodes of the deepest level first; per node its own slot and its four child slots
            std::map<int, std::array<int, 4>> child_slots;
            for (int nd : order) child_slots[nd] = {-1, -1, -1, -1};
            for (int sl = 0; sl < n_slots; sl++) { const int rec = slots[(size_t)4 * sl]; child_slots[rec >> 2][rec & 3] = sl; }
            int step = 0;
            for (int l = n_levels - 1; l >= 0; l--, step++) {
                step_begin[step] = (int)nodes.size() / 5;
                for (int nd : order) if (node_level[nd] == l) { nodes.push_back(node_slot[nd]); for (int ch = 0; ch < 4; ch++) nodes.push_back(child_slots[nd][ch]); }
            }
            for (int st = step; st < 9; st++) step_begin[st] = nn;
            std::vector<int> flat;
            flat.insert(flat.end(), lvl_begin.begin(), lvl_begin.end());
            flat.insert(flat.end(), step_begin.begin(), step_begin.end());
            flat.push_back((int)tris.size()); flat.push_back(n_slots); flat.push_back(n_levels); flat.push_back(0); flat.push_back(0); flat.push_back(0);
            flat.insert(flat.end(), slots.begin(), slots.end());
            flat.insert(flat.end(), tris.begin(), tris.end());
            flat.insert(flat.end(), nodes.begin(), nodes.end());
            while (flat.size() % 4) flat.push_back(0);
            const int rec_off = (int)flat.size();                 // (in 4-byte units)
            flat.resize(flat.size() + (size_t)n_slots * isdf::MESH_Q_REC);
            for (int sl = 0; sl < n_slots; sl++)
                std::memcpy(flat.data() + rec_off + (size_t)sl * isdf::MESH_Q_REC, boxq.data() + (size_t)slots[(size_t)4 * sl] * isdf::MESH_Q_REC, isdf::MESH_Q_REC * sizeof(float));
            const int trec_off = (int)flat.size();
            flat.resize(flat.size() + (size_t)n_slots * isdf::MESH_Q_TRI * 2);
            for (int sl = 0; sl < n_slots; sl++)
                std::memcpy(flat.data() + trec_off + (size_t)sl * isdf::MESH_Q_TRI * 2, triq.data() + (size_t)slots[(size_t)4 * sl] * isdf::MESH_Q_TRI, isdf::MESH_Q_TRI * sizeof(double));
            while (flat.size() % 4) flat.push_back(0);
            flat[21] = (int)flat.size(); flat[22] = rec_off; flat[23] = trec_off;
            return flat;
        };
        // Small meshes: the WHOLE hierarchy as one table
        {
            const std::vector<int> flat = build_blob(0, isdf::MESH_FLAT_SLOTS);
            if (!flat.empty()) {
                HIPCHK(c, hipMalloc(&c->d_mesh_flat, flat.size() * sizeof(int)));
                HIPCHK(c, hipMemcpy(c->d_mesh_flat, flat.data(), flat.size() * sizeof(int), hipMemcpyHostToDevice));
                hm.flat_words = (int)flat.size(); hm.flat_rec = flat[22]; hm.flat_trec = flat[23];
                hm.flat = c->d_mesh_flat; hm.flat_slots = flat[19]; hm.flat_nodes = nn_all; hm.flat_levels = flat[20];
            }
        }
        // Which form the swept-volume sweep takes: the FLAT evaluation for small meshes (<= 64 slots: the reference's 12- to 20-face
        // robots), else one task per workgroup with the quad-cooperative walks (round 6, C5 shape: drone.obj, 52 faces, 1.50 ms against
        // 1.69 with the wave-cooperative walks, kuang.obj, 60 faces, 1.96 / 2.26, box.obj, 96 faces, 3.6 / 4.7); the wave-cooperative
        // walks keep the hierarchies deeper than MESH_Q_LEVELS.  ISDF_MESH_WG=0/1, ISDF_MESH_FLAT=0 force.
        d.mesh_levels = std::max(2, std::min(depth, isdf::MESH_Q_LEVELS));
        c->mesh_depth = depth;
        d.mesh_wg = depth <= isdf::MESH_Q_LEVELS ? 1 : 0;
        if (const char *e = getenv("ISDF_MESH_WG")) d.mesh_wg = (e[0] == '1' && depth <= isdf::MESH_Q_LEVELS) ? 1 : 0;
        // small meshes: the flat evaluation (one task per workgroup as well); ISDF_MESH_FLAT=0 keeps the walks (A/B runs, tests)
        d.mesh_flat = hm.flat ? 1 : 0;
        d.mesh_flat_words = hm.flat_words; d.mesh_flat_slots = hm.flat_slots;
        if (const char *e = getenv("ISDF_MESH_FLAT")) if (e[0] == '0') d.mesh_flat = 0;
        HIPCHK(c, hipMemcpy(c->d_mesh, &hm, sizeof(hm), hipMemcpyHostToDevice));
        d.mesh = c->d_mesh;
        // The tile sweep's pre-filter for this kind: a lattice of distances over the box the row pruning lets voxels come from (the
        // shape's box inflated by the penalty band), 96 cells along its longest side.  A voxel whose nearest node is farther from the
        // surface than the widened band + the node spacing, on the outside, cannot carry a penalty (DevMesh::dl) - most of the listed
        // voxels of a mesh robot, each of which would cost a hierarchy query.  ISDF_NO_F32_FILTER=1 leaves it out.
        // (closed meshes only: next to an OPEN surface the winding number is a fraction, and (1 - 2 w) * distance says nothing
        // about the distance)
        bool closed = true;
        {
            std::vector<std::pair<long long, int>> edges;
            edges.reserve((size_t)3 * s->n_faces);
            for (int f = 0; f < s->n_faces; f++)
                for (int k = 0; k < 3; k++) {
                    const long long a = s->mesh_faces[3 * f + k], b = s->mesh_faces[3 * f + (k + 1) % 3];
                    edges.emplace_back(std::min(a, b) * (long long)s->n_vertices + std::max(a, b), a < b ? 1 : -1);
                }
            std::sort(edges.begin(), edges.end());
            for (size_t i = 0; i < edges.size() && closed; i += 2)       // every edge twice, once in each direction
                closed = i + 1 < edges.size() && edges[i].first == edges[i + 1].first && edges[i].second + edges[i + 1].second == 0 &&
                         (i + 2 >= edges.size() || edges[i + 2].first != edges[i].first);
        }
        // ... and the surface must bound a solid: exact winding number 0 / 1 on the two sides of every face (shape_eval.hip:
        // nested sheets, overlapping or inverted components and tears show up there whatever their thickness)
        int solid = 0;
        float s_range[2] = {0.f, 0.f};
        // (the edge pairing above goes by vertex INDEX: a triangle soup - the reference's Lthick.obj, box.obj, kuang.obj, drone.obj keep
        // three vertices of their own per face - is "open" to it whatever its geometry.  The exact test below is geometric and catches
        // open surfaces too (next to a boundary the winding number is a fraction), so it alone decides; `closed` is reported.)
        const bool want_lattice = d.prune_rows && !(getenv("ISDF_NO_F32_FILTER") && getenv("ISDF_NO_F32_FILTER")[0] == '1');
        float defect[2] = {0.f, 0.f};
        double llo[3] = {0, 0, 0}, lhi[3] = {0, 0, 0}, lat_ext = 0.0;
        int cells = 96;
        if (want_lattice) {
            // (the lattice's box: wide enough for the swept-volume scans' band, 2 safety_hor + 0.1, as well: mesh_lattice_not_below)
            const double margin = 1.05 * std::max(c->cfg.safety_hor, 2.0 * c->cfg.safety_hor + 0.1) + 0.01;
            for (int a = 0; a < 3; a++) { llo[a] = (double)d.bbox_lo[a] - margin; lhi[a] = (double)d.bbox_hi[a] + margin; lat_ext = std::max(lat_ext, lhi[a] - llo[a]); }
            if (const char *e = getenv("ISDF_MESH_LATTICE_CELLS")) { const int v = atoi(e); if (v >= 16 && v <= 256) cells = v; }      // (developer switch)
            double ext3 = 0.0;
            for (int a = 0; a < 3; a++) ext3 = std::max(ext3, 2.0 * bb_h[a]);
            // a defect pocket may be a tenth of the lattice's reach (half a cell diagonal) thick; anything thicker is a region
            const double tau_limit = 0.1 * 0.5 * 1.7320508 * (lat_ext / cells);
            const int rcv = isdf_mesh_surface_valid(c, c->d_mesh_tri, s->n_faces, ext3, tau_limit, &solid, defect);
            if (rcv) return rcv;
        }
        if (want_lattice && solid) {
            const int rc = isdf_mesh_lattice_build(c, &hm, llo, lhi, cells, s_range);
            if (rc) return rc;
            if (hm.dl) {
                hm.dl_tau = defect[0]; hm.dl_slack = 1.05f * defect[1] * defect[0];
                HIPCHK(c, hipMemcpy(c->d_mesh, &hm, sizeof(hm), hipMemcpyHostToDevice)); d.filter_f32 = 1;
            }
        }
        {   // isdf_mesh_info
            int *mi = c->mesh_info;
            mi[0] = s->n_faces; mi[1] = tree.n_nodes(); mi[2] = depth; mi[3] = d.mesh_wg; mi[4] = closed ? 1 : 0; mi[5] = want_lattice ? solid : -1;
            mi[6] = hm.dl ? hm.dln[0] : 0; mi[7] = hm.dl ? hm.dln[1] : 0; mi[8] = hm.dl ? hm.dln[2] : 0;
            mi[9] = (int)std::lround(1.0e6 * s_range[0]); mi[10] = (int)std::lround(1.0e6 * s_range[1]); mi[11] = d.mesh_flat ? hm.flat_slots : 0;
            mi[12] = (int)std::lround(1.0e9 * defect[0]); mi[13] = (int)std::lround(1.0e3 * defect[1]); mi[14] = mi[15] = 0;
        }
    }
    if (s->kind != ISDF_SHAPE_MESH) std::memset(c->mesh_info, 0, sizeof(c->mesh_info));
    isdf_frontend_release(c);       // the attitude kernels were voxelised from the previous shape
    c->shape = d;
    c->shape_host = *s;
    c->shape_host.mesh_vertices = nullptr;
    c->shape_host.mesh_faces = nullptr;
    c->have_shape = true;
    ISDF_REPLICATE(c, isdf_set_shape(p_, s));
    return ISDF_OK;
}

// A shape the library has no formula for, as the lattice BasicShape::initShape tabulates (Shape.hpp:361-404); sampled on the
// device like getonlySDFNum / getonlyGrad1Num / getSDFwithGrad1Num (:481-600).
extern "C" int isdf_set_shape_grid(isdf_ctx *c, const double *cells, int nx, int ny, int nz, const double grid_min[3], double nres,
                                   double bound_radius, const double *bbox_center, const double *bbox_half) {
    if (!c) return ISDF_ERR_INVALID_ARG;
    if (!cells || !grid_min || nx < 2 || ny < 2 || nz < 2 || !(nres > 0)) return fail(c, ISDF_ERR_INVALID_ARG, "bad lattice (at least 2 nodes per axis)");
    if ((double)nx * ny * nz > 2.0e8) return fail(c, ISDF_ERR_INVALID_ARG, "lattice too large");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t n = (size_t)nx * ny * nz;
    free_mesh(c);
    if (c->d_shape_grid) { (void)hipFree(c->d_shape_grid); c->d_shape_grid = nullptr; }
    HIPCHK(c, hipMalloc(&c->d_shape_grid, n * 4 * sizeof(double)));
    HIPCHK(c, hipMemcpy(c->d_shape_grid, cells, n * 4 * sizeof(double), hipMemcpyHostToDevice));
    DevShape d{};
    d.kind = ISDF_SHAPE_GRID; d.grad_mode = ISDF_GRAD_GRID;
    d.d.ident = d.f.ident = 1;
    for (int i = 0; i < 9; i++) { d.d.rot[i] = (i % 4 == 0) ? 1.0 : 0.0; d.f.rot[i] = (float)d.d.rot[i]; }
    d.grid = c->d_shape_grid; d.gn[0] = nx; d.gn[1] = ny; d.gn[2] = nz; d.gres = nres;
    for (int a = 0; a < 3; a++) d.gmin[a] = grid_min[a];
    // outside the lattice the sampler returns 1e20: the lattice box bounds everything that can carry a penalty
    const int dims[3] = {nx, ny, nz};
    double bb_c[3], bb_h[3], r2 = 0.0;
    for (int a = 0; a < 3; a++) {
        const double lo = grid_min[a], hi = grid_min[a] + (dims[a] - 1) * nres;
        bb_c[a] = 0.5 * (lo + hi); bb_h[a] = 0.5 * (hi - lo) + 1e-9;
        const double far = std::max(std::fabs(lo), std::fabs(hi));
        r2 += far * far;
    }
    if (bbox_center && bbox_half && bbox_half[0] > 0 && bbox_half[1] > 0 && bbox_half[2] > 0)
        for (int a = 0; a < 3; a++) { bb_c[a] = bbox_center[a]; bb_h[a] = bbox_half[a]; }
    d.bound_radius = bound_radius > 0 ? bound_radius : std::sqrt(r2);
    d.prune_rows = !(getenv("ISDF_NO_ROW_PRUNE") && getenv("ISDF_NO_ROW_PRUNE")[0] == '1');
    for (int a = 0; a < 3; a++) { d.bbox_lo[a] = (float)(bb_c[a] - bb_h[a]); d.bbox_hi[a] = (float)(bb_c[a] + bb_h[a]); }
    d.filter_f32 = 0;                      // no fp32 formula to pre-filter with
    d.mesh = nullptr;
    isdf_frontend_release(c);
    c->shape = d;
    c->shape_host = isdf_shape{};
    c->shape_host.kind = ISDF_SHAPE_GRID; c->shape_host.grad_mode = ISDF_GRAD_GRID;
    c->shape_host.bound_radius = d.bound_radius;
    for (int a = 0; a < 3; a++) { c->shape_host.bbox_center[a] = bb_c[a]; c->shape_host.bbox_half[a] = bb_h[a]; }
    c->have_shape = true;
    ISDF_REPLICATE(c, isdf_set_shape_grid(p_, cells, nx, ny, nz, grid_min, nres, bound_radius, bbox_center, bbox_half));
    return ISDF_OK;
}
extern "C" int isdf_set_shape_sampled(isdf_ctx *c, isdf_sdf_with_grad_fn fn, void *user, double ndx, double ndy, double ndz, double nres,
                                      double bound_radius, const double *bbox_center, const double *bbox_half) {
    if (!c) return ISDF_ERR_INVALID_ARG;
    if (!fn || !(ndx > 0) || !(ndy > 0) || !(ndz > 0) || !(nres > 0)) return fail(c, ISDF_ERR_INVALID_ARG, "bad sampling arguments");
    // initShape (Shape.hpp:368-376): sizes ceil(nd / nres), the lattice starts at -nd / 2
    const int X = (int)std::ceil(ndx / nres), Y = (int)std::ceil(ndy / nres), Z = (int)std::ceil(ndz / nres);
    if (X < 2 || Y < 2 || Z < 2 || (double)X * Y * Z > 2.0e8) return fail(c, ISDF_ERR_INVALID_ARG, "bad lattice size");
    const double mn[3] = {-ndx / 2, -ndy / 2, -ndz / 2};
    std::vector<double> cells((size_t)X * Y * Z * 4);
    for (int i = 0; i < X; i++)
        for (int j = 0; j < Y; j++)
            for (int k = 0; k < Z; k++) {
                const double p[3] = {mn[0] + i * nres, mn[1] + j * nres, mn[2] + k * nres};       // :390
                double g[3] = {0, 0, 0};
                const double dis = fn(user, p, g);                                                   // getSDFwithGrad1(p_rel, grad) :391
                double *o = cells.data() + 4 * (((size_t)i * Y + j) * Z + k);
                o[0] = g[0]; o[1] = g[1]; o[2] = g[2]; o[3] = dis;
            }
    return isdf_set_shape_grid(c, cells.data(), X, Y, Z, mn, nres, bound_radius, bbox_center, bbox_half);
}

extern "C" int isdf_set_points(isdf_ctx *c, const double *xyz, int M) {
    if (!c) return ISDF_ERR_INVALID_ARG;
    if (M < 0 || (M > 0 && !xyz)) return fail(c, ISDF_ERR_INVALID_ARG, "bad points");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->d_points) { (void)hipFree(c->d_points); c->d_points = nullptr; }
    if (c->d_tstar) { (void)hipFree(c->d_tstar); c->d_tstar = nullptr; }
    c->M = M;
    c->points_epoch++;
    if (c->d_v1_words) HIPCHK(c, hipMemset(c->d_v1_words, 0, 8 * sizeof(unsigned)));
    if (M > 0) {
        HIPCHK(c, hipMalloc(&c->d_points, (size_t)3 * M * sizeof(double)));
        HIPCHK(c, hipMemcpy(c->d_points, xyz, (size_t)3 * M * sizeof(double), hipMemcpyHostToDevice));
        HIPCHK(c, hipMalloc(&c->d_tstar, (size_t)M * sizeof(double)));
        HIPCHK(c, hipMemset(c->d_tstar, 0, (size_t)M * sizeof(double)));   // lastTstar starts at 0 (plan_manager.cpp:254)
    }
    ISDF_REPLICATE(c, isdf_set_points(p_, xyz, M));
    return ISDF_OK;
}

extern "C" int isdf_set_shard(isdf_ctx *c, int rank, int world) {
    if (!c) return ISDF_ERR_INVALID_ARG;
    if (world < 1 || rank < 0 || rank >= world) return fail(c, ISDF_ERR_INVALID_ARG, "bad shard");
    if (!c->peers.empty() || c->is_peer) return fail(c, ISDF_ERR_STATE, "a multi-device ctx (isdf_create_multi) shards by itself");
    c->rank = rank; c->world = world;
    return ISDF_OK;
}

// --------------------------------------------------------------------------------------------------------------
// per-step evaluation
// --------------------------------------------------------------------------------------------------------------
static void shard_range(long long total, int rank, int world, long long &b, long long &e) {
    const long long q = total / world, r = total % world;
    b = rank * q + (rank < r ? rank : r);
    e = b + q + (rank < r ? 1 : 0);
}

int isdf_ensure_doubles(isdf_ctx *c, double **p, size_t *cap, size_t need) {
    if (*cap >= need) return ISDF_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr; *cap = 0;
    HIPCHK(c, hipMalloc(p, need * sizeof(double)));
    *cap = need;
    return ISDF_OK;
}

static int ensure(isdf_ctx *c, double **p, size_t *cap, size_t need) { return isdf_ensure_doubles(c, p, cap, need); }

static void fill_flat(const isdf_config &cfg, FlatP &f) {
    f.mass = cfg.vehicle_mass; f.grav = cfg.grav_acc; f.dh = cfg.horiz_drag; f.dv = cfg.vert_drag;
    f.cp = cfg.paras_drag; f.veps = cfg.speed_eps; f.dh_over_m = f.dh / f.mass;
}

static int prof_begin(isdf_ctx *c, hipStream_t st, ProfEvent **ev) {
    *ev = nullptr;
    if (!c->prof_on) return ISDF_OK;
    if ((c->prof_tick++ % c->prof_every) != 0) return ISDF_OK;
    if (c->prof_used == c->prof_events.size()) {
        ProfEvent p;
        HIPCHK(c, hipEventCreate(&p.a));
        HIPCHK(c, hipEventCreate(&p.b));
        HIPCHK(c, hipEventCreate(&p.c));
        HIPCHK(c, hipEventCreate(&p.d));
        c->prof_events.push_back(p);
    }
    *ev = &c->prof_events[c->prof_used++];
    return ISDF_OK;
}

// scratch of the integral sweep for `total_pieces` pieces per launch (grows only; the batched optimizer reserves its largest
// round up front so that no round reallocates while another is in flight)
int isdf_reserve_sweep_buffers(isdf_ctx *c, long long total_pieces) {
    const size_t n_samples = (size_t)total_pieces * (c->cfg.integral_intervs + 1);
    if (c->sample_cap < n_samples) {
        if (c->d_acc) (void)hipFree(c->d_acc);
        if (c->d_sample_info) (void)hipFree(c->d_sample_info);
        c->d_acc = nullptr; c->d_sample_info = nullptr; c->sample_cap = 0;
        HIPCHK(c, hipMalloc(&c->d_acc, n_samples * ACC_STRIDE * sizeof(double)));
        HIPCHK(c, hipMalloc(&c->d_sample_info, n_samples * 2 * sizeof(int)));
        // result slots are created EMPTY (all-ones, tile_sweep.hip SLOT_EMPTY); every step leaves them empty again
        HIPCHK(c, hipMemset(c->d_acc, 0xFF, n_samples * ACC_STRIDE * sizeof(double)));
        HIPCHK(c, hipDeviceSynchronize());
        c->sample_cap = n_samples;
    }
    if (c->piece_cap < (size_t)total_pieces) {
        // piece-cost slots: created EMPTY like the collision sums' slots
        if (c->d_piece_cost) (void)hipFree(c->d_piece_cost);
        c->d_piece_cost = nullptr; c->piece_cap = 0;
        HIPCHK(c, hipMalloc(&c->d_piece_cost, (size_t)total_pieces * sizeof(double)));
        HIPCHK(c, hipMemset(c->d_piece_cost, 0xFF, (size_t)total_pieces * sizeof(double)));
        HIPCHK(c, hipDeviceSynchronize());
        c->piece_cap = (size_t)total_pieces;
    }
    return ISDF_OK;
}

// The result slots empty themselves only when their single consumer takes them.  Once a bounded wait has expired (overflow
// word, NaN cost) a producer may still publish late into a slot nobody takes any more, and the NEXT step would consume that
// stale value without any flag being raised.  Whoever sees the overflow word calls this: drain, then every slot empty again.
int isdf_reset_result_slots(isdf_ctx *c) {
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipDeviceSynchronize());
    if (c->d_acc) HIPCHK(c, hipMemset(c->d_acc, 0xFF, c->sample_cap * ACC_STRIDE * sizeof(double)));
    if (c->d_piece_cost) HIPCHK(c, hipMemset(c->d_piece_cost, 0xFF, c->piece_cap * sizeof(double)));
    isdf_xchg_reset_board(c);
    HIPCHK(c, hipDeviceSynchronize());
    for (isdf_ctx *p : c->peers) {                  // a multi-device ctx: the peers' sticky overflow words and slots as well
        const int rc = isdf_reset_result_slots(p);
        if (rc) { c->err = p->err; return rc; }
        HIPCHK(c, hipMemset(p->d_stats + 4, 0, sizeof(unsigned long long)));
    }
    if (!c->peers.empty()) HIPCHK(c, hipSetDevice(c->device));
    return ISDF_OK;
}

// mode 0: the sweep cfg.variant names; 1: the swept-volume sweep; 2: the integral sweep with the collision term off
// (modes 1 + 2 together are what costFunctionLmbm runs for the reference's live configuration)
// host-direct step (see isdf_ctx.hpp): device-visible addresses of the pinned inputs / outputs / flags of this step
struct HostDirect { const double *T, *coeffs; double *out; unsigned long long *flags; unsigned long long seq; bool via_bar; };
static int ensure_stage(isdf_ctx *c, size_t total_pieces) {
    const size_t n_groups = (total_pieces + STAGE_G - 1) / STAGE_G;
    { int rc1 = isdf_ensure_doubles(c, &c->d_stage, &c->stage_cap, total_pieces * 19); if (rc1) return rc1; }
    if (c->stage_flags_cap < n_groups) {
        if (c->d_stage_flags) (void)hipFree(c->d_stage_flags);
        c->d_stage_flags = nullptr; c->stage_flags_cap = 0;
        HIPCHK(c, hipMalloc(&c->d_stage_flags, n_groups * sizeof(unsigned long long)));
        HIPCHK(c, hipMemset(c->d_stage_flags, 0, n_groups * sizeof(unsigned long long)));
        c->stage_flags_cap = n_groups;
    }
    return ISDF_OK;
}
constexpr int ISDF_DIRECT_NA = 1;        // eval_device_impl: the step cannot run host-direct (nothing was launched)

static int eval_device_impl(isdf_ctx *c, int n_traj, int N, const double *d_T, const double *d_coeffs, double *d_out,
                            double *d_tstar, hipStream_t st, int mode = 0, bool fixed_tstar = false, const HostDirect *hd = nullptr) {
    isdf_config cfg = c->cfg;
    if (mode == 1) cfg.variant = ISDF_V1_SWEPT;
    if (mode == 2) { cfg.variant = ISDF_V3_ESDF_TILE; cfg.enable_pos = 0; }
    c->stats_cached = false;
    if (n_traj < 1 || N < 1) return fail(c, ISDF_ERR_INVALID_ARG, "n_traj and N must be >= 1");
    if (!d_T || !d_coeffs || !d_out) return fail(c, ISDF_ERR_INVALID_ARG, "null device buffer");
    if (!c->have_shape && (cfg.variant == ISDF_V1_SWEPT || cfg.enable_pos)) return fail(c, ISDF_ERR_STATE, "shape not set");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t ostride = isdf_out_stride(N);

    if (cfg.variant == ISDF_V1_SWEPT) {
        if (hd) return ISDF_DIRECT_NA;
        if (n_traj != 1) return fail(c, ISDF_ERR_UNSUPPORTED, "the swept-volume sweep takes one trajectory");
        if (isdf_xchg_fuse_on(c)) return fail(c, ISDF_ERR_UNSUPPORTED, "the in-kernel exchange belongs to the integral sweep (V2/V3)");
        if (c->M <= 0) {                 // no obstacle points: nothing to add
            HIPCHK(c, hipMemsetAsync(d_out, 0, (size_t)n_traj * ostride * sizeof(double), st));
            HIPCHK(c, hipMemsetAsync(c->d_stats, 0, 8 * sizeof(unsigned long long), st));
            return ISDF_OK;
        }
        // (no clearing here: the prepare / fixed kernel zeroes the statistics words and the reduction writes every output)
        long long b, e;
        shard_range(c->M, c->rank, c->world, b, e);
        SweptParams P{};
        P.shape = c->shape;
        fill_flat(cfg, P.flat);
        P.N = N; P.M = c->M; P.point_begin = (int)b; P.point_end = (int)e;
        P.safety_hor = cfg.safety_hor; P.weight_p = cfg.weight_p;
        P.T = d_T; P.coeffs = d_coeffs; P.points = c->d_points;
        P.tstar = (d_tstar && !fixed_tstar) ? d_tstar : c->d_tstar;
        P.tstar_stage = fixed_tstar ? nullptr : c->v1_tstar_stage;
        c->v1_tstar_stage = nullptr;
        P.traj_duration = c->d_traj_duration;
        P.max_coarse = 1536;   // traj_duration < 300 s, coarse step 0.2 s -> at most 1501 samples
        if (!c->d_coarse_t) {
            HIPCHK(c, hipMalloc(&c->d_coarse_t, (size_t)P.max_coarse * sizeof(double)));
            HIPCHK(c, hipMalloc(&c->d_coarse_pose, (size_t)P.max_coarse * 12 * sizeof(double)));
        }
        if (c->point_cap < c->M) {
            if (c->d_point_partial) (void)hipFree(c->d_point_partial);
            if (c->d_point_piece) (void)hipFree(c->d_point_piece);
            if (c->d_point_stat) (void)hipFree(c->d_point_stat);
            if (c->d_point_nr) (void)hipFree(c->d_point_nr);
            if (c->d_task_buf) (void)hipFree(c->d_task_buf);
            if (c->d_task_map) (void)hipFree(c->d_task_map);
            if (c->d_point_lmask) (void)hipFree(c->d_point_lmask);
            c->d_point_lmask = nullptr;
            if (c->d_scan_ticks) (void)hipFree(c->d_scan_ticks);
            if (c->d_scan_order) (void)hipFree(c->d_scan_order);
            c->d_scan_ticks = nullptr; c->d_scan_order = nullptr;
            c->d_point_partial = nullptr; c->d_point_piece = nullptr; c->d_point_stat = nullptr; c->point_cap = 0;
            c->d_point_nr = nullptr; c->d_task_buf = nullptr; c->d_task_map = nullptr;
            HIPCHK(c, hipMalloc(&c->d_point_partial, (size_t)c->M * PARTIAL_STRIDE * sizeof(double)));
            HIPCHK(c, hipMalloc(&c->d_point_piece, (size_t)c->M * sizeof(int)));
            HIPCHK(c, hipMalloc(&c->d_point_stat, (size_t)c->M * sizeof(unsigned long long)));
            HIPCHK(c, hipMalloc(&c->d_point_nr, (size_t)c->M * sizeof(int)));
            HIPCHK(c, hipMemset(c->d_point_nr, 0, (size_t)c->M * sizeof(int)));          // "scan passes last step": none yet
            HIPCHK(c, hipMalloc(&c->d_task_buf, (size_t)c->M * 32 * 6 * sizeof(double)));       // SW_MAX_RANGES x TASK_STRIDE per point
            HIPCHK(c, hipMalloc(&c->d_task_map, (size_t)c->M * 32 * sizeof(unsigned)));
            HIPCHK(c, hipMalloc(&c->d_point_lmask, (size_t)c->M * sizeof(unsigned)));
            HIPCHK(c, hipMemset(c->d_point_lmask, 0, (size_t)c->M * sizeof(unsigned)));
            HIPCHK(c, hipMalloc(&c->d_scan_ticks, (size_t)c->M * sizeof(unsigned)));
            HIPCHK(c, hipMemset(c->d_scan_ticks, 0, (size_t)c->M * sizeof(unsigned)));
            HIPCHK(c, hipMalloc(&c->d_scan_order, (size_t)c->M * sizeof(int)));
            if (!c->d_v1_words) HIPCHK(c, hipMalloc(&c->d_v1_words, 32 * sizeof(unsigned)));
            HIPCHK(c, hipMemset(c->d_v1_words, 0, 32 * sizeof(unsigned)));
            c->scan_order_b = c->scan_order_e = -1;
            c->point_cap = c->M;
        }
        P.coarse_t = c->d_coarse_t; P.coarse_pose = c->d_coarse_pose; P.n_coarse = c->d_n_coarse;
        P.point_partial = c->d_point_partial; P.point_piece = c->d_point_piece; P.point_stat = c->d_point_stat;
        P.point_nr = c->d_point_nr; P.task_buf = c->d_task_buf; P.task_map = c->d_task_map; P.words = c->d_v1_words; P.point_lmask = c->d_point_lmask;
        // The scan's dispatch order, longest first.  Mesh robots: sorted in this step's prepare kernel from last step's durations.  Analytic
        // robots: written by step k's back-prop kernel for step k + 1 - valid only for the same shard of the same points.
        const bool scan_lpt = !c->env_no_lpt && !fixed_tstar;
        const bool sort_here = c->shape.kind == ISDF_SHAPE_MESH;
        P.scan_ticks = scan_lpt ? c->d_scan_ticks : nullptr; P.scan_order_out = scan_lpt ? c->d_scan_order : nullptr; P.scan_sort_here = (scan_lpt && sort_here) ? 1 : 0;
        const bool order_valid = sort_here || (c->scan_order_b == (long long)b && c->scan_order_e == (long long)e && c->scan_order_epoch == c->points_epoch);
        P.scan_order = (scan_lpt && order_valid) ? c->d_scan_order : nullptr;
        // (what d_scan_order will hold once this step's launches are QUEUED; until then it counts as unwritten - a failure in between must
        // not leave the next step scanning through an order nobody wrote)
        const long long order_b = sort_here ? -1 : b, order_e = sort_here ? -1 : e;
        if (scan_lpt) { c->scan_order_b = c->scan_order_e = -1; }
        P.direct_records = fixed_tstar ? 1 : 0;
        { int rc2 = ensure(c, &c->d_hist, &c->hist_cap, (size_t)N); if (rc2) return rc2; }
        P.hist = c->d_hist;
        P.stats = c->d_stats;
        P.dbg = nullptr;
        if (const char *e = getenv("ISDF_DEBUG_TIMING")) if (e[0] == '1') {
            const size_t need = (size_t)c->M * 7 + (size_t)(N + 1) * 8;
            if (c->dbg_cap < need) {
                if (c->d_dbg) (void)hipFree(c->d_dbg);
                c->d_dbg = nullptr; c->dbg_cap = 0;
                HIPCHK(c, hipMalloc(&c->d_dbg, need * sizeof(unsigned long long)));
                c->dbg_cap = need;
            }
            HIPCHK(c, hipMemsetAsync(c->d_dbg, 0, need * sizeof(unsigned long long), st));
            c->dbg_used = need;
            P.dbg = c->d_dbg;
        }
        if (fixed_tstar) {               // the minimisers are given (isdf_eval_swept_at_tstar): no search
            launch_swept_fixed(P, d_tstar, st);
            launch_swept_reduce(P, d_out, st);
            HIPCHK(c, hipGetLastError());
            return ISDF_OK;
        }
        launch_swept_prepare(P, st);
        ProfEvent *ev;
        int rc = prof_begin(c, st, &ev);
        if (rc) return rc;
        launch_swept_sweep(P, st, ev ? ev->a : nullptr, ev ? ev->b : nullptr);
        const bool ev2 = ev && c->prof_secondary;
        if (ev2) { HIPCHK(c, hipEventRecord(ev->c, st)); }
        launch_swept_reduce(P, d_out, st);
        if (ev2) { HIPCHK(c, hipEventRecord(ev->d, st)); }
        HIPCHK(c, hipGetLastError());
        if (scan_lpt) { c->scan_order_b = order_b; c->scan_order_e = order_e; c->scan_order_epoch = c->points_epoch; }
        return ISDF_OK;
    }

    // ---- V2 / V3 integral sweep
    // the tile sweep's mesh walks keep MESH_Q_LEVELS levels of frames in LDS: a deeper hierarchy (a strongly unbalanced mesh)
    // would index past them - refused here, not corrupted there (the swept-volume sweep takes such meshes: wave-cooperative walks)
    if (cfg.enable_pos && c->shape.kind == ISDF_SHAPE_MESH && c->mesh_depth > isdf::MESH_Q_LEVELS)
        return fail(c, ISDF_ERR_UNSUPPORTED, "mesh hierarchy deeper than 12 levels: the tile sweep (V2 / V3) does not take it (the swept-volume sweep does)");
    if (cfg.enable_pos) {
        if (!c->have_geom) return fail(c, ISDF_ERR_STATE, "grid not set");
        if (cfg.variant == ISDF_V3_ESDF_TILE && !c->d_esdf) return fail(c, ISDF_ERR_STATE, "V3 needs an ESDF grid");
        if (cfg.variant == ISDF_V2_OCC_TILE && !c->d_occ) return fail(c, ISDF_ERR_STATE, "V2 needs an occupancy grid");
    }
    const long long total_pieces = (long long)n_traj * N;
    long long pb, pe;
    shard_range(total_pieces, c->rank, c->world, pb, pe);
    const size_t n_samples = (size_t)total_pieces * (cfg.integral_intervs + 1);
    { int rc0 = isdf_reserve_sweep_buffers(c, total_pieces); if (rc0) return rc0; }
    int rc = ISDF_OK;
    SweepParams P{};
    P.grid = c->grid;
    if (!c->have_geom) { P.grid.X = P.grid.Y = P.grid.Z = 1; P.grid.res = 1.0; }
    P.shape = c->shape;
    fill_flat(cfg, P.flat);
    P.variant = cfg.variant; P.K = cfg.integral_intervs; P.enable_dyn = cfg.enable_dyn; P.enable_pos = cfg.enable_pos;
    P.enable_cull = cfg.enable_cull;
    P.n_traj = n_traj; P.N = N; P.piece_begin = (int)pb; P.piece_end = (int)pe;
    P.bd_half = cfg.kernel_size * P.grid.res / 2;
    P.safety_hor = cfg.safety_hor; P.weight_p = cfg.weight_p; P.weight_v = cfg.weight_v; P.weight_omg = cfg.weight_omg;
    P.weight_theta = cfg.weight_theta;
    P.vel_sqr_max = cfg.vmax * cfg.vmax; P.omg_sqr_max = cfg.omgmax * cfg.omgmax; P.theta_max = cfg.thetamax;
    P.mu = cfg.smoothing_eps; P.inv_mu = 1.0 / cfg.smoothing_eps; P.inv_K = 1.0 / cfg.integral_intervs;
    P.occ_thresh = (float)cfg.occ_thresh;
    P.cull_threshold = 0.0;
    if (cfg.variant == ISDF_V3_ESDF_TILE && cfg.enable_cull && c->shape.bound_radius > 0)
        P.cull_threshold = c->shape.bound_radius + cfg.safety_hor + std::sqrt(3.0) * P.grid.res + (cfg.occ_thresh > 0 ? cfg.occ_thresh : 0.0);   // a qualifying voxel lies within occ_thresh of an occupied one
    P.T = d_T; P.coeffs = d_coeffs; P.acc = c->d_acc; P.sample_info = c->d_sample_info; P.piece_cost = c->d_piece_cost;
    P.out = d_out; P.stats = c->d_stats;
    P.dbg = nullptr;
    P.sample_map = (c->d_sample_map && c->sample_map_n == 4 * (((size_t)(pe - pb) * (cfg.integral_intervs + 1) + 3) / 4)) ? c->d_sample_map : nullptr;
    P.dbg_flags = 0;
    if (const char *e = getenv("ISDF_DEBUG_FLAGS")) P.dbg_flags = atoi(e);
    if (const char *e = getenv("ISDF_DEBUG_TIMING")) if (e[0] == '1') {
        const size_t need = n_samples * 8 + (size_t)total_pieces * 4 + 4;
        if (c->dbg_cap < need) {
            if (c->d_dbg) (void)hipFree(c->d_dbg);
            c->d_dbg = nullptr; c->dbg_cap = 0;
            HIPCHK(c, hipMalloc(&c->d_dbg, need * sizeof(unsigned long long)));
            c->dbg_cap = need;
        }
        HIPCHK(c, hipMemsetAsync(c->d_dbg, 0, need * sizeof(unsigned long long), st));
        c->dbg_used = need;
        P.dbg = c->d_dbg;
    }
    if (cfg.enable_pos && c->bits_dirty) {
        const int zw = (c->grid.Z + 31) / 32, xw = (c->grid.X + 31) / 32, yw = (c->grid.Y + 31) / 32;
        const size_t nwz = (size_t)c->grid.X * c->grid.Y * zw, nwx = (size_t)c->grid.Y * c->grid.Z * xw, nwy = (size_t)c->grid.X * c->grid.Z * yw;
        const size_t nw = nwz + nwx + nwy;
        if (c->bits_cap < nw) {
            if (c->d_bits) (void)hipFree(c->d_bits);
            c->d_bits = nullptr; c->bits_cap = 0;
            HIPCHK(c, hipMalloc(&c->d_bits, nw * sizeof(unsigned)));
            c->bits_cap = nw;
        }
        c->grid.ZW = zw; c->grid.XW = xw; c->grid.YW = yw;
        c->grid.bits = c->d_bits; c->grid.bits_x = c->d_bits + nwz; c->grid.bits_y = c->d_bits + nwz + nwx;
        launch_build_bits(c->grid, cfg.variant == ISDF_V3_ESDF_TILE ? 1 : 0, cfg.occ_thresh, c->d_bits, st);
        launch_build_bits_xy(c->grid, cfg.variant == ISDF_V3_ESDF_TILE ? 1 : 0, cfg.occ_thresh, c->d_bits + nwz, c->d_bits + nwz + nwx, st);
        c->bits_dirty = false;
        P.grid = c->grid;
    }
    c->last_P = P; c->have_last_P = true;
    ProfEvent *ev;
    rc = prof_begin(c, st, &ev);
    if (rc) return rc;
    if (ev && !cfg.enable_pos) { ev = nullptr; c->prof_used--; }      // no dominant kernel in a dynamics-only step
    const bool ev2 = ev && c->prof_secondary;
    // a small step (one trajectory) is ONE launch: the tail's workgroups ride behind the sweep's (tile_sweep.hip, FUSED)
    const bool fused = c->fuse_small && !ev2 && sweep_can_fuse(P);
    {   // multi-GPU step with the exchange inside the step's launches (isdf_xchg_fuse)
        int xerr = ISDF_OK;
        if (isdf_xchg_fill(c, &P.xf, (size_t)total_pieces, &xerr)) {
            if (xerr != ISDF_OK) return xerr;
            // a step that is not one fused launch carries the exchange in its tail launch (tail_kernel_xf); a step without a
            // collision term has no sweep launch either way
        }
    }
    if (hd) {
        if (!fused || P.xf.world > 1 || P.dbg) return ISDF_DIRECT_NA;
        { int rc1 = ensure_stage(c, (size_t)total_pieces); if (rc1) return rc1; }
        P.out = hd->out; P.host_flag = hd->flags; P.seq = hd->seq;
        if (!hd->via_bar) { P.host_T = hd->T; P.host_coeffs = hd->coeffs; P.stage = c->d_stage; P.stage_flags = c->d_stage_flags; }
        c->last_P = P;
    }
    // dispatch order built on the device from the work of earlier steps (tile_sweep.hip, plan_wave): a resident (fused) launch
    // gives the heaviest samples to the workgroups that are first on their CU; every other launch groups samples of like weight
    // into workgroups and dispatches the heaviest first
    P.plan_cls_in = nullptr; P.plan_cls_out = nullptr; P.plan_map_out = nullptr; P.plan_zone = 0; P.plan_group = 1;
    P.plan_lr_in = P.plan_hist_in = nullptr; P.plan_lr_out = P.plan_hist_out = nullptr;
    {
        const int K1 = cfg.integral_intervs + 1;
        const long long n_loc = pe - pb, ns_local = n_loc * K1;
        const int nb = (int)((ns_local + 3) / 4);
        P.n_sweep_blocks = nb;                  // (launch_sweep sets it again for its own copy; the tail launch needs it for the order it writes)
        if (c->n_cus == 0) { hipDeviceProp_t pr; c->n_cus = (hipGetDeviceProperties(&pr, c->device) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256; }
        const bool no_plan = c->env_no_lpt;
        // (not for the mesh kind: the cost of a mesh sample is its hierarchy walks, which the pair count does not predict -
        // measured 230 -> 256 us per step with the order on, 20-face mesh)
        bool plan = cfg.enable_pos && !no_plan && !P.sample_map && K1 <= 128 && nb > c->n_cus && ns_local < (1LL << 28) && c->shape.kind != ISDF_SHAPE_MESH;
        long long group = n_loc;
        int zone = 0;
        if (plan) {
            if (fused) {
                // resident launch: zones; the exchange inside a fused multi-GPU launch keeps the plain order
                plan = P.xf.world <= 1 && pb == 0 && pe == total_pieces && ns_local <= PLAN_GROUP_MAX_SAMPLES;
                zone = c->n_cus;
            } else if (nb <= sweep_resident_blocks(P, c->n_cus) && ns_local <= PLAN_GROUP_MAX_SAMPLES) {
                zone = c->n_cus;          // a two-launch step whose sweep is resident anyway (a 50-piece shard): zones as well
            } else if (ns_local > PLAN_GROUP_MAX_SAMPLES) {
                // a batch: every trajectory is sorted by itself (the shard must hold whole trajectories)
                group = N;
                plan = (long long)N * K1 <= PLAN_GROUP_MAX_SAMPLES && pb % N == 0 && pe % N == 0;
            }
        }
        if (!plan) c->plan_k = 0;
        else {
            const size_t ns_cap = (size_t)4 * nb, np_cap = (size_t)n_loc;
            if (c->plan_ns_cap < ns_cap || c->plan_np_cap < np_cap) {
                HIPCHK(c, hipStreamSynchronize(st));
                for (void *q : {(void *)c->d_plan_cls, (void *)c->d_plan_map, (void *)c->d_plan_lr, (void *)c->d_plan_hist}) if (q) (void)hipFree(q);
                c->d_plan_cls = nullptr; c->d_plan_map = nullptr; c->d_plan_lr = nullptr; c->d_plan_hist = nullptr;
                c->plan_ns_cap = c->plan_np_cap = 0; c->plan_k = 0; c->plan_cur = 0;
                const size_t nsc = std::max(ns_cap, (size_t)4096), npc = std::max(np_cap, (size_t)256);
                HIPCHK(c, hipMalloc(&c->d_plan_cls, 2 * nsc));
                HIPCHK(c, hipMalloc(&c->d_plan_map, 2 * nsc * sizeof(int)));
                HIPCHK(c, hipMalloc(&c->d_plan_lr, 2 * nsc * sizeof(unsigned short)));
                HIPCHK(c, hipMalloc(&c->d_plan_hist, 2 * npc * PLAN_CLASSES * sizeof(unsigned short)));
                c->plan_ns_cap = nsc; c->plan_np_cap = npc;
            }
            // the records of earlier steps are laid out by piece: they only carry over to a launch of the same geometry
            const long long geo = ((long long)n_loc << 32) | ((long long)K1 << 20) | ((long long)(fused ? 1 : 0) << 19) | (long long)(group & 0x7FFFF);
            if (c->plan_ns != ns_local || c->plan_nb != nb || c->plan_geo != geo) { c->plan_k = 0; c->plan_ns = ns_local; c->plan_nb = nb; c->plan_geo = geo; }
            // The order is rebuilt once per cycle of PLAN_CYCLE steps (the trajectory moves little between optimizer steps, and
            // building costs the step 0.4 us): step 0 of a cycle - the sweep leaves the classes; step 1 - the tail workgroups
            // turn them into records; step 2 - into the order, written into the buffer NOT in use; from step 3 on the launches
            // run in it.  Single copies of classes / records suffice: each is written in one step and read in the next.
            constexpr int PLAN_CYCLE = 8;
            const int k = c->plan_k, ph = k % PLAN_CYCLE;
            const size_t nsc = c->plan_ns_cap;
            P.plan_zone = zone;
            P.plan_group = (int)group;
            if (ph == 0) P.plan_cls_out = c->d_plan_cls;
            if (ph == 1) { P.plan_cls_in = c->d_plan_cls; P.plan_lr_out = c->d_plan_lr; P.plan_hist_out = c->d_plan_hist; }
            if (ph == 2) { P.plan_lr_in = c->d_plan_lr; P.plan_hist_in = c->d_plan_hist; P.plan_map_out = c->d_plan_map + (size_t)(1 - c->plan_cur) * nsc; }
            if (ph == 3) c->plan_cur = 1 - c->plan_cur;          // the order written by the previous step is complete
            if (k >= 3) P.sample_map = c->d_plan_map + (size_t)c->plan_cur * nsc;
            if (c->plan_k < (1 << 30)) c->plan_k++;
        }
    }
    if (!fused && cfg.enable_pos) {                 // the poses of a non-fused launch come from pose_kernel (tile_sweep.hip)
        const size_t need = (size_t)(pe - pb) * (cfg.integral_intervs + 1) * sweep_pose_bytes();
        if (c->pose_cap < need) {
            HIPCHK(c, hipStreamSynchronize(st));
            if (c->d_pose) (void)hipFree(c->d_pose);
            c->d_pose = nullptr; c->pose_cap = 0;
            HIPCHK(c, hipMalloc(&c->d_pose, need));
            c->pose_cap = need;
        }
        P.poses = c->d_pose;
    }
    // ---- mesh robots: the exact pass runs as its own launch over a queue of 64-voxel blocks (every block an independent work item:
    // the launch is balanced over the whole device instead of ending on its heaviest workgroup).  Sized for the worst case the
    // geometry allows - every voxel of the robot's inflated bounding box (or of the tile, if smaller) occupied; beyond the budget
    // below the exact pass stays inside the sweep kernel (ISDF_MESH_QUEUE=0 forces that).
    P.mq_items = nullptr;
    if (c->shape.kind == ISDF_SHAPE_MESH && cfg.enable_pos && !fused && !(getenv("ISDF_MESH_QUEUE") && getenv("ISDF_MESH_QUEUE")[0] == '0')) {
        const size_t ns_loc = (size_t)(pe - pb) * (cfg.integral_intervs + 1);
        const double k3 = (double)cfg.kernel_size * cfg.kernel_size * cfg.kernel_size;
        double vox = k3;
        if (c->shape.prune_rows) {
            double v = 1.0;
            for (int a = 0; a < 3; a++) v *= ((double)c->shape.bbox_hi[a] - (double)c->shape.bbox_lo[a] + 2.2 * cfg.safety_hor) / P.grid.res + 3.0;
            vox = std::min(k3, v * 1.8);                  // (rows are pruned by the box's extent on the WORLD axes: up to sqrt(3) per axis for a rotated box)
        }
        const int kmax = (int)(k3 / 16.0) + 2;            // items are 16 voxels (tile_sweep.hip MQ_BLOCK)
        const size_t per_sample = (size_t)(vox / 16.0) + 2;
        const size_t cap = ns_loc * per_sample;
        const size_t bytes = cap * (16 * 4 + 8 + 80) + ns_loc * ((size_t)kmax * 4 + 4);
        // budget: 1 GiB per ctx (ISDF_MESH_QUEUE_MAX_MB overrides).  A launch whose worst case needs more keeps the exact pass inside
        // the sweep kernel (whose 64-voxel blocks partition a sample's sums differently: equal to rounding, not bitwise)
        static const size_t mq_budget = [] { const char *e = getenv("ISDF_MESH_QUEUE_MAX_MB"); const long v = e ? atol(e) : 0; return (size_t)(v > 0 ? v : 1024) << 20; }();
        if (ns_loc > 0 && bytes <= mq_budget && cap < 0x7fffffffull) {
            if (c->mq_cap < cap || c->mq_samples_cap < ns_loc || c->mq_kmax != kmax) {
                HIPCHK(c, hipDeviceSynchronize());               // (an earlier step may still be reading the old queue)
                for (void *q : {(void *)c->d_mq_entries, c->d_mq_items, (void *)c->d_mq_res, (void *)c->d_mq_sample_items, (void *)c->d_mq_sample_n}) if (q) (void)hipFree(q);
                c->d_mq_entries = nullptr; c->d_mq_items = nullptr; c->d_mq_res = nullptr; c->d_mq_sample_items = nullptr; c->d_mq_sample_n = nullptr;
                c->mq_cap = 0; c->mq_samples_cap = 0;
                HIPCHK(c, hipMalloc(&c->d_mq_entries, cap * 16 * sizeof(unsigned)));
                HIPCHK(c, hipMalloc(&c->d_mq_items, cap * 8));
                HIPCHK(c, hipMalloc(&c->d_mq_res, cap * 10 * sizeof(double)));
                HIPCHK(c, hipMalloc(&c->d_mq_sample_items, ns_loc * (size_t)kmax * sizeof(int)));
                HIPCHK(c, hipMalloc(&c->d_mq_sample_n, ns_loc * sizeof(int)));
                if (!c->d_mq_count) HIPCHK(c, hipMalloc(&c->d_mq_count, 4 * sizeof(unsigned)));
                c->mq_cap = cap; c->mq_samples_cap = ns_loc; c->mq_kmax = kmax;
            }
            P.mq_entries = c->d_mq_entries; P.mq_items = (int2 *)c->d_mq_items; P.mq_res = c->d_mq_res;
            P.mq_sample_items = c->d_mq_sample_items; P.mq_sample_n = c->d_mq_sample_n; P.mq_count = c->d_mq_count;
            P.mq_cap = (unsigned)c->mq_cap; P.mq_kmax = kmax;
        }
    }
    launch_sweep(P, st, ev ? ev->a : nullptr, ev ? ev->b : nullptr, fused);
    if (!fused) launch_tail(P, st, ev2 ? ev->c : nullptr, ev2 ? ev->d : nullptr);
    HIPCHK(c, hipGetLastError());
    return ISDF_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// ONE host process, SEVERAL devices (isdf_create_multi; SURVEY 8(b) "Threading": launch -> all-reduce -> D2H from the calling
// thread, no extra host threads).  The lead ctx is shard 0 on devices[0]; every further device has a plain ctx of its own
// (shard r of n) that the lead owns.  A step: the inputs go to every device (peer copies ordered behind the caller's stream by
// an event), every shard is queued on its own device's stream FROM THE CALLING THREAD, every shard leaves [packed outputs | its 8
// statistics words as doubles] in its part buffer, and the parts are summed on the lead in rank order -
//   ISDF_MULTI_PEER_SUM  one kernel on the lead that reads the peers' parts straight over xGMI (peer access),
//   ISDF_MULTI_STAGED    peer copies into a staging buffer on the lead + the same kernel locally (no peer access needed),
//   ISDF_MULTI_RCCL      ncclAllReduce(sum, ncclDouble) over the part buffers in one group call (librccl.so by dlopen, so the
//                        library neither links nor needs RCCL unless asked: ISDF_MULTI_COLLECTIVE=rccl),
// after which the caller's stream holds the full [cost | gradT | gradC] - exactly what the single-device step leaves.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int MULTI_TAIL = 8;
struct MultiParts { const double *p[XCHG_MAX_WORLD]; const unsigned long long *st[XCHG_MAX_WORLD]; int n; };   // st[0] != null: the shards' statistics words are read in place
// the 8 statistics words of a shard behind its packed outputs, as doubles: they ride through whichever sum is in force
__global__ void multi_tail_kernel(const unsigned long long *stats, double *tail) {
    if (threadIdx.x < MULTI_TAIL) tail[threadIdx.x] = (double)stats[threadIdx.x];
}
// out[i] = part_0[i] + part_1[i] + ... (rank order: bitwise reproducible); the summed tail back into the lead's statistics words
// (`all_stats`: V1 counts straight into the statistics words, so their device sums are the step's statistics; V2 / V3 words are
// filled on demand by isdf_get_stats and only the overflow word [4] travels)
// host_words != null: `out` is host-mapped; the step's 8 statistics words follow it into host_words[0..7] and, once every block of
// this launch has written, host_words[8] = seq tells the spinning host that the step is complete
__global__ __launch_bounds__(256) void multi_sum_kernel(double *out, MultiParts parts, size_t count, unsigned long long *stats, int all_stats,
                                                        unsigned long long *host_words, unsigned *blocks_done, unsigned long long seq) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count + MULTI_TAIL) {
        double s;
        if (i >= count && parts.st[0]) {                                    // (peer access: no tail launches, the words are read where they are)
            s = (double)parts.st[0][i - count];
            for (int r = 1; r < parts.n; r++) s += (double)parts.st[r][i - count];
        } else {
            s = parts.p[0][i];
            for (int r = 1; r < parts.n; r++) s += parts.p[r][i];
        }
        if (i < count) out[i] = s;
        else {
            const int k = (int)(i - count);
            if (k == 4) { if (s != 0.0) stats[4] = 1ull; }                 // overflow: sticky until read
            else if (all_stats) stats[k] = (unsigned long long)s;
            if (host_words) host_words[k] = k == 4 ? ((s != 0.0 || stats[4] != 0ull) ? 1ull : 0ull) : (unsigned long long)s;
        }
    }
    if (host_words) {
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0 && atomicAdd(blocks_done, 1u) == gridDim.x - 1u) {
            *blocks_done = 0u;
            __threadfence_system();
            *(volatile unsigned long long *)(host_words + 8) = seq;
        }
    }
}

namespace {
struct RcclApi {
    void *lib = nullptr;
    int (*CommInitAll)(void **, int, const int *) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
RcclApi g_rccl;
bool rccl_load() {
    if (g_rccl.lib) return g_rccl.AllReduce != nullptr;
    for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) { g_rccl.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (g_rccl.lib) break; }
    if (!g_rccl.lib) return false;
    g_rccl.CommInitAll = (int (*)(void **, int, const int *))dlsym(g_rccl.lib, "ncclCommInitAll");
    g_rccl.CommDestroy = (int (*)(void *))dlsym(g_rccl.lib, "ncclCommDestroy");
    g_rccl.AllReduce = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))dlsym(g_rccl.lib, "ncclAllReduce");
    g_rccl.GroupStart = (int (*)())dlsym(g_rccl.lib, "ncclGroupStart");
    g_rccl.GroupEnd = (int (*)())dlsym(g_rccl.lib, "ncclGroupEnd");
    g_rccl.GetErrorString = (const char *(*)(int))dlsym(g_rccl.lib, "ncclGetErrorString");
    if (!g_rccl.CommInitAll || !g_rccl.CommDestroy || !g_rccl.AllReduce || !g_rccl.GroupStart || !g_rccl.GroupEnd) { g_rccl.AllReduce = nullptr; return false; }
    return true;
}
}  // namespace

static void multi_release(isdf_ctx *c) {
    if (c->rccl_comm && g_rccl.CommDestroy) { (void)g_rccl.CommDestroy(c->rccl_comm); c->rccl_comm = nullptr; }
    if (c->mev_in) { (void)hipEventDestroy(c->mev_in); c->mev_in = nullptr; }
    if (c->mev_done) { (void)hipEventDestroy(c->mev_done); c->mev_done = nullptr; }
    if (c->d_mpart) { (void)hipFree(c->d_mpart); c->d_mpart = nullptr; }
    if (c->d_mstage) { (void)hipFree(c->d_mstage); c->d_mstage = nullptr; }
}

extern "C" int isdf_create_multi(isdf_ctx **out, const isdf_config *cfg, const int *devices, int n_devices) {
    if (!out || !cfg || !devices) return fail(nullptr, ISDF_ERR_INVALID_ARG, "null argument");
    *out = nullptr;
    if (n_devices < 1 || n_devices > XCHG_MAX_WORLD) return fail(nullptr, ISDF_ERR_INVALID_ARG, "1 .. 16 devices");
    std::vector<isdf_ctx *> all;
    auto bail = [&](int code, const std::string &msg) { for (isdf_ctx *q : all) { q->is_peer = false; q->peers.clear(); (void)isdf_destroy(q); } return fail(nullptr, code, msg.c_str()); };
    for (int r = 0; r < n_devices; r++) {
        isdf_config cr = *cfg;
        cr.device = devices[r];
        isdf_ctx *q = nullptr;
        const int rc = isdf_create(&q, &cr);
        if (rc != ISDF_OK) return bail(rc, std::string("device ") + std::to_string(devices[r]) + ": " + isdf_last_error(nullptr));
        all.push_back(q);
        q->rank = r; q->world = n_devices;
        if (hipSetDevice(q->device) != hipSuccess || hipEventCreateWithFlags(&q->mev_done, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&q->mev_in, hipEventDisableTiming) != hipSuccess)
            return bail(ISDF_ERR_HIP, "event creation failed");
    }
    isdf_ctx *lead = all[0];
    // how the parts are summed: the lead reads the peers' buffers directly when every peer is reachable
    int mode = ISDF_MULTI_PEER_SUM;
    (void)hipSetDevice(lead->device);
    for (int r = 1; r < n_devices && mode == ISDF_MULTI_PEER_SUM; r++) {
        if (devices[r] == lead->device) continue;                         // the same device listed again (tests): plain pointers
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, lead->device, devices[r]) != hipSuccess || !can) { mode = ISDF_MULTI_STAGED; break; }
        const hipError_t e = hipDeviceEnablePeerAccess(devices[r], 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) mode = ISDF_MULTI_STAGED;
        (void)hipGetLastError();
    }
    if (const char *e = getenv("ISDF_MULTI_COLLECTIVE")) {
        if (!std::strcmp(e, "staged")) mode = ISDF_MULTI_STAGED;
        else if (!std::strcmp(e, "peer")) { if (mode != ISDF_MULTI_PEER_SUM) return bail(ISDF_ERR_UNSUPPORTED, "ISDF_MULTI_COLLECTIVE=peer: no peer access between the listed devices"); }
        else if (!std::strcmp(e, "rccl")) {
            for (int a = 0; a < n_devices; a++) for (int b = a + 1; b < n_devices; b++)
                if (devices[a] == devices[b]) return bail(ISDF_ERR_UNSUPPORTED, "ISDF_MULTI_COLLECTIVE=rccl needs DISTINCT devices (one communicator rank per GPU)");
            if (!rccl_load()) return bail(ISDF_ERR_UNSUPPORTED, "ISDF_MULTI_COLLECTIVE=rccl: librccl.so could not be loaded");
            std::vector<void *> comms(n_devices, nullptr);
            const int rr = g_rccl.CommInitAll(comms.data(), n_devices, devices);
            if (rr != 0) return bail(ISDF_ERR_UNSUPPORTED, std::string("ncclCommInitAll: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rr) : "failed") + " (one communicator per DISTINCT device)");
            for (int r = 0; r < n_devices; r++) all[r]->rccl_comm = comms[r];
            mode = ISDF_MULTI_RCCL;
        } else return bail(ISDF_ERR_INVALID_ARG, "ISDF_MULTI_COLLECTIVE must be peer, staged or rccl");
    }
    // PULL: with peer access in BOTH directions the peers' kernels read T | coeffs (and, for the swept-volume sweep, read and write
    // their points' lastTstar) straight in the lead's memory over xGMI - 152 N bytes per step - instead of two to four peer copies
    // per device and step queued from the calling thread; the lead's sum reads the peers' statistics words in place
    bool pull = mode == ISDF_MULTI_PEER_SUM && !(getenv("ISDF_MULTI_NO_PULL") && getenv("ISDF_MULTI_NO_PULL")[0] == '1');
    for (int r = 1; r < n_devices && pull; r++) {
        if (devices[r] == lead->device) continue;
        int can = 0;
        if (hipSetDevice(devices[r]) != hipSuccess || hipDeviceCanAccessPeer(&can, devices[r], lead->device) != hipSuccess || !can) { pull = false; break; }
        const hipError_t e = hipDeviceEnablePeerAccess(lead->device, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) pull = false;
        (void)hipGetLastError();
    }
    lead->multi_pull = pull;
    for (int r = 1; r < n_devices; r++) { all[r]->is_peer = true; lead->peers.push_back(all[r]); }
    lead->multi_collective = mode;
    (void)hipSetDevice(lead->device);
    *out = lead;
    return ISDF_OK;
}
extern "C" int isdf_multi_info(const isdf_ctx *c, int *n_devices_out, int *collective_out) {
    if (!c) return ISDF_ERR_INVALID_ARG;
    if (n_devices_out) *n_devices_out = 1 + (int)c->peers.size();
    if (collective_out) *collective_out = c->peers.empty() ? ISDF_MULTI_NONE : c->multi_collective;
    return ISDF_OK;
}

// One step on every device.  d_T / d_coeffs / d_out / d_tstar live on the LEAD's device and are ordered on `st` (a stream of the
// lead's device) like in the single-device call.
static int multi_eval_device(isdf_ctx *c, int n_traj, int N, const double *d_T, const double *d_coeffs, double *d_out,
                             double *d_tstar, hipStream_t st, int mode, bool fixed_tstar) {
    const int n = 1 + (int)c->peers.size();
    const size_t count = (size_t)n_traj * isdf_out_stride(N), in_T = (size_t)n_traj * N, in_C = (size_t)n_traj * 18 * N;
    const bool swept = c->cfg.variant == ISDF_V1_SWEPT && mode != 2;
    if (fixed_tstar) return fail(c, ISDF_ERR_UNSUPPORTED, "isdf_eval_swept_at_tstar on a multi-device ctx");
    HIPCHK(c, hipSetDevice(c->device));
    // the previous step's sum read the peers' part buffers and this ctx's own: a step issued on ANOTHER caller stream must not
    // overwrite them before that sum has run (the peers' streams follow through mev_in below)
    if (c->msum_recorded) HIPCHK(c, hipStreamWaitEvent(st, c->mev_done, 0));
    HIPCHK(c, hipEventRecord(c->mev_in, st));                              // the caller's inputs are ready from here on
    double *lead_ts = swept ? (d_tstar ? d_tstar : c->d_tstar) : nullptr;
    const bool pull = c->multi_pull && c->multi_collective == ISDF_MULTI_PEER_SUM;
    MultiParts parts{};
    parts.n = n;
    unsigned long long *const mh_words = c->mh_words;          // (isdf_eval arms this per step)
    c->mh_words = nullptr;
    for (int r = 1; r < n; r++) {
        isdf_ctx *p = c->peers[r - 1];
        HIPCHK(c, hipSetDevice(p->device));
        int rc = pull ? ISDF_OK : ensure(p, &p->d_in, &p->in_cap, in_T + in_C);
        if (rc == ISDF_OK) rc = ensure(p, &p->d_mpart, &p->mpart_cap, count + MULTI_TAIL);
        if (rc) { c->err = p->err; return rc; }
        HIPCHK(c, hipStreamWaitEvent(p->stream, c->mev_in, 0));
        long long pb = 0, pe = 0;
        if (pull) {
            // the shard reads the lead's inputs in place (peer access); lastTstar likewise - every point belongs to ONE shard
            rc = eval_device_impl(p, n_traj, N, d_T, d_coeffs, p->d_mpart, lead_ts, p->stream, mode, false);
            if (rc) { c->err = "device " + std::to_string(p->device) + ": " + p->err; return rc; }
            parts.st[r] = p->d_stats;
        } else {
            HIPCHK(c, hipMemcpyPeerAsync(p->d_in, p->device, d_T, c->device, in_T * sizeof(double), p->stream));
            HIPCHK(c, hipMemcpyPeerAsync(p->d_in + in_T, p->device, d_coeffs, c->device, in_C * sizeof(double), p->stream));
            if (swept && p->M > 0) {                                           // lastTstar of this shard's points travels with it
                shard_range(p->M, p->rank, p->world, pb, pe);
                if (pe > pb && lead_ts) HIPCHK(c, hipMemcpyPeerAsync(p->d_tstar + pb, p->device, lead_ts + pb, c->device, (size_t)(pe - pb) * sizeof(double), p->stream));
            }
            rc = eval_device_impl(p, n_traj, N, p->d_in, p->d_in + in_T, p->d_mpart, nullptr, p->stream, mode, false);
            if (rc) { c->err = "device " + std::to_string(p->device) + ": " + p->err; return rc; }
            hipLaunchKernelGGL(multi_tail_kernel, dim3(1), dim3(64), 0, p->stream, p->d_stats, p->d_mpart + count);
            if (swept && pe > pb && lead_ts) HIPCHK(c, hipMemcpyPeerAsync(lead_ts + pb, c->device, p->d_tstar + pb, p->device, (size_t)(pe - pb) * sizeof(double), p->stream));
        }
        HIPCHK(c, hipEventRecord(p->mev_done, p->stream));
        parts.p[r] = p->d_mpart;
    }
    HIPCHK(c, hipSetDevice(c->device));
    int rc = ensure(c, &c->d_mpart, &c->mpart_cap, count + MULTI_TAIL);
    if (rc) return rc;
    rc = eval_device_impl(c, n_traj, N, d_T, d_coeffs, c->d_mpart, d_tstar, st, mode, false);
    if (rc) return rc;
    if (pull) parts.st[0] = c->d_stats;
    else hipLaunchKernelGGL(multi_tail_kernel, dim3(1), dim3(64), 0, st, c->d_stats, c->d_mpart + count);
    parts.p[0] = c->d_mpart;
    const dim3 grid((unsigned)((count + MULTI_TAIL + 255) / 256)), block(256);
    if (c->multi_collective == ISDF_MULTI_RCCL) {
        // ONE all-reduce of the packed vector per step: every device's part in place, issued from this thread as one group
        for (int r = 1; r < n; r++) HIPCHK(c, hipStreamWaitEvent(st, c->peers[r - 1]->mev_done, 0));      // (the merged lastTstar)
        if (g_rccl.GroupStart() != 0) return fail(c, ISDF_ERR_HIP, "ncclGroupStart failed");
        for (int r = 0; r < n; r++) {
            isdf_ctx *q = r == 0 ? c : c->peers[r - 1];
            hipStream_t qs = r == 0 ? st : q->stream;
            if (r > 0) HIPCHK(c, hipSetDevice(q->device));
            const int rr = g_rccl.AllReduce(q->d_mpart, q->d_mpart, count + MULTI_TAIL, 8 /* ncclDouble */, 0 /* ncclSum */, q->rccl_comm, qs);
            if (rr != 0) { (void)g_rccl.GroupEnd(); return fail(c, ISDF_ERR_HIP, "ncclAllReduce failed"); }
        }
        if (g_rccl.GroupEnd() != 0) return fail(c, ISDF_ERR_HIP, "ncclGroupEnd failed");
        HIPCHK(c, hipSetDevice(c->device));
        MultiParts one{}; one.n = 1; one.p[0] = c->d_mpart;
        hipLaunchKernelGGL(multi_sum_kernel, grid, block, 0, st, d_out, one, count, c->d_stats, swept ? 1 : 0, mh_words, c->d_msum_blocks, c->mh_seq);
    } else {
        for (int r = 1; r < n; r++) HIPCHK(c, hipStreamWaitEvent(st, c->peers[r - 1]->mev_done, 0));
        if (c->multi_collective == ISDF_MULTI_STAGED) {
            rc = ensure(c, &c->d_mstage, &c->mstage_cap, (size_t)(n - 1) * (count + MULTI_TAIL));
            if (rc) return rc;
            for (int r = 1; r < n; r++) {
                double *dst = c->d_mstage + (size_t)(r - 1) * (count + MULTI_TAIL);
                HIPCHK(c, hipMemcpyPeerAsync(dst, c->device, c->peers[r - 1]->d_mpart, c->peers[r - 1]->device, (count + MULTI_TAIL) * sizeof(double), st));
                parts.p[r] = dst;
            }
        }
        hipLaunchKernelGGL(multi_sum_kernel, grid, block, 0, st, d_out, parts, count, c->d_stats, swept ? 1 : 0, mh_words, c->d_msum_blocks, c->mh_seq);
    }
    HIPCHK(c, hipGetLastError());
    // the next step's peer copies overwrite the peers' inputs: they are ordered behind THIS step's kernels by the peers' own
    // streams; the part buffers behind the lead's "sum done" event (its own mev_done: the lead records no other use of it)
    HIPCHK(c, hipEventRecord(c->mev_done, st));
    c->msum_recorded = true;
    return ISDF_OK;
}

// every sweep of the host paths goes through here
static int sweep_dispatch(isdf_ctx *c, int n_traj, int N, const double *d_T, const double *d_coeffs, double *d_out, double *d_tstar,
                          hipStream_t st, int mode = 0, bool fixed_tstar = false) {
    // (a one-device ctx created with the RCCL collective takes the multi-device path too: the all-reduce of a world of one)
    if (!c->peers.empty() || c->rccl_comm) return multi_eval_device(c, n_traj, N, d_T, d_coeffs, d_out, d_tstar, st, mode, fixed_tstar);
    return eval_device_impl(c, n_traj, N, d_T, d_coeffs, d_out, d_tstar, st, mode, fixed_tstar);
}

extern "C" int isdf_eval_device(isdf_ctx *c, int n_traj, int N, const double *d_T, const double *d_coeffs, double *d_out,
                                double *d_tstar, void *stream) {
    if (!c) return ISDF_ERR_INVALID_ARG;
    if (c->is_peer) return fail(c, ISDF_ERR_STATE, "this ctx belongs to a multi-device ctx");
    return sweep_dispatch(c, n_traj, N, d_T, d_coeffs, d_out, d_tstar, (hipStream_t)stream);
}

// The swept-volume sweep's back-prop with the minimisers GIVEN: for every obstacle point the robot SDF and its body-frame
// gradient are evaluated at d_tstar[pt] instead of being searched for (a negative / NaN entry = "no interval qualified").
extern "C" int isdf_eval_swept_at_tstar(isdf_ctx *c, int N, const double *d_T, const double *d_coeffs, double *d_out,
                                        const double *d_tstar, void *stream) {
    if (!c) return ISDF_ERR_INVALID_ARG;
    if (!d_tstar) return fail(c, ISDF_ERR_INVALID_ARG, "null t* buffer");
    // (a multi-device ctx would return its lead's shard only: refused, as the header says)
    if (!c->peers.empty() || c->is_peer || c->rccl_comm) return fail(c, ISDF_ERR_UNSUPPORTED, "isdf_eval_swept_at_tstar on a multi-device ctx");
    return eval_device_impl(c, 1, N, d_T, d_coeffs, d_out, const_cast<double *>(d_tstar), (hipStream_t)stream, 1, true);
}

// Host-array form of the above (accumulate semantics like isdf_eval): tstar = M doubles.
extern "C" int isdf_eval_swept_at_tstar_host(isdf_ctx *c, int N, const double *T, const double *coeffs, const double *tstar,
                                             double *cost_inout, double *gradT_inout, double *gradC_inout) {
    if (!c) return ISDF_ERR_INVALID_ARG;
    if (N < 1 || !T || !coeffs || !tstar || !cost_inout || !gradT_inout || !gradC_inout) return fail(c, ISDF_ERR_INVALID_ARG, "null argument");
    if (!c->peers.empty() || c->is_peer || c->rccl_comm) return fail(c, ISDF_ERR_UNSUPPORTED, "isdf_eval_swept_at_tstar on a multi-device ctx");
    if (c->M <= 0) return ISDF_OK;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t in_all = (size_t)19 * N, ostride = isdf_out_stride(N);
    int rc = ensure(c, &c->d_in, &c->in_cap, in_all + (size_t)c->M);
    if (rc) return rc;
    rc = ensure(c, &c->d_out, &c->out_cap, ostride);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->d_in, T, (size_t)N * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_in + N, coeffs, (size_t)18 * N * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_in + in_all, tstar, (size_t)c->M * sizeof(double), hipMemcpyHostToDevice, c->stream));
    rc = eval_device_impl(c, 1, N, c->d_in, c->d_in + N, c->d_out, c->d_in + in_all, c->stream, 1, true);
    if (rc) return rc;
    std::vector<double> h(ostride);
    HIPCHK(c, hipMemcpyAsync(h.data(), c->d_out, ostride * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *cost_inout += h[0];
    for (int i = 0; i < N; i++) gradT_inout[i] += h[1 + i];
    for (int i = 0; i < 18 * N; i++) gradC_inout[i] += h[1 + N + i];
    return ISDF_OK;
}

static int fetch_stats(isdf_ctx *c);
// the peers' pair statistics of the last launch added to the lead's last_stats (V2 / V3; the V1 words of a multi-device step are
// already the devices' sums)
static int add_peer_stats(isdf_ctx *c) {
    if (c->cfg.variant == ISDF_V1_SWEPT || c->peers.empty()) return ISDF_OK;
    for (isdf_ctx *p : c->peers) {
        HIPCHK(c, hipSetDevice(p->device));
        HIPCHK(c, hipDeviceSynchronize());
        const int rc = fetch_stats(p);
        if (rc) { c->err = p->err; return rc; }
        c->last_stats.n_units += p->last_stats.n_units; c->last_stats.n_units_culled += p->last_stats.n_units_culled;
        c->last_stats.n_pairs += p->last_stats.n_pairs; c->last_stats.n_grad_pairs += p->last_stats.n_grad_pairs;
        c->last_stats.overflow |= p->last_stats.overflow;
    }
    HIPCHK(c, hipSetDevice(c->device));
    return ISDF_OK;
}
static int fetch_stats(isdf_ctx *c) {
    unsigned long long h[8];
    if (c->cfg.variant != ISDF_V1_SWEPT && c->have_last_P) {
        HIPCHK(c, hipDeviceSynchronize());
        HIPCHK(c, hipMemset(c->d_stats, 0, 4 * sizeof(unsigned long long)));   // keep [4] = overflow flag
        launch_stats(c->last_P, nullptr);
        HIPCHK(c, hipDeviceSynchronize());
    }
    HIPCHK(c, hipMemcpy(h, c->d_stats, sizeof(h), hipMemcpyDeviceToHost));
    c->last_stats.n_units = (int64_t)h[0]; c->last_stats.n_units_culled = (int64_t)h[1];
    c->last_stats.n_pairs = (int64_t)h[2]; c->last_stats.n_grad_pairs = (int64_t)h[3];
    c->last_stats.overflow = (int32_t)h[4];
    if (h[4]) {                                                                       // sticky until read
        HIPCHK(c, hipMemset(c->d_stats + 4, 0, sizeof(unsigned long long)));
        const int rr = isdf_reset_result_slots(c);                                      // a late producer must not feed the next step
        if (rr) return rr;
    }
    return ISDF_OK;
}

// pinned staging buffer of the host entry points (isdf_eval): [inputs | outputs | 8 statistics words]
static int ensure_eval_pin(isdf_ctx *c, size_t doubles) {
    if (c->eval_pin_cap >= doubles) return ISDF_OK;
    if (c->h_eval_pin) (void)hipHostFree(c->h_eval_pin);
    c->h_eval_pin = nullptr; c->eval_pin_cap = 0;
    HIPCHK(c, hipHostMalloc((void **)&c->h_eval_pin, doubles * sizeof(double), hipHostMallocDefault));
    HIPCHK(c, hipHostGetDevicePointer((void **)&c->h_eval_pin_dev, c->h_eval_pin, 0));
    std::memset(c->h_eval_pin, 0, doubles * sizeof(double));
    c->eval_pin_cap = doubles;
    return ISDF_OK;
}

// ---- host-direct steps -----------------------------------------------------------------------------------------------
static bool direct_enabled(const isdf_ctx *c) {
    return !c->env_no_direct && !c->prof_on && c->world == 1 && c->cfg.variant != ISDF_V1_SWEPT && c->cfg.enable_pos;
}
// pinned, device-mapped [inputs 19 n nb | outputs (1 + 19 n) nb | nb flags]
// Results that a kernel stores into host-mapped memory, and the completion word it stores after them, are separate PCIe writes issued
// by different wavefronts; "release, then the word" orders them for the DEVICE's view of memory, not for the order in which posted
// writes become visible to the CPU.  Observed on MI355X: the first host-direct step of a fresh ctx, about one process in twenty when two
// processes share the GPU - word and cost there, all gradient rows still the zeros of the fresh allocation, a wrong gradient returned
// without any flag (tests/native/xchg_fail_worker.py caught it as a "wrong" reference).  So the word only says the kernel is done:
// the result area is filled with a pattern no result can have (all ones: not the canonical NaN, not a count) before the launch, and
// after the word the host waits until none of it is left - normally a scan of a few hundred doubles that finds nothing.
static void host_rows_mark(double *p, size_t n) { std::memset((void *)p, 0xFF, n * sizeof(double)); }
static bool host_rows_wait(isdf_ctx *c, const double *p, size_t n, bool another_area_of_the_same_step = false) {
    const volatile unsigned long long *w = (const volatile unsigned long long *)p;
    if (!another_area_of_the_same_step) c->host_steps++;
    bool late = false;
    const auto t0 = std::chrono::steady_clock::now();
    for (size_t i = 0; i < n; i++) {
        for (unsigned spin = 0; w[i] == ~0ull; spin++) {
            late = true;
            c->host_late_spins++;
            if ((spin & 0x3FFFu) == 0x3FFFu && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 5.0) return false;
        }
    }
    if (late) { if (!another_area_of_the_same_step || c->host_late_mark != c->host_steps) { c->host_late++; c->host_late_mark = c->host_steps; } std::atomic_thread_fence(std::memory_order_acquire); }
    return true;
}

static int direct_reserve(isdf_ctx *c, int nb, int n) {
    const size_t in = (size_t)19 * n * nb, out = isdf_out_stride(n) * nb, need = in + out + (size_t)nb;
    if (c->dir_cap < need) {
        if (c->h_dir) (void)hipHostFree(c->h_dir);
        c->h_dir = nullptr; c->dir_cap = 0;
        HIPCHK(c, hipHostMalloc((void **)&c->h_dir, need * sizeof(double), hipHostMallocDefault));      // pinned host memory is device-visible (unified addressing)
        HIPCHK(c, hipHostGetDevicePointer((void **)&c->h_dir_dev, c->h_dir, 0));
        c->dir_cap = need;
        std::memset(c->h_dir, 0, need * sizeof(double));
    }
    c->dir_in = in; c->dir_out = out; c->dir_flags = (size_t)nb;
    return ISDF_OK;
}
// Can the host write device memory directly (large PCIe BAR)?  Verified once per ctx THE WAY THE STEPS USE IT: a kernel reads
// the buffer (its lines may now sit in an L2) and raises a host-mapped flag; the host, which waited for nothing but that flag,
// overwrites the buffer through the BAR and launches the kernel again without any host-side synchronisation in between; the
// second launch must see the second pattern (and the first launch the first).  Writes only - host READS over the BAR cost
// ~1 us per access.
__global__ void bar_probe_kernel(const double *buf, int m, double *copy, volatile unsigned long long *host_flag, unsigned long long seq) {
    for (int i = threadIdx.x; i < m; i += blockDim.x) copy[i] = buf[i];
    __syncthreads();
    if (threadIdx.x == 0) { __threadfence_system(); *host_flag = seq; }
}
static bool bar_usable(isdf_ctx *c, double *d_buf, size_t n) {
    if (c->bar_state != 0) return c->bar_state > 0;
    c->bar_state = -1;
    if (c->env_no_bar) return false;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, c->device) != hipSuccess || !prop.isLargeBar) return false;
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, d_buf) != hipSuccess) { (void)hipGetLastError(); return false; }
    const int m = (int)(n < 64 ? n : 64);
    double *d_copy = nullptr;
    unsigned long long *h_flag = nullptr, *h_flag_dev = nullptr;
    bool ok = hipMalloc(&d_copy, 2 * 64 * sizeof(double)) == hipSuccess &&
              hipHostMalloc((void **)&h_flag, 64, hipHostMallocDefault) == hipSuccess &&
              hipHostGetDevicePointer((void **)&h_flag_dev, h_flag, 0) == hipSuccess;
    std::vector<double> pat(2 * (size_t)m), back(2 * (size_t)m, 0.0);
    for (int i = 0; i < m; i++) { pat[i] = 1.0 + (double)i * 0.5; pat[m + i] = -3.0 - (double)i * 0.25; }
    for (int round = 0; ok && round < 2; round++) {
        *(volatile unsigned long long *)h_flag = 0ull;
        std::memcpy(d_buf, pat.data() + (size_t)round * m, (size_t)m * sizeof(double));          // CPU stores into device memory
        __sync_synchronize();
        hipLaunchKernelGGL(bar_probe_kernel, dim3(1), dim3(64), 0, c->stream, d_buf, m, d_copy + (size_t)round * 64, h_flag_dev, (unsigned long long)(round + 1));
        ok = hipGetLastError() == hipSuccess;
        const auto t0 = std::chrono::steady_clock::now();
        while (ok && *(volatile unsigned long long *)h_flag != (unsigned long long)(round + 1))       // the steps' own hand-over: no stream sync
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2.0) ok = false;
    }
    if (ok) ok = hipMemcpy(back.data(), d_copy, (size_t)m * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess &&
                 hipMemcpy(back.data() + m, d_copy + 64, (size_t)m * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess &&
                 std::memcmp(pat.data(), back.data(), 2 * (size_t)m * sizeof(double)) == 0;
    (void)hipStreamSynchronize(c->stream);
    (void)hipGetLastError();
    if (d_copy) (void)hipFree(d_copy);
    if (h_flag) (void)hipHostFree(h_flag);
    if (ok) c->bar_state = 1;
    return ok;
}
// Places the inputs of nb trajectories of n pieces for a host-direct step and launches it.  ISDF_DIRECT_NA: not applicable
// (the step is not one fused launch), nothing launched.  T / coeffs: per-trajectory host arrays (trajectory `first` onwards).
static int direct_launch(isdf_ctx *c, int nb, int n, const double *const *T, const double *const *coeffs, int first, hipStream_t st, int mode) {
    int rc = direct_reserve(c, nb, n);
    if (rc) return rc;
    const size_t in_all = (size_t)19 * n * nb;
    rc = ensure(c, &c->d_in, &c->in_cap, in_all);
    if (rc) return rc;
    HostDirect hd;
    hd.via_bar = bar_usable(c, c->d_in, in_all);
    // inputs: through the BAR straight into device memory (posted writes, done before the doorbell is rung), or into the
    // pinned buffer the first workgroups of the launch read
    double *dst = hd.via_bar ? c->d_in : c->h_dir;
    for (int b = 0; b < nb; b++) {
        std::memcpy(dst + (size_t)b * n, T[first + b], (size_t)n * sizeof(double));
        std::memcpy(dst + (size_t)n * nb + (size_t)b * 18 * n, coeffs[first + b], (size_t)18 * n * sizeof(double));
    }
    host_rows_mark(c->h_dir + c->dir_in, c->dir_out);            // (the step STORES its sums there; it never reads them)
    __sync_synchronize();
    hd.T = c->h_dir_dev; hd.coeffs = c->h_dir_dev + (size_t)n * nb;
    hd.out = c->h_dir_dev + c->dir_in;
    hd.flags = (unsigned long long *)(c->h_dir_dev + c->dir_in + c->dir_out);
    hd.seq = ++c->dir_seq;
    rc = eval_device_impl(c, nb, n, c->d_in, c->d_in + (size_t)n * nb, hd.out, nullptr, st, mode, false, &hd);
    if (rc == ISDF_OK) { c->dir_pending = true; c->dir_nb = nb; c->dir_n = n; c->last_host_path = hd.via_bar ? ISDF_HOST_PATH_DIRECT_BAR : ISDF_HOST_PATH_DIRECT_MAPPED; }
    return rc;
}
// the host's side of the hand-over: spin on the trajectories' flags (the launch stores them last); bounded - a launch that
// never finishes is reported, not waited for forever
static int direct_wait(isdf_ctx *c, hipStream_t st, bool *overflow) {
    c->dir_pending = false;
    volatile unsigned long long *flags = (volatile unsigned long long *)(c->h_dir + c->dir_in + c->dir_out);
    const unsigned long long seq = c->dir_seq;
    const auto t0 = std::chrono::steady_clock::now();
    *overflow = false;
    for (int b = 0; b < c->dir_nb; b++) {
        unsigned long long f;
        for (unsigned spin = 0;; spin++) {
            f = flags[b];
            if ((f & ~HOST_FLAG_OVERFLOW) == seq) break;
            if ((spin & 0x3FFFu) == 0x3FFFu && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 5.0) {
                (void)hipStreamSynchronize(st);
                return fail(c, ISDF_ERR_HIP, "host-direct step did not complete (flag never arrived)");
            }
        }
        if (f & HOST_FLAG_OVERFLOW) *overflow = true;
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    if (!host_rows_wait(c, c->h_dir + c->dir_in, (size_t)c->dir_nb * isdf_out_stride(c->dir_n))) {
        (void)hipStreamSynchronize(st);
        return fail(c, ISDF_ERR_HIP, "host-direct step: its completion word arrived but not all of its results");
    }
    return ISDF_OK;
}

// ---- host-direct form of the steps that are SEVERAL launches (the swept-volume sweep: prepare, scan, descent, back-prop, reduce):
// the inputs go down through the PCIe BAR, the launches run as ever, and one small kernel behind them copies [cost | gradT |
// gradC], the statistics words and lastTstar into host-mapped memory and raises a flag there - no copy commands (each a DMA
// packet with its own completion), no stream synchronisation (a scheduler wake-up).
__global__ __launch_bounds__(1024) void publish_kernel(const double *out, size_t count, const unsigned long long *stats, const double *tstar, int M,
                                                       double *h_out, unsigned long long *h_stats, double *h_tstar, unsigned long long *h_flag, unsigned long long seq) {
    for (size_t i = threadIdx.x; i < count; i += blockDim.x) h_out[i] = out[i];
    if (threadIdx.x < 8) h_stats[threadIdx.x] = stats[threadIdx.x];
    if (tstar) for (int i = threadIdx.x; i < M; i += blockDim.x) h_tstar[i] = tstar[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(h_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
static bool v1_direct_enabled(const isdf_ctx *c) {
    return !c->env_no_direct && !c->prof_on && c->world == 1 && c->peers.empty() && !c->rccl_comm && c->cfg.variant == ISDF_V1_SWEPT;
}
// returns ISDF_DIRECT_NA when the host cannot write device memory (no large BAR): the copy path then
static int v1_direct_eval(isdf_ctx *c, int n, const double *T, const double *coeffs, double *tstar_inout, const double **h_out, const unsigned long long **h_stat) {
    const size_t in_all = (size_t)19 * n, ostride = isdf_out_stride(n);
    const bool ts = tstar_inout && c->M > 0;
    // d_in: [T | coeffs | lastTstar] - CPU-written, GPU-read only.  (lastTstar does NOT go straight into d_tstar: the GPU itself wrote
    // that array in the step before, and CPU stores through the BAR into memory the device has written are outside what bar_usable
    // probes; the prepare kernel copies the staged values over, SweptParams::tstar_stage)
    int rc = ensure(c, &c->d_in, &c->in_cap, in_all + (size_t)(ts ? c->M : 0));
    if (rc) return rc;
    rc = ensure(c, &c->d_out, &c->out_cap, ostride);
    if (rc) return rc;
    if (!bar_usable(c, c->d_in, in_all + (size_t)(ts ? c->M : 0))) return ISDF_DIRECT_NA;
    // pinned, device-mapped: [out | 8 statistics words | flag | lastTstar]
    const size_t need = ostride + 8 + 2 + (size_t)(ts ? c->M : 0);
    if (c->v1_pin_cap < need) {
        if (c->h_v1_pin) (void)hipHostFree(c->h_v1_pin);
        c->h_v1_pin = nullptr; c->v1_pin_cap = 0;
        HIPCHK(c, hipHostMalloc((void **)&c->h_v1_pin, need * sizeof(double), hipHostMallocDefault));
        HIPCHK(c, hipHostGetDevicePointer((void **)&c->h_v1_pin_dev, c->h_v1_pin, 0));
        std::memset(c->h_v1_pin, 0, need * sizeof(double));
        c->v1_pin_cap = need;
    }
    std::memcpy(c->d_in, T, (size_t)n * sizeof(double));                       // CPU stores into device memory
    std::memcpy(c->d_in + n, coeffs, (size_t)18 * n * sizeof(double));
    if (ts) { std::memcpy(c->d_in + in_all, tstar_inout, (size_t)c->M * sizeof(double)); c->v1_tstar_stage = c->d_in + in_all; }
    __sync_synchronize();
    rc = eval_device_impl(c, 1, n, c->d_in, c->d_in + n, c->d_out, ts ? c->d_tstar : nullptr, c->stream, 0);
    c->v1_tstar_stage = nullptr;
    if (rc) return rc;
    const unsigned long long seq = ++c->dir_seq;
    host_rows_mark(c->h_v1_pin, ostride);
    if (ts) host_rows_mark(c->h_v1_pin + ostride + 10, (size_t)c->M);
    __sync_synchronize();
    double *ho = c->h_v1_pin_dev;
    unsigned long long *hs = (unsigned long long *)(c->h_v1_pin_dev + ostride);
    hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(1024), 0, c->stream, c->d_out, ostride, c->d_stats, ts ? c->d_tstar : nullptr, c->M,
                       ho, hs, c->h_v1_pin_dev + ostride + 10, hs + 8, seq);
    HIPCHK(c, hipGetLastError());
    volatile unsigned long long *flag = (volatile unsigned long long *)(c->h_v1_pin + ostride) + 8;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 0; *flag != seq; spin++) {
        if ((spin & 0x3FFFu) == 0x3FFFu && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 10.0) {
            (void)hipStreamSynchronize(c->stream);
            if (*flag == seq) break;
            return fail(c, ISDF_ERR_HIP, "host-direct swept-volume step did not complete (its flag never arrived)");
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    if (!host_rows_wait(c, c->h_v1_pin, ostride) || (ts && !host_rows_wait(c, c->h_v1_pin + ostride + 10, (size_t)c->M, true))) {
        (void)hipStreamSynchronize(c->stream);
        return fail(c, ISDF_ERR_HIP, "host-direct swept-volume step: its flag arrived but not all of its results");
    }
    if (ts) std::memcpy(tstar_inout, c->h_v1_pin + ostride + 10, (size_t)c->M * sizeof(double));
    *h_out = c->h_v1_pin;
    *h_stat = (const unsigned long long *)(c->h_v1_pin + ostride);
    c->last_host_path = ISDF_HOST_PATH_DIRECT_BAR;
    return ISDF_OK;
}

extern "C" int isdf_mesh_info(const isdf_ctx *c, int info_out[16]) {
    if (!c || !info_out) return ISDF_ERR_INVALID_ARG;
    for (int k = 0; k < 16; k++) info_out[k] = c->mesh_info[k];
    return ISDF_OK;
}

extern "C" int isdf_host_path(const isdf_ctx *c) { return c ? c->last_host_path : ISDF_ERR_INVALID_ARG; }
extern "C" int isdf_host_info(const isdf_ctx *c, int64_t info_out[8]) {
    if (!c || !info_out) return ISDF_ERR_INVALID_ARG;
    for (int k = 0; k < 8; k++) info_out[k] = 0;
    info_out[0] = (int64_t)c->host_steps; info_out[1] = (int64_t)c->host_late; info_out[2] = (int64_t)c->host_late_spins;
    return ISDF_OK;
}

extern "C" int isdf_eval(isdf_ctx *c, int n_traj, const int *N, const double *const *T, const double *const *coeffs,
                         double *cost_inout, double *const *gradT_inout, double *const *gradC_inout, double *tstar_inout) {
    if (!c) return ISDF_ERR_INVALID_ARG;
    if (n_traj < 1 || !N || !T || !coeffs || !cost_inout || !gradT_inout || !gradC_inout) return fail(c, ISDF_ERR_INVALID_ARG, "null argument");
    HIPCHK(c, hipSetDevice(c->device));
    // trajectories with equal N go out as one batched launch; ragged input is evaluated group by group.
    // Per group: inputs gathered into one pinned buffer, ONE upload, the launches, ONE download (outputs + the overflow
    // word), ONE synchronisation.  The pair statistics are computed on demand (isdf_get_stats) unless the input is ragged.
    int start = 0, groups = 0;
    isdf_stats total{};
    bool overflow = false;
    for (int s0 = 0; s0 < n_traj;) { int e = s0 + 1; while (e < n_traj && N[e] == N[s0]) e++; groups++; s0 = e; }
    while (start < n_traj) {
        int end = start + 1;
        while (end < n_traj && N[end] == N[start]) end++;
        const int nb = end - start, n = N[start];
        if (n < 1) return fail(c, ISDF_ERR_INVALID_ARG, "N must be >= 1");
        const size_t in_per = (size_t)19 * n, ostride = isdf_out_stride(n);
        const size_t in_all = in_per * nb, out_all = ostride * nb;
        for (int b = 0; b < nb; b++)
            if (!T[start + b] || !coeffs[start + b] || !gradT_inout[start + b] || !gradC_inout[start + b])
                return fail(c, ISDF_ERR_INVALID_ARG, "null trajectory buffer");
        if (direct_enabled(c)) {
            // ONE launch, no copy commands, no stream synchronisation: the inputs go straight into device memory through the
            // PCIe BAR (or are fetched from host-mapped memory by the launch itself), the launch copies [cost | gradT | gradC]
            // and a completion flag per trajectory into host-mapped memory (csrc/tile_sweep.hip)
            int rcd = direct_launch(c, nb, n, T, coeffs, start, c->stream, 0);
            if (rcd < 0) return rcd;
            if (rcd == ISDF_OK) {
                bool ovf = false;
                rcd = direct_wait(c, c->stream, &ovf);
                if (rcd) return rcd;
                if (ovf) {
                    overflow = true;
                    HIPCHK(c, hipMemset(c->d_stats + 4, 0, sizeof(unsigned long long)));
                    const int rr = isdf_reset_result_slots(c);
                    if (rr) return rr;
                }
                const double *hout = c->h_dir + c->dir_in;
                for (int b = 0; b < nb; b++) {
                    const double *o = hout + ostride * b;
                    cost_inout[start + b] += o[0];
                    double *gT = gradT_inout[start + b], *gC = gradC_inout[start + b];
                    for (int i = 0; i < n; i++) gT[i] += o[1 + i];
                    for (int i = 0; i < 18 * n; i++) gC[i] += o[1 + n + i];
                }
                if (groups > 1) {
                    rcd = fetch_stats(c);
                    if (rcd) return rcd;
                    total.n_units += c->last_stats.n_units; total.n_units_culled += c->last_stats.n_units_culled;
                    total.n_pairs += c->last_stats.n_pairs; total.n_grad_pairs += c->last_stats.n_grad_pairs;
                }
                start = end;
                continue;
            }
            // ISDF_DIRECT_NA: this step is not one fused launch - the copy path below
        }
        if (nb == 1 && v1_direct_enabled(c)) {
            const double *ho = nullptr; const unsigned long long *hs = nullptr;
            const int rcv = v1_direct_eval(c, n, T[start], coeffs[start], tstar_inout, &ho, &hs);
            if (rcv < 0) return rcv;
            if (rcv == ISDF_OK) {
                if (hs[4]) {
                    overflow = true;
                    HIPCHK(c, hipMemset(c->d_stats + 4, 0, sizeof(unsigned long long)));
                    const int rr = isdf_reset_result_slots(c);
                    if (rr) return rr;
                }
                total.n_units += (int64_t)hs[0]; total.n_units_culled += (int64_t)hs[1];
                total.n_pairs += (int64_t)hs[2]; total.n_grad_pairs += (int64_t)hs[3];
                cost_inout[start] += ho[0];
                double *gT = gradT_inout[start], *gC = gradC_inout[start];
                for (int i = 0; i < n; i++) gT[i] += ho[1 + i];
                for (int i = 0; i < 18 * n; i++) gC[i] += ho[1 + n + i];
                start = end;
                continue;
            }
        }
        c->last_host_path = ISDF_HOST_PATH_COPY;
        int rc = ensure(c, &c->d_in, &c->in_cap, in_all);
        if (rc) return rc;
        rc = ensure(c, &c->d_out, &c->out_cap, out_all);
        if (rc) return rc;
        rc = ensure_eval_pin(c, in_all + out_all + 16);
        if (rc) return rc;
        double *hin = c->h_eval_pin, *hout = c->h_eval_pin + in_all;
        unsigned long long *hstat = (unsigned long long *)(c->h_eval_pin + in_all + out_all);
        double *dT = c->d_in, *dC = c->d_in + (size_t)n * nb;
        for (int b = 0; b < nb; b++) {
            if (!T[start + b] || !coeffs[start + b] || !gradT_inout[start + b] || !gradC_inout[start + b])
                return fail(c, ISDF_ERR_INVALID_ARG, "null trajectory buffer");
            std::memcpy(hin + (size_t)b * n, T[start + b], (size_t)n * sizeof(double));
            std::memcpy(hin + (size_t)n * nb + (size_t)b * 18 * n, coeffs[start + b], (size_t)18 * n * sizeof(double));
        }
        HIPCHK(c, hipMemcpyAsync(c->d_in, hin, in_all * sizeof(double), hipMemcpyHostToDevice, c->stream));
        double *dts = nullptr;
        if (c->cfg.variant == ISDF_V1_SWEPT && tstar_inout && c->M > 0) {
            dts = c->d_tstar;
            HIPCHK(c, hipMemcpyAsync(dts, tstar_inout, (size_t)c->M * sizeof(double), hipMemcpyHostToDevice, c->stream));
        }
        // a multi-device step ends in a sum kernel on the lead: it writes straight into the pinned buffer and raises a word there
        const bool host_out = (!c->peers.empty() || c->rccl_comm) && !dts && !c->env_multi_no_hostout && !c->prof_on;
        if (host_out) {
            if (!c->d_msum_blocks) { HIPCHK(c, hipMalloc(&c->d_msum_blocks, sizeof(unsigned))); HIPCHK(c, hipMemset(c->d_msum_blocks, 0, sizeof(unsigned))); }
            c->mh_words = (unsigned long long *)(c->h_eval_pin_dev + in_all + out_all);
            c->mh_seq++;
            host_rows_mark(hout, out_all);                       // (see host_rows_wait: the word does not order the results for the CPU)
            host_rows_mark((double *)hstat, 8);
            __sync_synchronize();
        }
        rc = sweep_dispatch(c, nb, n, dT, dC, host_out ? c->h_eval_pin_dev + in_all : c->d_out, dts, c->stream);
        if (rc) { c->mh_words = nullptr; return rc; }
        if (host_out) {
            volatile unsigned long long *word = hstat + 8;
            const auto t0 = std::chrono::steady_clock::now();
            for (unsigned spin = 0; *word != c->mh_seq; spin++) {
                if ((spin & 0x3FFFu) == 0x3FFFu && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 5.0) {
                    (void)hipStreamSynchronize(c->stream);
                    if (*word == c->mh_seq) break;
                    return fail(c, ISDF_ERR_HIP, "multi-device step did not complete (its completion word never arrived)");
                }
            }
            std::atomic_thread_fence(std::memory_order_acquire);
            if (!host_rows_wait(c, hout, out_all) || !host_rows_wait(c, (const double *)hstat, 8, true)) {
                (void)hipStreamSynchronize(c->stream);
                return fail(c, ISDF_ERR_HIP, "multi-device step: its completion word arrived but not all of its results");
            }
        } else {
        HIPCHK(c, hipMemcpyAsync(hout, c->d_out, out_all * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(hstat, c->d_stats, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
        if (dts) HIPCHK(c, hipMemcpyAsync(tstar_inout, dts, (size_t)c->M * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        }
        if (hstat[4]) {   // sticky until read; slots a late producer may still fill are emptied again
            overflow = true;
            HIPCHK(c, hipMemset(c->d_stats + 4, 0, sizeof(unsigned long long)));
            const int rr = isdf_reset_result_slots(c);
            if (rr) return rr;
        }
        if (c->cfg.variant == ISDF_V1_SWEPT) {       // the V1 kernels count straight into the statistics words
            total.n_units += (int64_t)hstat[0]; total.n_units_culled += (int64_t)hstat[1];
            total.n_pairs += (int64_t)hstat[2]; total.n_grad_pairs += (int64_t)hstat[3];
        } else if (groups > 1) {
            rc = fetch_stats(c);                 // ragged input: the per-group counts have to be added up now
            if (rc == ISDF_OK) rc = add_peer_stats(c);
            if (rc) return rc;
            total.n_units += c->last_stats.n_units; total.n_units_culled += c->last_stats.n_units_culled;
            total.n_pairs += c->last_stats.n_pairs; total.n_grad_pairs += c->last_stats.n_grad_pairs;
        }
        for (int b = 0; b < nb; b++) {
            const double *o = hout + ostride * b;
            cost_inout[start + b] += o[0];
            double *gT = gradT_inout[start + b], *gC = gradC_inout[start + b];
            for (int i = 0; i < n; i++) gT[i] += o[1 + i];
            for (int i = 0; i < 18 * n; i++) gC[i] += o[1 + n + i];
        }
        start = end;
    }
    if (groups > 1 || c->cfg.variant == ISDF_V1_SWEPT) {
        total.overflow = overflow ? 1 : 0;
        c->last_stats = total;
        c->stats_cached = true;
    } else {
        c->stats_cached = false;                 // isdf_get_stats counts the pairs of the last launch when asked
    }
    if (overflow) return fail(c, ISDF_ERR_OVERFLOW, "a bounded device work list overflowed; result invalid");
    return ISDF_OK;
}

// --------------------------------------------------------------------------------------------------------------
// full objective callback: TrajOptimizer::costFunctionLmbm (back_end_optimizer.hpp:358-430)
// --------------------------------------------------------------------------------------------------------------
extern "C" int isdf_set_trajectory(isdf_ctx *c, int N, const double head_pva[9], const double tail_pva[9], double rho) {
    if (!c) return ISDF_ERR_INVALID_ARG;
    if (N < 1 || !head_pva || !tail_pva) return fail(c, ISDF_ERR_INVALID_ARG, "bad trajectory setup");
    c->minco.set_conditions(head_pva, tail_pva, N);
    c->rho = rho;
    c->have_traj = true;
    std::memcpy(c->cb_ends, head_pva, 9 * sizeof(double)); std::memcpy(c->cb_ends + 9, tail_pva, 9 * sizeof(double));
    c->cb_ends_dirty = true;
    c->cb_T.assign(N, 0.0); c->cb_gdC.assign((size_t)18 * N, 0.0); c->cb_gdT.assign(N, 0.0);
    c->cb_gradP.assign((size_t)3 * (N > 1 ? N - 1 : 1), 0.0); c->cb_gradT.assign(N, 0.0);
    return ISDF_OK;
}

extern "C" int isdf_num_variables(const isdf_ctx *c) { return (c && c->have_traj) ? c->minco.N + 3 * (c->minco.N - 1) : 0; }

// x = [tau(N) | waypoints 3(N-1)]: backwardT / backwardP of optimize_traj_lmbm (back_end_optimizer.cpp:22-28)
extern "C" int isdf_pack_variables(isdf_ctx *c, const double *T, const double *waypoints, double *x) {
    if (!c || !T || !x) return ISDF_ERR_INVALID_ARG;
    if (!c->have_traj) return fail(c, ISDF_ERR_STATE, "isdf_set_trajectory not called");
    const int N = c->minco.N;
    if (N > 1 && !waypoints) return fail(c, ISDF_ERR_INVALID_ARG, "null waypoints");
    for (int i = 0; i < N; i++) {
        if (!(T[i] > 0.0)) return fail(c, ISDF_ERR_INVALID_ARG, "durations must be positive");
        x[i] = isdf_host::T_to_tau(T[i]);
    }
    for (int i = 0; i < 3 * (N - 1); i++) x[N + i] = waypoints[i];
    return ISDF_OK;
}

// forwardT / forwardP + minco.setParameters: the trajectory x stands for (T: N, coeffs: 6N x 3 column-major)
extern "C" int isdf_unpack_variables(isdf_ctx *c, const double *x, double *T, double *coeffs) {
    if (!c || !x) return ISDF_ERR_INVALID_ARG;
    if (!c->have_traj) return fail(c, ISDF_ERR_STATE, "isdf_set_trajectory not called");
    const int N = c->minco.N;
    for (int i = 0; i < N; i++) c->cb_T[i] = isdf_host::tau_to_T(x[i]);
    c->minco.set_parameters(x + N, c->cb_T.data());
    if (T) std::memcpy(T, c->cb_T.data(), (size_t)N * sizeof(double));
    if (coeffs) std::memcpy(coeffs, c->minco.c.data(), (size_t)18 * N * sizeof(double));
    return ISDF_OK;
}

// First half of the callback: tau -> T, MINCO, energy, and the sweeps queued on `st`.  Leaves this rank's partial sums
// ([cost | gradT | gradC] per sweep, cb_n_out blocks) in c->d_cb + 19N on the device.
// ---- the callback with its MINCO half on the device (csrc/minco_dev.hip): x goes down (through the PCIe BAR when the host can
// write device memory, else the first kernel fetches it from host-mapped memory), cb_pre_kernel writes (T, coefficients) where
// the sweeps read them, the sweeps accumulate as ever, cb_post_kernel leaves (cost, g, the four cost parts) and a completion
// word in host-mapped memory.  No copy commands, no stream synchronisation.
static size_t cb_res_stride(int N) { return (size_t)1 + (size_t)(N + 3 * (N - 1)) + 4; }
static int cb_dev_fill(isdf_ctx *c, int N, CbDev *P, hipStream_t st) {
    const size_t nvar = (size_t)N + 3 * (size_t)(N - 1), ostride = isdf_out_stride(N), in_len = (size_t)19 * N;
    // device: [x | ends | u | energy block]
    const size_t off_ends = nvar, off_u = off_ends + 18, off_e = off_u + (size_t)6 * (N + 1);
    const size_t need = off_e + ostride;
    if (c->cbdev_cap < need) { c->cb_ends_dirty = true; }
    int rc = ensure(c, &c->d_cbdev, &c->cbdev_cap, need);
    if (rc) return rc;
    rc = ensure(c, &c->d_cb, &c->cb_cap, in_len + 2 * ostride);
    if (rc) return rc;
    const size_t rs = cb_res_stride(N), pin_need = nvar + rs + 2;
    if (c->cbres_cap < pin_need) {
        if (c->h_cbres) (void)hipHostFree(c->h_cbres);
        c->h_cbres = nullptr; c->cbres_cap = 0;
        HIPCHK(c, hipHostMalloc((void **)&c->h_cbres, pin_need * sizeof(double), hipHostMallocDefault));
        HIPCHK(c, hipHostGetDevicePointer((void **)&c->h_cbres_dev, c->h_cbres, 0));
        std::memset(c->h_cbres, 0, pin_need * sizeof(double));
        c->cbres_cap = pin_need;
    }
    if (c->cb_ends_dirty) {
        HIPCHK(c, hipMemcpyAsync(c->d_cbdev + off_ends, c->cb_ends, 18 * sizeof(double), hipMemcpyHostToDevice, st));
        HIPCHK(c, hipStreamSynchronize(st));            // (cb_ends may change before an asynchronous copy has read it)
        c->cb_ends_dirty = false;
    }
    P->N = N; P->nb = 1; P->n_out = c->cb_n_out; P->res_stride = (int)rs;
    P->x = c->d_cbdev; P->ends = c->d_cbdev + off_ends; P->u = c->d_cbdev + off_u; P->epart = c->d_cbdev + off_e;
    P->T = c->d_cb; P->coeffs = c->d_cb + N; P->sweep = c->d_cb + in_len;
    P->rho = c->rho;
    P->res = c->h_cbres_dev + nvar; P->flag = (unsigned long long *)(c->h_cbres_dev + nvar + rs);
    P->seq = c->cb_seq; P->stats = c->d_stats;
    return ISDF_OK;
}
// Where a callback's MINCO half runs.  Forced either way by isdf_set_minco_mode; left to itself (mode 0) the device takes it unless
// the step is the small single-trajectory kind whose sweep is one fused host-direct launch (C2: N <= 64, one GPU, tile sweep) -
// there the host's band LU (10 us at N = 40, growing with N) still beats two more kernels either side of a 16 us launch
// (measured: 35 us against 38; profiles/r5_callback_bench.txt), everywhere else the device form is the faster one
// (N = 400: 135 us against 216; swept-volume ctx; the batched optimiser's rounds).
static bool cb_device_minco(const isdf_ctx *c) {
    if (!c->have_traj || c->minco.N > CB_MAX_N || c->prof_on || c->minco_mode == 1) return false;
    if (c->minco_mode == 2) return true;
    const bool small_fused = c->cfg.variant != ISDF_V1_SWEPT && direct_enabled(c) && c->peers.empty() && c->minco.N <= CB_AUTO_HOST_MAX_N;
    return !small_fused;
}

static int cost_function_launch_dev(isdf_ctx *c, const double *x, int n, hipStream_t st, bool whole) {
    const int N = c->minco.N;
    const bool swept = c->cfg.variant == ISDF_V1_SWEPT;
    c->cb_n_out = swept ? 2 : 1;
    c->cb_direct = false; c->cb_dev = true; c->cb_post_queued = false;
    c->cb_seq = ++c->dir_seq;          // (one counter with the host-direct steps: they share the staging buffer's ready flags)
    CbDev P{};
    int rc = cb_dev_fill(c, N, &P, st);
    if (rc) return rc;
    const size_t nvar = (size_t)n;
    if (bar_usable(c, c->d_cbdev, nvar)) std::memcpy(c->d_cbdev, x, nvar * sizeof(double));      // CPU stores into device memory
    else { std::memcpy(c->h_cbres, x, nvar * sizeof(double)); P.x = c->h_cbres_dev; }
    host_rows_mark(c->h_cbres + nvar, cb_res_stride(N));
    __sync_synchronize();
    const size_t ostride = isdf_out_stride(N);
    double *d_T = c->d_cb, *d_C = c->d_cb + N, *d_o = c->d_cb + (size_t)19 * N;
    launch_cb_pre(P, st);
    if (swept) {
        rc = sweep_dispatch(c, 1, N, d_T, d_C, d_o, nullptr, st, 1);
        if (rc) return rc;
        rc = sweep_dispatch(c, 1, N, d_T, d_C, d_o + ostride, nullptr, st, 2);
    } else {
        rc = sweep_dispatch(c, 1, N, d_T, d_C, d_o, nullptr, st, 0);
    }
    if (rc) return rc;
    if (whole) { launch_cb_post(P, st); c->cb_post_queued = true; }      // the split form queues it in _finish, behind the caller's all-reduce
    HIPCHK(c, hipGetLastError());
    c->last_host_path = ISDF_HOST_PATH_DEVICE_CALLBACK;
    c->last_minco_path = 1;
    c->cb_pending = true;
    return ISDF_OK;
}
static int cost_function_finish_dev(isdf_ctx *c, double *g, double *cost_out, hipStream_t st) {
    c->cb_pending = false;
    const int N = c->minco.N;
    const size_t nvar = (size_t)N + 3 * (size_t)(N - 1), rs = cb_res_stride(N);
    if (!c->cb_post_queued) {
        CbDev P{};
        const int rc = cb_dev_fill(c, N, &P, st);
        if (rc) return rc;
        if (!bar_usable(c, c->d_cbdev, nvar)) P.x = c->h_cbres_dev;
        launch_cb_post(P, st);
        HIPCHK(c, hipGetLastError());
    }
    volatile unsigned long long *flag = (volatile unsigned long long *)(c->h_cbres + nvar + rs);
    const auto t0 = std::chrono::steady_clock::now();
    unsigned long long f;
    for (unsigned spin = 0;; spin++) {
        f = *flag;
        if ((f & ~HOST_FLAG_OVERFLOW) == c->cb_seq) break;
        if ((spin & 0x3FFFu) == 0x3FFFu && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 5.0) {
            (void)hipStreamSynchronize(st);
            if ((*flag & ~HOST_FLAG_OVERFLOW) == c->cb_seq) { f = *flag; break; }
            return fail(c, ISDF_ERR_HIP, "device callback did not complete (its completion word never arrived)");
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    if (f & HOST_FLAG_OVERFLOW) {
        (void)hipMemset(c->d_stats + 4, 0, sizeof(unsigned long long));
        (void)isdf_reset_result_slots(c);
        return fail(c, ISDF_ERR_OVERFLOW, "a bounded device work list overflowed; result invalid");
    }
    const double *res = c->h_cbres + nvar;
    if (!host_rows_wait(c, res, rs)) {
        (void)hipStreamSynchronize(st);
        return fail(c, ISDF_ERR_HIP, "device callback: its completion word arrived but not all of its results");
    }
    *cost_out = res[0];
    std::memcpy(g, res + 1, nvar * sizeof(double));
    for (int q = 0; q < 4; q++) c->last_parts[q] = res[1 + nvar + q];
    return ISDF_OK;
}

static int cost_function_launch(isdf_ctx *c, const double *x, int n, hipStream_t st, bool allow_direct = false) {
    if (!x) return fail(c, ISDF_ERR_INVALID_ARG, "null argument");
    if (!c->have_traj) return fail(c, ISDF_ERR_STATE, "isdf_set_trajectory not called");
    const int N = c->minco.N;
    if (n != N + 3 * (N - 1)) return fail(c, ISDF_ERR_INVALID_ARG, "n must be N + 3(N-1)");
    HIPCHK(c, hipSetDevice(c->device));
    if (cb_device_minco(c)) return cost_function_launch_dev(c, x, n, st, allow_direct);
    c->cb_dev = false; c->last_minco_path = 0;
    // tau -> T, xi -> waypoints, MINCO coefficients, energy and its partials (:363-381)
    c->cb_x.assign(x, x + n);
    for (int i = 0; i < N; i++) c->cb_T[i] = isdf_host::tau_to_T(x[i]);
    c->minco.set_parameters(x + N, c->cb_T.data());
    c->cb_energy = c->minco.energy(c->cb_gdC.data(), c->cb_gdT.data());
    // the two sweeps on the device (:386-405)
    const size_t in_len = (size_t)19 * N, ostride = isdf_out_stride(N);
    const bool swept = c->cfg.variant == ISDF_V1_SWEPT;
    c->cb_n_out = swept ? 2 : 1;
    c->cb_direct = false;
    if (allow_direct && !swept && direct_enabled(c)) {
        // one launch that reads (T, coefficients) from host-mapped memory and writes the sums back into it: no copy commands,
        // no stream synchronisation (csrc/tile_sweep.hip, host-direct step)
        const double *Tp = c->cb_T.data(), *Cp = c->minco.c.data();
        const int rcd = direct_launch(c, 1, N, &Tp, &Cp, 0, st, 0);
        if (rcd < 0) return rcd;
        if (rcd == ISDF_OK) { c->cb_direct = true; c->cb_pending = true; return ISDF_OK; }
    }
    c->last_host_path = ISDF_HOST_PATH_COPY;
    const size_t need = in_len + c->cb_n_out * ostride;
    if (c->pin_cap < need) {
        if (c->h_pin) (void)hipHostFree(c->h_pin);
        c->h_pin = nullptr; c->pin_cap = 0;
        HIPCHK(c, hipHostMalloc((void **)&c->h_pin, need * sizeof(double), hipHostMallocDefault));
        c->pin_cap = need;
    }
    int rc = ensure(c, &c->d_cb, &c->cb_cap, need);
    if (rc) return rc;
    std::memcpy(c->h_pin, c->cb_T.data(), (size_t)N * sizeof(double));
    std::memcpy(c->h_pin + N, c->minco.c.data(), (size_t)18 * N * sizeof(double));
    HIPCHK(c, hipMemcpyAsync(c->d_cb, c->h_pin, in_len * sizeof(double), hipMemcpyHostToDevice, st));
    double *d_T = c->d_cb, *d_C = c->d_cb + N, *d_o = c->d_cb + in_len;
    if (swept) {
        rc = sweep_dispatch(c, 1, N, d_T, d_C, d_o, nullptr, st, 1);
        if (rc) return rc;
        rc = sweep_dispatch(c, 1, N, d_T, d_C, d_o + ostride, nullptr, st, 2);
    } else {
        rc = sweep_dispatch(c, 1, N, d_T, d_C, d_o, nullptr, st, 0);
    }
    if (rc) return rc;
    c->cb_pending = true;
    return ISDF_OK;
}

// Second half: download the (all-reduced) sums, add them in the reference's order, propagateGrad, time term, chain rule.
static int cost_function_finish(isdf_ctx *c, double *g, double *cost_out, hipStream_t st) {
    if (!g || !cost_out) return fail(c, ISDF_ERR_INVALID_ARG, "null argument");
    if (!c->cb_pending) return fail(c, ISDF_ERR_STATE, "no callback evaluation in flight");
    if (c->cb_dev) { HIPCHK(c, hipSetDevice(c->device)); return cost_function_finish_dev(c, g, cost_out, st); }
    c->cb_pending = false;
    const int N = c->minco.N;
    const size_t in_len = (size_t)19 * N, ostride = isdf_out_stride(N);
    const int n_out = c->cb_n_out;
    const bool swept = n_out == 2;
    const double *res = nullptr;
    if (c->cb_direct) {
        bool ovf = false;
        const int rcd = direct_wait(c, st, &ovf);
        if (rcd) return rcd;
        if (ovf) {
            (void)hipMemset(c->d_stats + 4, 0, sizeof(unsigned long long));
            (void)isdf_reset_result_slots(c);
            return fail(c, ISDF_ERR_OVERFLOW, "a bounded device work list overflowed; result invalid");
        }
        res = c->h_dir + c->dir_in;
    } else {
        HIPCHK(c, hipSetDevice(c->device));
        HIPCHK(c, hipMemcpyAsync(c->h_pin + in_len, c->d_cb + in_len, n_out * ostride * sizeof(double), hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        res = c->h_pin + in_len;
    }
    double cost = c->cb_energy;
    double part[2] = {0.0, 0.0};
    for (int k = 0; k < n_out; k++) {       // swept-volume sweep first, then the integral sweep (:386-405)
        const double *o = res + k * ostride;
        cost += o[0];
        part[k] = o[0];
        for (int i = 0; i < N; i++) c->cb_gdT[i] += o[1 + i];
        for (int i = 0; i < 18 * N; i++) c->cb_gdC[i] += o[1 + N + i];
    }
    // dCost/d(c, T) -> dCost/d(waypoints, T) (:416), time regulariser (:417-420), chain rule to (tau, xi) (:426-427)
    c->minco.propagate_grad(c->cb_gdC.data(), c->cb_gdT.data(), c->cb_gradP.data(), c->cb_gradT.data());
    double tsum = 0.0;
    for (int i = 0; i < N; i++) tsum += c->cb_T[i];
    cost += c->rho * tsum;
    for (int i = 0; i < N; i++) g[i] = isdf_host::grad_T_to_tau(c->cb_x[i], c->cb_gradT[i] + c->rho);
    for (int i = 0; i < 3 * (N - 1); i++) g[N + i] = c->cb_gradP[i];
    c->last_parts[0] = c->cb_energy;
    c->last_parts[1] = swept ? part[0] : 0.0;            // swept-volume sweep
    c->last_parts[2] = swept ? part[1] : part[0];        // integral sweep
    c->last_parts[3] = c->rho * tsum;
    *cost_out = cost;
    return ISDF_OK;
}

extern "C" int isdf_cost_function(isdf_ctx *c, const double *x, double *g, int n, double *cost_out) {
    if (!c) return ISDF_ERR_INVALID_ARG;
    if (!x || !g || !cost_out) return fail(c, ISDF_ERR_INVALID_ARG, "null argument");
    // a sharded ctx returns only this rank's partial sums from the sweeps: the un-split callback (and the drivers built on it)
    // would optimise on them without any error - the split form (_launch / all-reduce / _finish) is the one to use
    if (c->world > 1 && c->peers.empty() && !isdf_xchg_fuse_on(c))
        return fail(c, ISDF_ERR_STATE, "sharded ctx: use isdf_cost_function_launch / _finish around the all-reduce (or switch the in-kernel exchange on)");
    const int rc = cost_function_launch(c, x, n, c->stream, true);
    if (rc) return rc;
    return cost_function_finish(c, g, cost_out, c->stream);
}

// Multi-GPU form (one process per GPU, isdf_set_shard): _launch queues this rank's share of the sweeps on `stream` and
// hands back the device buffer of partial sums; the caller sums it over the ranks IN PLACE on the same stream (one
// all-reduce, RCCL); _finish then yields the same (cost, g) on every rank.
extern "C" int isdf_cost_function_launch(isdf_ctx *c, const double *x, int n, void *stream, double **d_partial_out, size_t *count_out) {
    if (!c) return ISDF_ERR_INVALID_ARG;
    if (!d_partial_out || !count_out) return fail(c, ISDF_ERR_INVALID_ARG, "null argument");
    const int rc = cost_function_launch(c, x, n, (hipStream_t)stream);
    if (rc) return rc;
    *d_partial_out = c->d_cb + (size_t)19 * c->minco.N;       // (both MINCO paths keep the sweeps' sums here)
    *count_out = (size_t)c->cb_n_out * isdf_out_stride(c->minco.N);
    return ISDF_OK;
}
extern "C" int isdf_cost_function_finish(isdf_ctx *c, double *g, double *cost_out, void *stream) {
    if (!c) return ISDF_ERR_INVALID_ARG;
    return cost_function_finish(c, g, cost_out, (hipStream_t)stream);
}

// Same callback with the signature LMBM / the optimizer drivers bind (lmbm_evaluate_t, lmbm.h:206-209):
// instance = isdf_ctx*.  Errors surface as +infinity (the reference has no error channel here).
extern "C" double isdf_cost_function_lmbm(void *instance, const double *x, double *g, const int n) {
    double cost = 0.0;
    const int rc = isdf_cost_function((isdf_ctx *)instance, x, g, n, &cost);
    return rc == ISDF_OK ? cost : INFINITY;
}

// where MINCO runs: 0 = wherever it is faster (cb_device_minco), 1 = on the host (band LU, the reference's elimination order bit
// for bit), 2 = on the device whenever the trajectory fits (N <= 400).  Results agree to rounding (1e-10 relative on the coefficients).
extern "C" int isdf_set_minco_mode(isdf_ctx *c, int mode) {
    if (!c) return ISDF_ERR_INVALID_ARG;
    if (mode < 0 || mode > 2) return fail(c, ISDF_ERR_INVALID_ARG, "minco mode must be 0 (auto), 1 (host) or 2 (device)");
    if (c->cb_pending) return fail(c, ISDF_ERR_STATE, "a callback evaluation is in flight");
    c->minco_mode = mode;
    return ISDF_OK;
}
extern "C" int isdf_minco_path(const isdf_ctx *c) { return c ? c->last_minco_path : ISDF_ERR_INVALID_ARG; }

// energy | swept-volume sweep | integral sweep | rho * sum(T) of the last isdf_cost_function call
extern "C" int isdf_cost_parts(const isdf_ctx *c, double parts[4]) {
    if (!c || !parts) return ISDF_ERR_INVALID_ARG;
    for (int k = 0; k < 4; k++) parts[k] = c->last_parts[k];
    return ISDF_OK;
}

// --------------------------------------------------------------------------------------------------------------
// optimizer driver: L-BFGS behind the callback (lbfgs::lbfgs_optimize, src/utils/include/utils/lbfgs.hpp:480-835)
// --------------------------------------------------------------------------------------------------------------
extern "C" void isdf_lbfgs_params_default(isdf_lbfgs_params *p) {
    if (!p) return;
    const isdf_host::LbfgsParams d;
    p->mem_size = d.mem_size; p->past = d.past; p->max_iterations = d.max_iterations; p->max_linesearch = d.max_linesearch;
    p->weak_wolfe = d.weak_wolfe; p->reference_patches = d.reference_patches;
    p->g_epsilon = d.g_epsilon; p->delta = d.delta; p->min_step = d.min_step; p->max_step = d.max_step;
    p->f_dec_coeff = d.f_dec_coeff; p->s_curv_coeff = d.s_curv_coeff; p->cautious_factor = d.cautious_factor;
    p->machine_prec = d.machine_prec; p->dir_norm_cap = d.dir_norm_cap;
}

static int lbfgs_run(isdf_evaluate_fn evaluate, void *instance, isdf_progress_fn progress, void *progress_instance, double *x_inout, int n,
                     const isdf_lbfgs_params *p, isdf_lbfgs_result *out) {
    if (!evaluate || !x_inout || !p || !out) return ISDF_ERR_INVALID_ARG;
    isdf_host::Lbfgs opt;
    opt.param.mem_size = p->mem_size; opt.param.past = p->past; opt.param.max_iterations = p->max_iterations;
    opt.param.max_linesearch = p->max_linesearch; opt.param.weak_wolfe = p->weak_wolfe; opt.param.reference_patches = p->reference_patches;
    opt.param.g_epsilon = p->g_epsilon; opt.param.delta = p->delta; opt.param.min_step = p->min_step; opt.param.max_step = p->max_step;
    opt.param.f_dec_coeff = p->f_dec_coeff; opt.param.s_curv_coeff = p->s_curv_coeff; opt.param.cautious_factor = p->cautious_factor;
    opt.param.machine_prec = p->machine_prec; opt.param.dir_norm_cap = p->dir_norm_cap;
    opt.evaluate = evaluate;
    opt.instance = instance;
    opt.progress = progress;                  // (isdf_progress_fn == lbfgs_host's lbfgs_progress_fn: lbfgs_progress_t with plain pointers)
    opt.progress_instance = progress_instance;
    const auto t0 = std::chrono::steady_clock::now();
    const isdf_host::LbfgsResult r = opt.minimize(x_inout, n);
    const auto t1 = std::chrono::steady_clock::now();
    out->f = r.f; out->status = r.status; out->iterations = r.iterations; out->evaluations = r.evaluations;
    out->wall_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    return ISDF_OK;
}

extern "C" int isdf_lbfgs_minimize(isdf_evaluate_fn evaluate, void *instance, double *x_inout, int n,
                                   const isdf_lbfgs_params *p, isdf_lbfgs_result *out) {
    return lbfgs_run(evaluate, instance, nullptr, nullptr, x_inout, n, p, out);
}
// ... with the reference's progress / cancel callback (lbfgs_optimize's proc_progress, lbfgs.hpp:256-262,480-492): called once per
// iteration with the SAME instance as evaluate; non-zero return -> status LBFGS_CANCELED (2), x_inout = the iterate it was shown
extern "C" int isdf_lbfgs_minimize_progress(isdf_evaluate_fn evaluate, isdf_progress_fn progress, void *instance, double *x_inout, int n,
                                            const isdf_lbfgs_params *p, isdf_lbfgs_result *out) {
    return lbfgs_run(evaluate, instance, progress, nullptr, x_inout, n, p, out);
}

extern "C" int isdf_set_progress(isdf_ctx *c, isdf_progress_fn progress, void *instance, size_t batch_instance_stride) {
    if (!c) return ISDF_ERR_INVALID_ARG;
    c->progress = progress; c->progress_instance = instance; c->progress_stride = batch_instance_stride;
    return ISDF_OK;
}

extern "C" int isdf_optimize_lbfgs(isdf_ctx *c, double *x_inout, int n, const isdf_lbfgs_params *p, isdf_lbfgs_result *out) {
    if (!c) return ISDF_ERR_INVALID_ARG;
    if (!x_inout || !p || !out) return fail(c, ISDF_ERR_INVALID_ARG, "null argument");
    if (!c->have_traj) return fail(c, ISDF_ERR_STATE, "isdf_set_trajectory not called");
    if (n != isdf_num_variables(c)) return fail(c, ISDF_ERR_INVALID_ARG, "n must be N + 3(N-1)");
    return lbfgs_run(isdf_cost_function_lmbm, c, c->progress, c->progress_instance, x_inout, n, p, out);
}

// --------------------------------------------------------------------------------------------------------------
// instrumentation
// --------------------------------------------------------------------------------------------------------------
extern "C" int isdf_profile_enable(isdf_ctx *c, int on) {
    if (!c) return ISDF_ERR_INVALID_ARG;
    c->prof_on = on != 0;          // on = N > 0: instrument every N-th launch (event records cost GPU time)
    c->prof_secondary = on > 0 && (on & ISDF_PROFILE_SECONDARY) != 0;
    on &= ~ISDF_PROFILE_SECONDARY;
    c->prof_every = on > 0 ? on : 1;
    c->prof_tick = 0;
    c->prof_used = 0;
    return ISDF_OK;
}

extern "C" int isdf_profile_read(isdf_ctx *c, int *n, double *mean_ms) {
    if (!c || !n || !mean_ms) return ISDF_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    double sum = 0.0, sum2 = 0.0;
    for (size_t i = 0; i < c->prof_used; i++) {
        HIPCHK(c, hipEventSynchronize(c->prof_secondary ? c->prof_events[i].d : c->prof_events[i].b));
        float ms = 0.f;
        HIPCHK(c, hipEventElapsedTime(&ms, c->prof_events[i].a, c->prof_events[i].b));
        sum += ms;
        if (c->prof_secondary) {
            HIPCHK(c, hipEventElapsedTime(&ms, c->prof_events[i].c, c->prof_events[i].d));
            sum2 += ms;
        }
    }
    *n = (int)c->prof_used;
    *mean_ms = c->prof_used ? sum / c->prof_used : 0.0;
    c->last_exact_ms = c->prof_used ? sum2 / c->prof_used : 0.0;
    c->prof_used = 0;
    return ISDF_OK;
}

extern "C" int isdf_profile_read_secondary(isdf_ctx *c, double *mean_ms) {
    if (!c || !mean_ms) return ISDF_ERR_INVALID_ARG;
    *mean_ms = c->last_exact_ms;
    return ISDF_OK;
}

// developer tool (not declared in the ABI header): copies the timing words of the last step when ISDF_DEBUG_TIMING=1
extern "C" long long isdf_debug_timing(isdf_ctx *c, unsigned long long *out, long long cap) {
    if (!c || !c->d_dbg) return 0;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    const long long n = (long long)c->dbg_used < cap ? (long long)c->dbg_used : cap;
    if (out && n > 0) (void)hipMemcpy(out, c->d_dbg, (size_t)n * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    return (long long)c->dbg_used;
}

extern "C" int isdf_get_stats(isdf_ctx *c, isdf_stats *out) {
    if (!c || !out) return ISDF_ERR_INVALID_ARG;
    if (!c->stats_cached) {
        HIPCHK(c, hipSetDevice(c->device));
        HIPCHK(c, hipDeviceSynchronize());
        int rc = fetch_stats(c);
        if (rc) return rc;
        rc = add_peer_stats(c);
        if (rc) return rc;
    }
    *out = c->last_stats;
    return ISDF_OK;
}

// developer tool (not declared in the ABI header): dispatch order of the tile sweep's samples (a permutation of 0..n-1)
extern "C" int isdf_debug_set_sample_map(isdf_ctx *c, const int *map, long long n) {
    if (!c) return ISDF_ERR_INVALID_ARG;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    if (c->d_sample_map) (void)hipFree(c->d_sample_map);
    c->d_sample_map = nullptr; c->sample_map_n = 0;
    if (!map || n <= 0) return ISDF_OK;
    if (hipMalloc(&c->d_sample_map, (size_t)n * sizeof(int)) != hipSuccess) return ISDF_ERR_HIP;
    (void)hipMemcpy(c->d_sample_map, map, (size_t)n * sizeof(int), hipMemcpyHostToDevice);
    c->sample_map_n = (size_t)n;
    return ISDF_OK;
}

// developer tool (not declared in the ABI header): the longest-first dispatch order the NEXT fused step would use
// (4 * workgroups entries, -1 = none); returns the number of consecutive steps the plan has been running, 0 = inactive
extern "C" int isdf_debug_plan_map(isdf_ctx *c, int *out, long long cap) {
    if (!c || c->plan_k < 4 || !c->d_plan_map) return 0;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    const long long n = 4LL * c->plan_nb < cap ? 4LL * c->plan_nb : cap;
    if (out && n > 0) (void)hipMemcpy(out, c->d_plan_map + (size_t)c->plan_cur * c->plan_ns_cap, (size_t)n * sizeof(int), hipMemcpyDeviceToHost);
    return c->plan_k;
}

// developer tool (not declared in the ABI header): per-sample (exact pairs, active pairs | culled << 31) of the last integral step
extern "C" long long isdf_debug_sample_info(isdf_ctx *c, int *out, long long cap) {
    if (!c || !c->d_sample_info || !c->have_last_P) return 0;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    const long long n = (long long)(c->last_P.piece_end - c->last_P.piece_begin) * (c->last_P.K + 1);
    const long long m = 2 * n < cap ? 2 * n : cap;
    if (out && m > 0) (void)hipMemcpy(out, c->d_sample_info, (size_t)m * sizeof(int), hipMemcpyDeviceToHost);
    return n;
}

// developer tool (not declared in the ABI header): the HIP runtime's pending error of the calling thread, without clearing it
extern "C" const char *isdf_debug_peek_hip_error(void) { return hipGetErrorString(hipPeekAtLastError()); }
