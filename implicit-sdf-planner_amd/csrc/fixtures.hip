// Host-side ingestion of the reference's own input formats (no device needed): the files a reference user already has.
//   isdf_read_pcd           ASCII .pcd as pcl::io::loadPCDFile<pcl::PointXYZ> reads it for the global map
//                           (src/map_manager/src/globalmap_gene.cpp:433-460 -> /globalmap -> PCSmapManager::rcvGlobalMapHandler,
//                           src/map_manager/src/PCSmap_manager.cpp:87-200); the points then go to isdf_set_pointcloud.
//   isdf_read_obj           Wavefront .obj as igl::read_triangle_mesh reads it (src/utils/src/Shape.cpp:36): v / f records,
//                           polygons fanned from their first vertex.
//   isdf_body_transform     Generalshape's constructor (Shape.cpp:37-49): trans, Rotate = yaw * pitch * roll from poly_params,
//                           vertices V <- (V.homogeneous() * Trans^T).hnormalized().
//   isdf_load_yaml_config   the flat yaml files of src/plan_manager/config (rosparam -> Config::loadParameters,
//                           src/utils/include/utils/config.hpp:90-202).
//   isdf_shape_from_config  SweptVolumeManager::initShape (sw_manager.hpp:255-275): registry lookup by the obj file's stem,
//                           anything else is the mesh Generalshape over the transformed obj.
#include "../../include/isdf_accel.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {
constexpr double PI_REF = 3.14159265358979323846;     // Shape.cpp:23

// Eigen::AngleAxisd(angle, unit axis a).toRotationMatrix() (Eigen/src/Geometry/AngleAxis.h), row-major out
void angle_axis_matrix(double angle, int axis, double R[9]) {
    const double ax[3] = {axis == 0 ? 1.0 : 0.0, axis == 1 ? 1.0 : 0.0, axis == 2 ? 1.0 : 0.0};
    const double s = std::sin(angle), c = std::cos(angle);
    const double sin_axis[3] = {s * ax[0], s * ax[1], s * ax[2]};
    const double cos1_axis[3] = {(1.0 - c) * ax[0], (1.0 - c) * ax[1], (1.0 - c) * ax[2]};
    double tmp;
    tmp = cos1_axis[0] * ax[1]; R[1] = tmp - sin_axis[2]; R[3] = tmp + sin_axis[2];
    tmp = cos1_axis[0] * ax[2]; R[2] = tmp + sin_axis[1]; R[6] = tmp - sin_axis[1];
    tmp = cos1_axis[1] * ax[2]; R[5] = tmp - sin_axis[0]; R[7] = tmp + sin_axis[0];
    R[0] = cos1_axis[0] * ax[0] + c; R[4] = cos1_axis[1] * ax[1] + c; R[8] = cos1_axis[2] * ax[2] + c;
}
void mat3_mul(const double A[9], const double B[9], double C[9]) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0.0;
            for (int k = 0; k < 3; k++) s += A[3 * i + k] * B[3 * k + j];
            C[3 * i + j] = s;
        }
}
std::string trim(const std::string &s) {
    size_t a = 0, b = s.size();
    while (a < b && (s[a] == ' ' || s[a] == '\t' || s[a] == '\r' || s[a] == '\n')) a++;
    while (b > a && (s[b - 1] == ' ' || s[b - 1] == '\t' || s[b - 1] == '\r' || s[b - 1] == '\n')) b--;
    return s.substr(a, b - a);
}
}  // namespace

// Rotate = yaw.toRotationMatrix() * pitch.toRotationMatrix() * roll.toRotationMatrix() (Shape.cpp:39-43), row-major
extern "C" int isdf_poly_rotation(const double poly_params[6], double rotate_out[9]) {
    if (!poly_params || !rotate_out) return ISDF_ERR_INVALID_ARG;
    double Rx[9], Ry[9], Rz[9], T[9];
    angle_axis_matrix(poly_params[3] * PI_REF / 180.0, 0, Rx);
    angle_axis_matrix(poly_params[4] * PI_REF / 180.0, 1, Ry);
    angle_axis_matrix(poly_params[5] * PI_REF / 180.0, 2, Rz);
    mat3_mul(Rz, Ry, T);
    mat3_mul(T, Rx, rotate_out);
    return ISDF_OK;
}

extern "C" long long isdf_read_pcd(const char *path, float *xyz_out, long long capacity) {
    if (!path) return ISDF_ERR_INVALID_ARG;
    FILE *f = std::fopen(path, "rb");
    if (!f) return ISDF_ERR_INVALID_ARG;
    char line[4096];
    int ix = -1, iy = -1, iz = -1, n_fields = 0;      // ix / iy / iz: FIELD positions first, token columns after the header
    std::vector<int> counts;                          // COUNT line (pcl: elements per field; absent = all 1)
    long long points = -1;
    bool ascii = false, have_data = false;
    while (std::fgets(line, sizeof(line), f)) {
        if (line[0] == '#') continue;
        char key[64] = {0};
        if (std::sscanf(line, "%63s", key) != 1) continue;
        if (!std::strcmp(key, "FIELDS")) {
            char *save = nullptr, *tok = strtok_r(line, " \t\r\n", &save);
            int k = 0;
            while ((tok = strtok_r(nullptr, " \t\r\n", &save))) {
                if (!std::strcmp(tok, "x")) ix = k; else if (!std::strcmp(tok, "y")) iy = k; else if (!std::strcmp(tok, "z")) iz = k;
                k++;
            }
            n_fields = k;
        } else if (!std::strcmp(key, "COUNT")) {
            char *save = nullptr, *tok = strtok_r(line, " \t\r\n", &save);
            while ((tok = strtok_r(nullptr, " \t\r\n", &save))) counts.push_back(std::atoi(tok));
        } else if (!std::strcmp(key, "POINTS")) {
            std::sscanf(line, "%*s %lld", &points);
        } else if (!std::strcmp(key, "DATA")) {
            char kind[64] = {0};
            std::sscanf(line, "%*s %63s", kind);
            ascii = !std::strcmp(kind, "ascii");
            have_data = true;
            break;
        }
    }
    if (!have_data || ix < 0 || iy < 0 || iz < 0) { std::fclose(f); return ISDF_ERR_INVALID_ARG; }
    if (!ascii) { std::fclose(f); return ISDF_ERR_UNSUPPORTED; }      // the shipped maps are DATA ascii
    // a field with COUNT c occupies c tokens of a data line (pcl's ascii reader walks fields x counts): x / y / z themselves
    // must be single values, the columns after a wider field shift accordingly
    if (!counts.empty()) {
        if ((int)counts.size() != n_fields) { std::fclose(f); return ISDF_ERR_INVALID_ARG; }
        std::vector<int> col(n_fields, 0);
        int c0 = 0;
        for (int k = 0; k < n_fields; k++) { if (counts[k] < 1) { std::fclose(f); return ISDF_ERR_INVALID_ARG; } col[k] = c0; c0 += counts[k]; }
        if (counts[ix] != 1 || counts[iy] != 1 || counts[iz] != 1) { std::fclose(f); return ISDF_ERR_UNSUPPORTED; }
        ix = col[ix]; iy = col[iy]; iz = col[iz];
        n_fields = c0;
    }
    long long n = 0;
    while (std::fgets(line, sizeof(line), f)) {
        float v[3] = {0, 0, 0};
        char *p = line;
        int got = 0;
        for (int k = 0; k < n_fields; k++) {
            char *end = nullptr;
            const float val = std::strtof(p, &end);         // pcl copies each token through an istream >> float: the nearest float
            if (end == p) break;
            if (k == ix) { v[0] = val; got++; } else if (k == iy) { v[1] = val; got++; } else if (k == iz) { v[2] = val; got++; }
            p = end;
        }
        if (got != 3) continue;
        if (xyz_out && n < capacity) { xyz_out[3 * n] = v[0]; xyz_out[3 * n + 1] = v[1]; xyz_out[3 * n + 2] = v[2]; }
        n++;
        if (points >= 0 && n >= points) break;
    }
    std::fclose(f);
    return n;
}

extern "C" int isdf_read_obj(const char *path, double *V_out, int capV, int32_t *F_out, int capF, int *nV_out, int *nF_out) {
    if (!path || !nV_out || !nF_out) return ISDF_ERR_INVALID_ARG;
    FILE *f = std::fopen(path, "rb");
    if (!f) return ISDF_ERR_INVALID_ARG;
    char line[8192];
    int nV = 0, nF = 0;
    std::vector<int> poly;
    while (std::fgets(line, sizeof(line), f)) {
        if (line[0] == 'v' && (line[1] == ' ' || line[1] == '\t')) {
            double x = 0, y = 0, z = 0;
            if (std::sscanf(line + 1, "%lf %lf %lf", &x, &y, &z) != 3) { std::fclose(f); return ISDF_ERR_INVALID_ARG; }
            if (V_out && nV < capV) { V_out[3 * nV] = x; V_out[3 * nV + 1] = y; V_out[3 * nV + 2] = z; }
            nV++;
        } else if (line[0] == 'f' && (line[1] == ' ' || line[1] == '\t')) {
            poly.clear();
            char *save = nullptr, *tok = strtok_r(line + 1, " \t\r\n", &save);
            while (tok) {
                const long idx = std::strtol(tok, nullptr, 10);      // "v", "v/vt", "v//vn", "v/vt/vn": the vertex index leads
                if (idx != 0) poly.push_back(idx > 0 ? (int)idx - 1 : nV + (int)idx);
                tok = strtok_r(nullptr, " \t\r\n", &save);
            }
            for (size_t j = 1; j + 1 < poly.size(); j++) {           // fan from the polygon's first vertex
                if (F_out && nF < capF) { F_out[3 * nF] = poly[0]; F_out[3 * nF + 1] = poly[j]; F_out[3 * nF + 2] = poly[j + 1]; }
                nF++;
            }
        }
    }
    std::fclose(f);
    *nV_out = nV; *nF_out = nF;
    if (F_out)
        for (int k = 0; k < 3 * (nF < capF ? nF : capF); k++) if (F_out[k] < 0 || F_out[k] >= nV) return ISDF_ERR_INVALID_ARG;
    return ISDF_OK;
}

// V <- (V.rowwise().homogeneous() * Trans^T).rowwise().hnormalized(), Trans = [Rotate | trans; 0 0 0 1] (Shape.cpp:44-49)
extern "C" int isdf_body_transform(const double poly_params[6], double *V_inout, int nV, double trans_out[3], double rotate_out[9]) {
    if (!poly_params || (nV > 0 && !V_inout)) return ISDF_ERR_INVALID_ARG;
    double R[9];
    isdf_poly_rotation(poly_params, R);
    for (int i = 0; i < nV; i++) {
        const double x = V_inout[3 * i], y = V_inout[3 * i + 1], z = V_inout[3 * i + 2];
        double o[3];
        for (int a = 0; a < 3; a++) o[a] = x * R[3 * a] + y * R[3 * a + 1] + z * R[3 * a + 2] + 1.0 * poly_params[a];
        const double w = x * 0.0 + y * 0.0 + z * 0.0 + 1.0 * 1.0;
        for (int a = 0; a < 3; a++) V_inout[3 * i + a] = o[a] / w;
    }
    if (trans_out) for (int a = 0; a < 3; a++) trans_out[a] = poly_params[a];
    if (rotate_out) std::memcpy(rotate_out, R, sizeof(R));
    return ISDF_OK;
}

extern "C" void isdf_plan_config_default(isdf_plan_config *p) {
    if (!p) return;
    std::memset(p, 0, sizeof(*p));
    isdf_config_default(&p->sweep);
    p->sweep.variant = ISDF_V1_SWEPT;             // the reference's live configuration
    // Config's in-class defaults (config.hpp:16-87) for what the yaml files may leave out
    p->threads_num = 30; p->sta_threshold = 1; p->occupancy_resolution = 1.0; p->inittime = 10.0; p->momentum = 0.0;
    p->frontend.kernel_size = p->sweep.kernel_size; p->frontend.kernel_max_roll = 45.0; p->frontend.kernel_max_pitch = 45.0;
    p->frontend.kernel_ang_res = 9.0; p->frontend.front_end_safeh = 0.0;
    p->traj_parlength = 3.0;                      // plan_manager.cpp:153
}

extern "C" int isdf_load_yaml_config(const char *path, isdf_plan_config *p) {
    if (!path || !p) return ISDF_ERR_INVALID_ARG;
    FILE *f = std::fopen(path, "rb");
    if (!f) return ISDF_ERR_INVALID_ARG;
    isdf_plan_config_default(p);
    char buf[4096];
    auto list = [](const std::string &v, double *out, int n) {
        std::string s = v;
        for (char &c : s) if (c == '[' || c == ']' || c == ',') c = ' ';
        const char *q = s.c_str();
        for (int k = 0; k < n; k++) { char *e = nullptr; const double d = std::strtod(q, &e); if (e == q) return k; out[k] = d; q = e; }
        return n;
    };
    while (std::fgets(buf, sizeof(buf), f)) {
        std::string line(buf);
        bool in_q = false;                                         // strip the comment (a '#' outside quotes)
        for (size_t i = 0; i < line.size(); i++) {
            if (line[i] == '"' || line[i] == '\'') in_q = !in_q;
            if (line[i] == '#' && !in_q) { line.resize(i); break; }
        }
        const size_t colon = line.find(':');
        if (colon == std::string::npos) continue;
        const std::string key = trim(line.substr(0, colon));
        std::string val = trim(line.substr(colon + 1));
        if (key.empty() || val.empty()) continue;
        if (val.size() >= 2 && (val.front() == '"' || val.front() == '\'')) val = val.substr(1, val.size() - 2);
        const double d = std::strtod(val.c_str(), nullptr);
        isdf_config &s = p->sweep;
        if (key == "threads_num") p->threads_num = (int)d;
        else if (key == "occupancy_resolution") p->occupancy_resolution = d;
        else if (key == "sta_threshold") p->sta_threshold = (int)d;
        else if (key == "kernel_size") { s.kernel_size = (int)d; p->frontend.kernel_size = (int)d; }
        else if (key == "kernel_max_roll") p->frontend.kernel_max_roll = d;
        else if (key == "kernel_max_pitch") p->frontend.kernel_max_pitch = d;
        else if (key == "kernel_ang_res") p->frontend.kernel_ang_res = d;
        else if (key == "front_end_safeh") p->frontend.front_end_safeh = d;
        else if (key == "integralIntervs") s.integral_intervs = (int)d;
        else if (key == "safety_hor") s.safety_hor = d;
        else if (key == "weight_p") s.weight_p = d;
        else if (key == "weight_v") s.weight_v = d;
        else if (key == "weight_omg") s.weight_omg = d;
        else if (key == "weight_theta") s.weight_theta = d;
        else if (key == "vmax") s.vmax = d;
        else if (key == "omgmax") s.omgmax = d;
        else if (key == "thetamax") s.thetamax = d;
        else if (key == "smoothingEps") s.smoothing_eps = d;
        else if (key == "vehicleMass") s.vehicle_mass = d;
        else if (key == "gravAcc") s.grav_acc = d;
        else if (key == "horizDrag") s.horiz_drag = d;
        else if (key == "vertDrag") s.vert_drag = d;
        else if (key == "parasDrag") s.paras_drag = d;
        else if (key == "speedEps") s.speed_eps = d;
        else if (key == "rho") p->rho = d;
        else if (key == "inittime") p->inittime = d;
        else if (key == "momentum") p->momentum = d;
        else if (key == "box_x") p->box[0] = d;
        else if (key == "box_y") p->box[1] = d;
        else if (key == "box_z") p->box[2] = d;
        else if (key == "poly_params") list(val, p->poly_params, 6);
        else if (key == "offsetAABBbox") list(val, p->offset_aabb, 3);
        else if (key == "mapBound") list(val, p->map_bound, 6);
        else if (key == "inputdata") std::snprintf(p->inputdata, sizeof(p->inputdata), "%s", val.c_str());
        else if (key == "pcdmapname") std::snprintf(p->pcdmapname, sizeof(p->pcdmapname), "%s", val.c_str());
    }
    std::fclose(f);
    return ISDF_OK;
}

extern "C" int isdf_shape_from_config(isdf_shape *shape, const isdf_plan_config *p, const char *package_dir, double *V_buf, int capV,
                                      int32_t *F_buf, int capF) {
    if (!shape || !p) return ISDF_ERR_INVALID_ARG;
    // stem of inputdata: "shapes/RoundedCone.obj" -> "RoundedCone" (sw_manager.hpp:247-253)
    std::string in(p->inputdata);
    const size_t slash = in.find_last_of('/');
    std::string stem = slash == std::string::npos ? in : in.substr(slash + 1);
    const size_t dot = stem.find_last_of('.');
    if (dot != std::string::npos) stem.resize(dot);
    if (isdf_shape_from_name(shape, stem.c_str()) == ISDF_OK) {
        // analytic class: the constructor keeps trans / Rotate of poly_params, every SDF maps p -> (p - trans) * Rotate first
        for (int a = 0; a < 3; a++) shape->trans[a] = p->poly_params[a];
        isdf_poly_rotation(p->poly_params, shape->rotate);
        return ISDF_OK;
    }
    if (!package_dir || !V_buf || !F_buf) return ISDF_ERR_INVALID_ARG;
    const std::string path = std::string(package_dir) + "/" + in;
    int nV = 0, nF = 0;
    int rc = isdf_read_obj(path.c_str(), V_buf, capV, F_buf, capF, &nV, &nF);
    if (rc) return rc;
    if (nV > capV || nF > capF) return ISDF_ERR_OVERFLOW;
    std::memset(shape, 0, sizeof(*shape));
    shape->kind = ISDF_SHAPE_MESH;
    shape->grad_mode = ISDF_GRAD_DEFAULT;
    rc = isdf_body_transform(p->poly_params, V_buf, nV, shape->trans, shape->rotate);
    if (rc) return rc;
    shape->mesh_vertices = V_buf; shape->mesh_faces = F_buf; shape->n_vertices = nV; shape->n_faces = nF;
    return ISDF_OK;
}
