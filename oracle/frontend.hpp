// ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked into the product path.
// Front-end pose feasibility ("kernel convolution"): restates
//   BasicShape::initShape<true,*> kernel part            src/utils/include/utils/Shape.hpp:400-459
//   ByteShapeKernel::generateByteKernel / getOccupied    Shape.hpp:205-255
//   PCSmapManager::generateMapKernel                     src/map_manager/include/map_manager/PCSmap_manager.h:46-78
//   SweptVolumeManager::kernelConv<true>                 src/swept_volume/include/swept_volume/sw_manager.hpp:775-847
//   SweptVolumeManager::visit_kernels_by_distance        sw_manager.hpp:850-909
//   SweptVolumeManager::checkKernelValue                 sw_manager.hpp:911-942
//   the per-neighbour test of AstarPathSearcher::AstarGetSucc   planner_algorithm/front_end_Astar.hpp:208-222
//   AstarPathSearcher::AstarPathSearch / getHeu / getPath / the (roll, pitch) chain of getastarSE3Path
//                                                         front_end_Astar.hpp:172-403
// Byte layouts are the reference's: bit z of a row lives in byte z/8 under mask 0x80 >> (z % 8).
// Pinned against the reference's own code (oracle/_ref/libref_frontend.so, tests/test_oracle_ref.py).
#pragma once
#include "grid.hpp"
#include "shapes.hpp"
#include <cmath>
#include <map>
#include <queue>
#include <vector>

namespace orc {

struct FrontEnd {
    int k = 0, xk = 0, yk = 0;          // kernel_size, xkernel_size, ykernel_size
    double max_roll = 0, max_pitch = 0, ang_res = 1, safeh = 0, res = 1;
    int bl = 0;                          // bytes of a kernel row: (k + 7) / 8
    std::vector<uint8_t> shape_bytes;    // xk * yk kernels, each k * k * bl bytes
    std::vector<double> rot;             // xk * yk rotation matrices (row-major), Rx(roll) * Ry(pitch)
    std::vector<uint8_t> map_kernel;     // inflated map, (X + 2h) * (Y + 2h) * ((Z + 2h + 7) / 8) bytes
    int mX = 0, mY = 0, mZ = 0;

    void configure(const isdf_frontend_config &c, double resolution) {
        k = c.kernel_size; max_roll = c.kernel_max_roll; max_pitch = c.kernel_max_pitch; ang_res = c.kernel_ang_res; safeh = c.front_end_safeh;
        res = resolution;
        xk = (int)std::floor(2 * max_roll / ang_res) + 1;         // Shape.hpp:297-298
        yk = (int)std::floor(2 * max_pitch / ang_res) + 1;
        bl = (k + 7) / 8;
    }
    // Eigen::AngleAxisd(rx, UnitX) * Eigen::AngleAxisd(ry, UnitY) assigned to a Matrix3d (Shape.hpp:429): the product of two
    // AngleAxis is a quaternion product, converted with QuaternionBase::toRotationMatrix
    static void rot_xy(double rx, double ry, double *R) {
        const double aw = std::cos(0.5 * rx), ax = std::sin(0.5 * rx);       // (aw, ax, 0, 0)
        const double bw = std::cos(0.5 * ry), by = std::sin(0.5 * ry);       // (bw, 0, by, 0)
        const double w = aw * bw - ax * 0.0 - 0.0 * by - 0.0 * 0.0;
        const double x = aw * 0.0 + ax * bw + 0.0 * 0.0 - 0.0 * by;
        const double y = aw * by + 0.0 * bw + 0.0 * 0.0 - ax * 0.0;
        const double z = aw * 0.0 + 0.0 * bw + ax * by - 0.0 * 0.0;
        const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
        const double twx = tx * w, twy = ty * w, twz = tz * w;
        const double txx = tx * x, txy = ty * x, txz = tz * x;
        const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
        R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
        R[3] = txy + twz; R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
        R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1.0 - (txx + tyy);
    }
    // Shape.hpp:400-459 (the roll / pitch loops accumulate in double exactly as written)
    void build_shape_kernels(const isdf_shape &S, const MeshSdf *mesh) {
        const int per = k * k * bl;
        shape_bytes.assign((size_t)xk * yk * per, 0);
        rot.assign((size_t)xk * yk * 9, 0.0);
        const int size_side = (int)(0.5 * (k - 1));
        const double safemargin = std::max(safeh, res / 2);
        int i = 0;
        for (double roll = -max_roll; roll <= max_roll; roll += ang_res, i++) {
            int j = 0;
            for (double pitch = -max_pitch; pitch <= max_pitch; pitch += ang_res, j++) {
                if (i >= xk || j >= yk) continue;          // the reference would write past its arrays here
                double *R = rot.data() + (size_t)(i * yk + j) * 9;
                rot_xy(roll * M_PI / 180.0, pitch * M_PI / 180.0, R);
                uint8_t *m = shape_bytes.data() + (size_t)(i * yk + j) * per;
                for (int a = 0; a < k; a++)
                    for (int b = 0; b < k; b++)
                        for (int c = 0; c < k; c++) {
                            const V3 pos(res * a - size_side * res, res * b - size_side * res, res * c - size_side * res);
                            if (shape_sdf_rotated(S, mesh, pos, R) <= safemargin) m[(a * k + b) * bl + c / 8] |= (uint8_t)(0x80 >> (c % 8));
                        }
            }
        }
    }
    // PCSmap_manager.h:46-78
    void build_map_kernel(const Grid &g) {
        const int h = (k - 1) / 2;
        mX = g.X; mY = g.Y; mZ = g.Z;
        const int ys = g.Y + 2 * h, zb = (g.Z + 2 * h + 7) / 8;
        map_kernel.assign((size_t)(g.X + 2 * h) * ys * zb + 1, 0);       // +1: kernelConv reads one byte past a row's last byte
        for (int x = 0; x < g.X; x++)
            for (int y = 0; y < g.Y; y++)
                for (int z = 0; z < g.Z; z++)
                    if (g.occ[g.addr(x, y, z)] == 1) {
                        const int fz = z + h;
                        map_kernel[((size_t)(x + h) * ys + (y + h)) * zb + fz / 8] |= (uint8_t)(0x80 >> (fz % 8));
                    }
    }
    // sw_manager.hpp:813-847: true = no voxel of the kernel overlaps an occupied voxel
    bool kernel_conv(int ki, int kj, const int ind[3]) const {
        const int h = (k - 1) / 2;
        const int ys = mY + 2 * h, zb = (mZ + 2 * h + 7) / 8;
        const uint8_t *m = shape_bytes.data() + (size_t)(ki * yk + kj) * (k * k * bl);
        for (int i = 0; i < k; i++)
            for (int j = 0; j < k; j++) {
                const size_t start = ((size_t)(ind[0] + i) * ys + (ind[1] + j)) * zb + ind[2] / 8;
                const int off = ind[2] % 8;
                for (int q = 0; q < bl; q++) {
                    const uint8_t block = (uint8_t)((map_kernel[start + q] << off) | (map_kernel[start + q + 1] >> (8 - off)));
                    if (m[(i * k + j) * bl + q] & block) return false;
                }
            }
        return true;
    }
    // sw_manager.hpp:850-909
    bool visit(int &ri, int &rj, int sx, int sy, const int ind[3], int maxdepth = 800) const {
        std::vector<char> visited((size_t)xk * yk, 0);
        std::queue<std::pair<int, int>> q;
        const int zi = (xk - 1) / 2, zj = (yk - 1) / 2;
        if (kernel_conv(zi, zj, ind)) { ri = zi; rj = zj; return true; }
        q.push({sx, sy});
        visited[(size_t)sx * yk + sy] = 1;
        const int dirs[4][2] = {{0, 1}, {0, -1}, {1, 0}, {-1, 0}};
        int deep = 0;
        while (!q.empty()) {
            deep++;
            const int x = q.front().first, y = q.front().second;
            q.pop();
            if (x != zi || y != zj)
                if (kernel_conv(x, y, ind)) { ri = x; rj = y; return true; }
            for (const auto &d : dirs) {
                const int nx = x + d[0], ny = y + d[1];
                if (nx < 0 || nx >= xk || ny < 0 || ny >= yk) continue;
                if (visited[(size_t)nx * yk + ny]) continue;
                visited[(size_t)nx * yk + ny] = 1;
                q.push({nx, ny});
            }
            if (deep > maxdepth) return false;
        }
        return false;
    }
    // sw_manager.hpp:911-942
    bool check(double fr, double fp, double &cr, double &cp, const int ind[3], int *kidx = nullptr) const {
        const int fi = (int)((fr + max_roll) / ang_res), fj = (int)((fp + max_pitch) / ang_res);
        int ri = fi, rj = fj;
        if (visit(ri, rj, fi, fj, ind)) {
            cr = fr + (ri - fi) * ang_res;
            cp = fp + (rj - fj) * ang_res;
            if (kidx) *kidx = ri * yk + rj;
            return true;
        }
        if (kidx) *kidx = -1;
        return false;
    }
    // front_end_Astar.hpp:214-217: isIndexValid(vi) && !isIndexOccupiedFlate(vi, 0) && checkKernelValue(fr, fp, cr, cp, vi)
    bool successor_ok(const Grid &g, double fr, double fp, double &cr, double &cp, const int ind[3], int *kidx = nullptr) const {
        if (kidx) *kidx = -1;
        if (!g.index_valid(ind[0], ind[1], ind[2])) return false;
        if (g.occ[g.addr(ind[0], ind[1], ind[2])] == 1) return false;
        return check(fr, fp, cr, cp, ind, kidx);
    }

    // ---- the SE(3) A* itself (front_end_Astar.hpp:172-357), one search on a fresh node table (the caller's reset(), :160-170)
    struct AstarNode { double g = 0, f = 0, roll = 0, pitch = 0; int id = 0; int father = -1; };      // father: node index, -2 = the start node object
    struct AstarOut {
        bool success = false;
        std::vector<double> xyz, rp;          // path start -> goal: cube centres, (roll, pitch) in degrees
        long checks = 0, expansions = 0;      // neighbour tests (total_kernel, :218), nodes taken off the open set
    };
    // :172-188
    static double heu(const int a[3], const int b[3]) {
        const double p = 1.0 / 1000;
        const int dx = std::abs(a[0] - b[0]), dy = std::abs(a[1] - b[1]), dz = std::abs(a[2] - b[2]);
        const int dmin = std::min(dx, std::min(dy, dz)), dmax = std::max(dx, std::max(dy, dz));
        const int dmid = dx + dy + dz - dmin - dmax;
        const double h = std::sqrt(3) * dmin + std::sqrt(2) * (dmid - dmin) + (dmax - dmid);
        return h * (1 + p);
    }
    AstarOut astar(const Grid &g, const V3 &start, const V3 &goal) const {
        AstarOut out;
        if (!g.in_map(start) || !g.in_map(goal)) return out;                      // :244-249
        int si[3], gi[3];
        g.grid_index(start, si[0], si[1], si[2]);
        g.grid_index(goal, gi[0], gi[1], gi[2]);
        std::vector<AstarNode> nodes((size_t)g.X * g.Y * g.Z);                     // GridNodeMap after reset()
        // the start node is an object of its own (`new GridNode`, :260): it is what sits in the open set and what the first
        // successors get as father; the map's node of the start cell is only MARKED open (:276-278) and never leaves that state
        AstarNode start_node;
        const int START = -2;
        auto node = [&](int i) -> AstarNode & { return i == START ? start_node : nodes[(size_t)i]; };
        auto cell = [&](int i, int c[3]) {
            if (i == START) { c[0] = si[0]; c[1] = si[1]; c[2] = si[2]; return; }
            c[2] = i % g.Z; c[1] = (i / g.Z) % g.Y; c[0] = i / (g.Z * g.Y);
        };
        start_node.g = 0; start_node.f = heu(si, gi); start_node.id = 1;
        std::multimap<double, int> open;
        open.insert({start_node.f, START});
        AstarNode &smap = nodes[g.addr(si[0], si[1], si[2])];
        smap.id = 1; smap.g = 0; smap.f = start_node.f;
        int terminate = -1;
        bool found = false;
        while (!open.empty()) {
            const int cur = open.begin()->second;
            open.erase(open.begin());
            node(cur).id = -1;
            out.expansions++;
            int ci[3];
            cell(cur, ci);
            if (ci[0] == gi[0] && ci[1] == gi[1] && ci[2] == gi[2]) { terminate = cur; found = true; break; }       // :289
            const double fr = node(cur).roll, fp = node(cur).pitch, gcur = node(cur).g;
            // AstarGetSucc (:197-236) and the relaxation (:301-343) neighbour by neighbour in the loops' order
            struct Succ { int idx; double ec; };
            Succ succ[27];
            int ns = 0;
            for (int i = -1; i < 2; i++)
                for (int j = -1; j < 2; j++)
                    for (int k2 = -1; k2 < 2; k2++) {
                        const int vi[3] = {ci[0] + i, ci[1] + j, ci[2] + k2};
                        double cr = 0, cp = 0;
                        const bool valid_free = g.index_valid(vi[0], vi[1], vi[2]) && g.occ[g.addr(vi[0], vi[1], vi[2])] != 1;
                        // the reference evaluates checkKernelValue only behind && (:214-216) but counts every neighbour (:217)
                        const bool cond = valid_free && check(fr, fp, cr, cp, vi);
                        out.checks++;
                        if (cond) {
                            AstarNode &p = nodes[g.addr(vi[0], vi[1], vi[2])];
                            p.pitch = cp; p.roll = cr;                                                                   // :227-228: written whatever the node's state
                            succ[ns++] = {(int)g.addr(vi[0], vi[1], vi[2]), std::sqrt((double)(i * i + j * j + k2 * k2))};
                        }
                    }
            for (int s = 0; s < ns; s++) {
                AstarNode &nb = nodes[(size_t)succ[s].idx];
                int ni[3];
                cell(succ[s].idx, ni);
                const double tg = succ[s].ec + node(cur).g;
                (void)gcur;
                if (nb.id == 0) {                                            // :306-317
                    nb.father = cur; nb.g = tg; nb.f = tg + heu(ni, gi) + 0.0; nb.id = 1;
                    open.insert({nb.f, succ[s].idx});
                } else if (nb.id == 1) {                                     // :319-328: the key in the open set keeps the OLD fScore
                    if (tg < nb.g) { nb.father = cur; nb.g = tg; nb.f = tg + heu(ni, gi) + 0.0; }
                } else {                                                     // :330-343: a closed node is reopened
                    if (tg < nb.g) {
                        nb.father = cur; nb.g = tg; nb.f = tg + heu(ni, gi) + 0.0; nb.id = 1;
                        open.insert({nb.f, succ[s].idx});
                    }
                }
            }
        }
        if (!found) return out;
        out.success = true;
        // getPath (:359-376) / getastarSE3Path (:378-403): the father chain, reversed
        std::vector<int> chain;
        for (int p = terminate; p != -1; p = node(p).father) chain.push_back(p);
        for (size_t q = chain.size(); q-- > 0;) {
            int c[3];
            cell(chain[q], c);
            const V3 ctr = g.cube_center(c[0], c[1], c[2]);
            out.xyz.push_back(ctr.x); out.xyz.push_back(ctr.y); out.xyz.push_back(ctr.z);
            out.rp.push_back(node(chain[q]).roll); out.rp.push_back(node(chain[q]).pitch);
        }
        return out;
    }
};

} // namespace orc
