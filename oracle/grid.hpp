// ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked into the product path.
// CPU restatement of the environment grid (occupancy + ESDF) the sweeps read.
//   addressing / centres   src/map_manager/include/map_manager/GridMap3D.h:194-195, src/map_manager/src/Gridmap3D.cpp:182-194
//   getGridIndex (quirks)  src/map_manager/src/Gridmap3D.cpp:135-175
//   isInMap / isIndexValid src/map_manager/src/Gridmap3D.cpp:40-125
//   generateESDF3d         src/map_manager/src/Gridmap3D.cpp:361-414 (positive DT only reaches grid_esdf), fillESDF :494-533
//   getSDFValueWithGrad    src/map_manager/include/map_manager/GridMap3D.h:155-193
//   getPointsInAABB        src/map_manager/include/map_manager/PCSmap_manager.h:130-170
// Storage here is float for the ESDF (what the device kernel stages; SURVEY.md §7 "Precision") and uint8 for
// occupancy; the reference stores double for both.  Values are exactly representable either way for occupancy,
// and the ESDF is only compared against occ_thresh and used for the conservative cull.
#pragma once
#include "vecmath.hpp"
#include <vector>
#include <cstdint>
#include <limits>

namespace orc {

struct Grid {
    int X = 0, Y = 0, Z = 0;
    double res = 1.0;
    V3 bmin, bmax;                 // boundary_xyzmin / max; bmax = bmin + size*res for grids set through the ABI
    std::vector<uint8_t> occ;      // 1 = occupied
    std::vector<float> esdf;       // metres, unsigned (0 inside obstacles)
    bool has_occ = false, has_esdf = false;

    size_t addr(int ix, int iy, int iz) const { return (size_t)ix * Y * Z + (size_t)iy * Z + iz; }
    bool index_valid(int ix, int iy, int iz) const {
        return ix >= 0 && ix < X && iy >= 0 && iy < Y && iz >= 0 && iz < Z;
    }
    bool in_map(const V3 &p) const {
        if (p.x < bmin.x || p.y < bmin.y || p.z < bmin.z) return false;
        if (p.x > bmax.x || p.y > bmax.y || p.z > bmax.z) return false;
        return true;
    }
    // Gridmap3D.cpp:135-175, including the "iy<0 -> ix=0" / "iz<0 -> ix=0" typos
    void grid_index(const V3 &p, int &ix, int &iy, int &iz) const {
        if (!in_map(p)) { ix = iy = iz = 0; return; }
        const double dx = p.x - bmin.x, dy = p.y - bmin.y, dz = p.z - bmin.z;
        ix = (int)std::floor(dx / res);
        iy = (int)std::floor(dy / res);
        iz = (int)std::floor(dz / res);
        if (ix < 0) ix = 0;
        if (ix >= X) ix = X - 1;
        if (iy < 0) ix = 0;
        if (iy >= Y) iy = Y - 1;
        if (iz < 0) ix = 0;
        if (iz >= Z) iz = Z - 1;
    }
    // Gridmap3D.cpp:182-194 (invalid index -> (0,0,0))
    V3 cube_center(int ix, int iy, int iz) const {
        if (!index_valid(ix, iy, iz)) return V3(0, 0, 0);
        const V3 d((ix + 0.5) * res, (iy + 0.5) * res, (iz + 0.5) * res);
        return d + bmin;
    }
    // isIndexOccupied(int,int,int): out-of-range counts as occupied (Gridmap3D.cpp:237-)
    bool index_occupied(int ix, int iy, int iz) const {
        if (!index_valid(ix, iy, iz)) return true;
        return occ[addr(ix, iy, iz)] != 0;
    }
    // V3 qualification: esdf <= thresh, compared in double like the reference's double grid_esdf would be
    // (GridMap3D.h:155-193 reads doubles); the stored fp32 value converts exactly
    bool index_qualifies_esdf(int ix, int iy, int iz, double thresh) const {
        if (!index_valid(ix, iy, iz)) return true;
        return (double)esdf[addr(ix, iy, iz)] <= thresh;
    }
    // getGridSDFValue(Vector3i): invalid -> 0 (Gridmap3D.cpp:535-542)
    double esdf_at(int ix, int iy, int iz) const {
        if (!index_valid(ix, iy, iz)) return 0;
        return (double)esdf[addr(ix, iy, iz)];
    }

    // PCSmap_manager.h:130-137
    void proj_in_map(V3 &p) const {
        if (p.x < bmin.x) p.x = bmin.x;
        if (p.y < bmin.y) p.y = bmin.y;
        if (p.z < bmin.z) p.z = bmin.z;
        if (p.x > bmax.x) p.x = bmax.x;
        if (p.y > bmax.y) p.y = bmax.y;
        if (p.z > bmax.z) p.z = bmax.z;
    }
    // index box of getPointsInAABB(center, half...) — PCSmap_manager.h:150-155
    void aabb_index_box(const V3 &center, double half, int lo[3], int hi[3]) const {
        V3 c1 = center - V3(half, half, half), c2 = center + V3(half, half, half);
        proj_in_map(c1);
        proj_in_map(c2);
        grid_index(c1, lo[0], lo[1], lo[2]);
        grid_index(c2, hi[0], hi[1], hi[2]);
    }

    // GridMap3D.h:155-193.  The isInMap test result is overwritten (:170-173) so it is omitted.
    double esdf_trilinear(const V3 &pos, V3 &grad) const {
        const V3 pm = pos - V3(0.5 * res, 0.5 * res, 0.5 * res);
        int ix, iy, iz;
        grid_index(pm, ix, iy, iz);
        const V3 ipos = cube_center(ix, iy, iz);
        const V3 diff = (pos - ipos) * (1.0 / res);
        double v[2][2][2];
        for (int x = 0; x < 2; x++)
            for (int y = 0; y < 2; y++)
                for (int z = 0; z < 2; z++) v[x][y][z] = esdf_at(ix + x, iy + y, iz + z);
        const double v00 = (1 - diff.x) * v[0][0][0] + diff.x * v[1][0][0];
        const double v01 = (1 - diff.x) * v[0][0][1] + diff.x * v[1][0][1];
        const double v10 = (1 - diff.x) * v[0][1][0] + diff.x * v[1][1][0];
        const double v11 = (1 - diff.x) * v[0][1][1] + diff.x * v[1][1][1];
        const double v0 = (1 - diff.y) * v00 + diff.y * v10;
        const double v1 = (1 - diff.y) * v01 + diff.y * v11;
        const double dist = (1 - diff.z) * v0 + diff.z * v1;
        grad.z = (v1 - v0) * (1.0 / res);
        grad.y = ((1 - diff.z) * (v10 - v00) + diff.z * (v11 - v01)) * (1.0 / res);
        grad.x = (1 - diff.z) * (1 - diff.y) * (v[1][0][0] - v[0][0][0]);
        grad.x += (1 - diff.z) * diff.y * (v[1][1][0] - v[0][1][0]);
        grad.x += diff.z * (1 - diff.y) * (v[1][0][1] - v[0][0][1]);
        grad.x += diff.z * diff.y * (v[1][1][1] - v[0][1][1]);
        grad.x *= (1.0 / res);
        return dist;
    }
};

// Felzenszwalb 1-D squared distance transform exactly as GridMap3D::fillESDF (Gridmap3D.cpp:494-533),
// including the DBL_MAX sentinel arithmetic.
template <typename FGet, typename FSet>
inline void fill_esdf_1d(FGet f_get, FSet f_set, int start, int end, int n) {
    std::vector<int> v(n);
    std::vector<double> z(n + 1);
    int k = start;
    v[start] = start;
    z[start] = -std::numeric_limits<double>::max();
    z[start + 1] = std::numeric_limits<double>::max();
    for (int q = start + 1; q <= end; q++) {
        k++;
        double s;
        do {
            k--;
            s = ((f_get(q) + q * q) - (f_get(v[k]) + v[k] * v[k])) / (2 * q - 2 * v[k]);
        } while (s <= z[k]);
        k++;
        v[k] = q;
        z[k] = s;
        z[k + 1] = std::numeric_limits<double>::max();
    }
    k = start;
    for (int q = start; q <= end; q++) {
        while (z[k + 1] < q) k++;
        const double val = (q - v[k]) * (q - v[k]) + f_get(v[k]);
        f_set(q, val);
    }
}

// generateESDF3d, positive DT: z pass, y pass, x pass; grid_esdf = res * sqrt(d^2) (double), Gridmap3D.cpp:373-414.
inline void build_esdf(const Grid &g, std::vector<double> &out) {
    const size_t total = (size_t)g.X * g.Y * g.Z;
    std::vector<double> b1(total), b2(total);
    out.assign(total, 0.0);
    const double BIG = std::numeric_limits<double>::max();
    for (int x = 0; x < g.X; x++)
        for (int y = 0; y < g.Y; y++)
            fill_esdf_1d([&](int z) { return g.index_occupied(x, y, z) ? 0.0 : BIG; },
                         [&](int z, double val) { b1[g.addr(x, y, z)] = val; }, 0, g.Z - 1, g.Z);
    for (int x = 0; x < g.X; x++)
        for (int z = 0; z < g.Z; z++)
            fill_esdf_1d([&](int y) { return b1[g.addr(x, y, z)]; },
                         [&](int y, double val) { b2[g.addr(x, y, z)] = val; }, 0, g.Y - 1, g.Y);
    for (int y = 0; y < g.Y; y++)
        for (int z = 0; z < g.Z; z++)
            fill_esdf_1d([&](int x) { return b2[g.addr(x, y, z)]; },
                         [&](int x, double val) { out[g.addr(x, y, z)] = g.res * std::sqrt(val); }, 0, g.X - 1, g.X);
}

} // namespace orc
