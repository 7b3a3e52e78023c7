// Fast winding number hierarchy for the MESH robot shape, built on the host at isdf_set_shape and walked on the device.
//
// ATTRIBUTION.  The host-side BUILD in this file (the 4-wide BVH construction and the per-node order-2 expansion
// coefficients) follows, function by function, one specific third-party source:
//     "Fast Winding Numbers for Soups" (WindingNumber / HDK sample UT_SolidAngle, UT_BVH), https://github.com/alecjacobson/WindingNumber
//     Copyright (c) 2018 Side Effects Software Inc., MIT License
//     as vendored by libigl 2.4.0 in igl/FastWindingNumberForSoups.h (the reference ships it at
//     src/utils/include/igl/FastWindingNumberForSoups.h):
//       BVH<N>::init / initNode / multiSplit / split / nthElement / partitionByCentre      :4555-5921
//       UT_SolidAngle::init: PrecomputeFunctors::item / post                                :6514-7097
//       UT_SolidAngle::computeSolidAngle, UTsignedSolidAngleTri (host evaluation, tests)    :6044-6082, :7119-7258
// It is a de-templated, compacted transliteration of those functions (same control flow, largely the same local names), not
// an independent design, and it is here for one reason: the reference's mesh SDF takes its sign from
// igl::fast_winding_number(fwn_bvh, 2.0, p, w) (Shape.cpp:86,110), whose far-field terms are an APPROXIMATION (errors up to
// ~2e-3 in w, entering the SDF through s = 1 - 2w) - so parity needs the SAME tree (same splits, same child order) and the
// SAME coefficients, which only the same construction yields.  The product may not link the vendored header.  The upstream
// licence text:
//
//   MIT License.  Copyright (c) 2018 Side Effects Software Inc.
//   Permission is hereby granted, free of charge, to any person obtaining a copy of this software and associated documentation
//   files (the "Software"), to deal in the Software without restriction, including without limitation the rights to use, copy,
//   modify, merge, publish, distribute, sublicense, and/or sell copies of the Software, and to permit persons to whom the
//   Software is furnished to do so, subject to the following conditions:  The above copyright notice and this permission
//   notice shall be included in all copies or substantial portions of the Software.
//   THE SOFTWARE IS PROVIDED "AS IS", WITHOUT WARRANTY OF ANY KIND, EXPRESS OR IMPLIED, INCLUDING BUT NOT LIMITED TO THE
//   WARRANTIES OF MERCHANTABILITY, FITNESS FOR A PARTICULAR PURPOSE AND NONINFRINGEMENT.  IN NO EVENT SHALL THE AUTHORS OR
//   COPYRIGHT HOLDERS BE LIABLE FOR ANY CLAIM, DAMAGES OR OTHER LIABILITY, WHETHER IN AN ACTION OF CONTRACT, TORT OR OTHERWISE,
//   ARISING FROM, OUT OF OR IN CONNECTION WITH THE SOFTWARE OR THE USE OR OTHER DEALINGS IN THE SOFTWARE.
//
// What is this repository's own: the flat node record (92 floats per node) the device walks, and the device traversal itself
// (csrc/dev_mesh.hpp: explicit stack, one combined winding-number / closest-point question per voxel).
// All arithmetic is float, like UT_SolidAngle<float,float>.  tests/test_fwn_host.py compares tree, coefficients and w with the
// vendored library itself, node by node (oracle/_ref/libref_fwn.so).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

#ifdef __clang__
#pragma clang fp contract(off)     // the split heuristics compare float expressions: keep them as written
#endif

namespace isdf_host {

constexpr int FWN_BOX_FLOATS = 92;            // per node: 23 quantities x 4 children
constexpr int32_t FWN_EMPTY = -1;
constexpr uint32_t FWN_INTERNAL_BIT = 0x80000000u;
// offsets (in floats) of the quantities inside a node's record, each 4 wide (one per child)
enum { FWN_MAXP2 = 0, FWN_P = 4, FWN_N = 16, FWN_NII = 28, FWN_NXY = 40, FWN_NYZ = 44, FWN_NZX = 48, FWN_NIII = 52, FWN_SUMPERM = 64,
       FWN_2NXXY = 68, FWN_2NXXZ = 72, FWN_2NYYZ = 76, FWN_2NYYX = 80, FWN_2NZZX = 84, FWN_2NZZY = 88 };

struct FwnTree {
    std::vector<int32_t> child;        // 4 per node: triangle index | FWN_EMPTY | node number with the top bit set
    std::vector<float> box;            // FWN_BOX_FLOATS per node
    std::vector<float> pos;            // 3 per vertex (float copies)
    std::vector<int32_t> tri;          // 3 per triangle
    int n_nodes() const { return (int)(child.size() / 4); }
};

namespace fwn_detail {

struct Box3 {
    float v[3][2];
    void init_empty() { for (int a = 0; a < 3; a++) { v[a][0] = std::numeric_limits<float>::max(); v[a][1] = -std::numeric_limits<float>::max(); } }
    void combine(const Box3 &s) {
        for (int a = 0; a < 3; a++) {
            v[a][0] = (v[a][0] < s.v[a][0]) ? v[a][0] : s.v[a][0];
            v[a][1] = (v[a][1] > s.v[a][1]) ? v[a][1] : s.v[a][1];
        }
    }
    float half_area() const {
        const float d0 = v[0][1] - v[0][0], d1 = v[1][1] - v[1][0], d2 = v[2][1] - v[2][0];
        return d0 * d1 + d1 * d2 + d2 * d0;
    }
    float centre2(int axis) const { return v[axis][0] + v[axis][1]; }      // utBoxCenter: twice the centre
};

constexpr int NSPANS = 16, NSPLITS = 15, MIN_FRACTION = 16, SMALL_LIMIT = 6, MID_LIMIT = 32;

struct Builder {
    const Box3 *boxes;
    std::vector<int32_t> nodes;      // 4 per node

    // partitionByCentre (:5876-5921)
    void partition_by_centre(int *indices, const int *indices_end, int axis, float pivotx2, int *&ppivot_start, int *&ppivot_end) {
        int *pivot_start = indices, *pivot_end = indices;
        for (int *psrc = indices; psrc != indices_end; ++psrc) {
            const float srcsum = boxes[*psrc].centre2(axis);
            if (srcsum < pivotx2) {
                if (psrc != pivot_start) {
                    if (pivot_start == pivot_end) { const int t = *psrc; *psrc = *pivot_start; *pivot_start = t; }
                    else { const int t = *psrc; *psrc = *pivot_end; *pivot_end = *pivot_start; *pivot_start = t; }
                }
                ++pivot_start; ++pivot_end;
            } else if (srcsum == pivotx2) {
                if (psrc != pivot_end) { const int t = *psrc; *psrc = *pivot_end; *pivot_end = t; }
                ++pivot_end;
            }
        }
        ppivot_start = pivot_start; ppivot_end = pivot_end;
    }
    // nthElement (:5821-5874)
    void nth_element(int *indices, const int *indices_end, int axis, int *const nth) {
        for (;;) {
            float pv[3] = {boxes[indices[0]].centre2(axis), boxes[indices[(indices_end - indices) / 2]].centre2(axis), boxes[*(indices_end - 1)].centre2(axis)};
            if (pv[0] < pv[1]) std::swap(pv[0], pv[1]);
            if (pv[0] < pv[2]) std::swap(pv[0], pv[2]);
            if (pv[1] < pv[2]) std::swap(pv[1], pv[2]);
            int *ps, *pe;
            partition_by_centre(indices, indices_end, axis, pv[1], ps, pe);
            if (nth < ps) indices_end = ps;
            else if (nth < pe) return;
            else indices = pe;
            if (indices_end <= indices + 1) return;
        }
    }
    // split (:5326-5775): two parts [indices, split) and [split, indices + n), with their boxes
    void split(const Box3 &axes, int *indices, int nboxes, int *&split_indices, Box3 *split_boxes) {
        if (nboxes == 2) { split_boxes[0] = boxes[indices[0]]; split_boxes[1] = boxes[indices[1]]; split_indices = indices + 1; return; }
        if (nboxes <= SMALL_LIMIT) {
            // every partition of the boxes with box 0 on side 0
            Box3 local[SMALL_LIMIT];
            for (int b = 0; b < nboxes; b++) local[b] = boxes[indices[b]];
            const int limit = 1 << (nboxes - 1);
            int best = -1;
            float best_h = 0.f;
            for (int bits = 1; bits < limit; ++bits) {
                Box3 sub[2];
                sub[0] = local[0];
                sub[1].init_empty();
                int cnt[2] = {1, 0};
                for (int bit = 0; bit < nboxes - 1; ++bit) { const int dst = (bits >> bit) & 1; sub[dst].combine(local[bit + 1]); ++cnt[dst]; }
                const float h = sub[0].half_area() * cnt[0] + sub[1].half_area() * cnt[1];
                if (best == -1 || h < best_h) { best = bits; best_h = h; split_boxes[0] = sub[0]; split_boxes[1] = sub[1]; }
            }
            int local_idx[SMALL_LIMIT - 1];
            for (int b = 0; b < nboxes - 1; b++) local_idx[b] = indices[b + 1];
            int *dst = indices + 1;
            for (int bit = 0; bit < nboxes - 1; ++bit) if (!((best >> bit) & 1)) *dst++ = local_idx[bit];
            split_indices = dst;
            for (int bit = 0; bit < nboxes - 1; ++bit) if ((best >> bit) & 1) *dst++ = local_idx[bit];
            return;
        }
        int max_axis = 0;
        float max_len = axes.v[0][1] - axes.v[0][0];
        for (int a = 1; a < 3; a++) { const float l = axes.v[a][1] - axes.v[a][0]; if (l > max_len) { max_axis = a; max_len = l; } }
        if (!(max_len > 0.f)) { split_indices = indices + nboxes / 2; split_boxes[0] = axes; split_boxes[1] = axes; return; }
        const int axis = max_axis;
        if (nboxes <= MID_LIMIT) {
            // sort along the axis (4 insertion-sorted chunks, merged), then try every split position
            float mid[MID_LIMIT];
            int li[MID_LIMIT], tmp[MID_LIMIT];
            for (int i = 0; i < nboxes; i++) { mid[i] = boxes[indices[i]].centre2(axis); li[i] = i; }
            const int cs[5] = {0, nboxes / 4, nboxes / 2, (int)((3 * (uint64_t)nboxes) / 4), nboxes};
            for (int chunk = 0; chunk < 4; ++chunk) {
                const int start = cs[chunk], end = cs[chunk + 1];
                for (int i = start + 1; i < end; ++i) {
                    int indexi = li[i];
                    const float vi = mid[indexi];
                    for (int j = start; j < i; ++j) {
                        int indexj = li[j];
                        const float vj = mid[indexj];
                        if (vi < vj) {
                            do {
                                li[j] = indexi; indexi = indexj; ++j;
                                if (j == i) { li[j] = indexi; break; }
                                indexj = li[j];
                            } while (true);
                            break;
                        }
                    }
                }
            }
            auto less = [&mid](int a, int b) { return mid[a] < mid[b]; };
            std::merge(li, li + cs[1], li + cs[1], li + cs[2], tmp, less);
            std::merge(li + cs[2], li + cs[3], li + cs[3], li + cs[4], tmp + cs[2], less);
            std::merge(tmp, tmp + cs[2], tmp + cs[2], tmp + cs[4], li, less);
            for (int i = 0; i < nboxes; i++) li[i] = indices[li[i]];
            for (int i = 0; i < nboxes; i++) indices[i] = li[i];
            Box3 left[MID_LIMIT - 1], right[MID_LIMIT - 1];
            const int nsplits = nboxes - 1;
            Box3 acc = boxes[li[0]];
            left[0] = acc;
            for (int i = 1; i < nsplits; ++i) { acc.combine(boxes[li[i]]); left[i] = acc; }
            acc = boxes[li[nsplits - 1]];              // sic (:5533): the last box never enters the right-hand boxes
            right[nsplits - 1] = acc;
            for (int i = nsplits - 1; i > 0; --i) { acc.combine(boxes[li[i]]); right[i - 1] = acc; }
            int best_split = 0;
            float best_h = left[0].half_area() + right[0].half_area() * (nboxes - 1);
            for (int s = 1; s < nsplits; ++s) {
                const float h = left[s].half_area() * (s + 1) + right[s].half_area() * (nboxes - (s + 1));
                if (h < best_h) { best_split = s; best_h = h; }
            }
            split_indices = indices + best_split + 1;
            split_boxes[0] = left[best_split];
            split_boxes[1] = right[best_split];
            return;
        }
        // 16 equal spans along the axis
        const float axis_min = axes.v[max_axis][0], axis_length = max_len;
        Box3 span_boxes[NSPANS];
        int span_counts[NSPANS];
        for (int i = 0; i < NSPANS; i++) { span_boxes[i].init_empty(); span_counts[i] = 0; }
        const float axis_min_x2 = 2 * axis_min;
        const float axis_index_scale = (float(1.0 / 2) * NSPANS) / axis_length;
        for (int k = 0; k < nboxes; ++k) {
            const Box3 &box = boxes[indices[k]];
            const float sum = box.centre2(axis);
            int si = int((sum - axis_min_x2) * axis_index_scale);
            si = si < 0 ? 0 : (si > NSPANS - 1 ? NSPANS - 1 : si);
            ++span_counts[si];
            span_boxes[si].combine(box);
        }
        Box3 left_boxes[NSPLITS], right_boxes[NSPLITS];
        Box3 acc = span_boxes[0];
        left_boxes[0] = acc;
        for (int i = 1; i < NSPLITS; ++i) { acc.combine(span_boxes[i]); left_boxes[i] = acc; }
        acc = span_boxes[NSPANS - 1];
        right_boxes[NSPLITS - 1] = acc;
        for (int i = NSPLITS - 1; i > 0; --i) { acc.combine(span_boxes[i]); right_boxes[i - 1] = acc; }
        int left_counts[NSPLITS];
        int cacc = span_counts[0];
        left_counts[0] = cacc;
        for (int s = 1; s < NSPLITS; ++s) { cacc += span_counts[s]; left_counts[s] = cacc; }
        const int min_count = nboxes / MIN_FRACTION;
        const int max_count = (int)(((MIN_FRACTION - 1) * (uint64_t)nboxes) / MIN_FRACTION);
        float smallest = std::numeric_limits<float>::infinity();
        int split_index = -1;
        for (int s = 0; s < NSPLITS; ++s) {
            const int lc = left_counts[s];
            if (lc < min_count || lc > max_count) continue;
            const int rc = nboxes - lc;
            const float h = lc * left_boxes[s].half_area() + rc * right_boxes[s].half_area();
            if (h < smallest) { smallest = h; split_index = s; }
        }
        int *const indices_end = indices + nboxes;
        if (split_index == -1) {
            int *nth;
            if (left_counts[0] > max_count) nth = indices + max_count;
            else if (left_counts[NSPLITS - 1] < min_count) nth = indices + min_count;
            else nth = indices + nboxes / 2;
            nth_element(indices, indices + nboxes, max_axis, nth);
            split_indices = nth;
            Box3 lb = boxes[indices[0]];
            for (int *p = indices + 1; p < nth; ++p) lb.combine(boxes[*p]);
            Box3 rb = boxes[nth[0]];
            for (int *p = nth + 1; p < indices_end; ++p) rb.combine(boxes[*p]);
            split_boxes[0] = lb; split_boxes[1] = rb;
        } else {
            const float pivotx2 = axis_min_x2 + (split_index + 1) * axis_length / (NSPANS / 2);
            int *ps, *pe;
            partition_by_centre(indices, indices + nboxes, max_axis, pivotx2, ps, pe);
            split_indices = indices + left_counts[split_index];
            if (split_indices >= ps && split_indices <= pe) { split_boxes[0] = left_boxes[split_index]; split_boxes[1] = right_boxes[split_index]; return; }
            if (split_indices < ps) split_indices = ps; else split_indices = pe;
            if (split_indices == indices) ++split_indices;
            else if (split_indices == indices_end) --split_indices;
            Box3 lb = boxes[indices[0]];
            for (int *p = indices + 1; p < split_indices; ++p) lb.combine(boxes[*p]);
            Box3 rb = boxes[split_indices[0]];
            for (int *p = split_indices + 1; p < indices_end; ++p) rb.combine(boxes[*p]);
            split_boxes[0] = lb; split_boxes[1] = rb;
        }
    }
    // multiSplit, BOX_AREA branch (:5225-5322): four parts
    void multi_split(const Box3 &axes, int *indices, int nboxes, int *sub_indices[5], Box3 sub_boxes[4]) {
        sub_indices[0] = indices;
        sub_indices[2] = indices + nboxes;
        split(axes, indices, nboxes, sub_indices[1], &sub_boxes[0]);
        float areas[4];
        areas[0] = sub_boxes[0].half_area();
        areas[1] = sub_boxes[1].half_area();
        for (int nsub = 2; nsub < 4; ++nsub) {
            int choice = -1;
            float max_h = 0.f;
            for (int i = 0; i < nsub; ++i) {
                const int cnt = (int)(sub_indices[i + 1] - sub_indices[i]);
                if (cnt > 1) {
                    const float h = areas[i] * cnt;
                    if (choice == -1 || h > max_h) { choice = i; max_h = h; }
                }
            }
            int *sel_start = sub_indices[choice], *sel_end = sub_indices[choice + 1];
            for (int i = nsub; i > choice; --i) sub_indices[i + 1] = sub_indices[i];
            for (int i = nsub - 1; i > choice; --i) sub_boxes[i + 1] = sub_boxes[i];
            for (int i = nsub - 1; i > choice; --i) areas[i + 1] = areas[i];
            const Box3 parent = sub_boxes[choice];
            split(parent, sel_start, (int)(sel_end - sel_start), sub_indices[choice + 1], &sub_boxes[choice]);
            areas[choice] = sub_boxes[choice].half_area();
            areas[choice + 1] = sub_boxes[choice + 1].half_area();
        }
    }
    // initNode (:4894-5023); child nodes are appended in depth-first order
    void init_node(int nodei, const Box3 &axes, int *indices, int nboxes) {
        if (nboxes <= 4) {
            for (int i = 0; i < nboxes; i++) nodes[4 * nodei + i] = indices[i];
            for (int i = nboxes; i < 4; i++) nodes[4 * nodei + i] = FWN_EMPTY;
            return;
        }
        int *sub_indices[5];
        Box3 sub_boxes[4];
        multi_split(axes, indices, nboxes, sub_indices, sub_boxes);
        for (int i = 0; i < 4; i++) if (sub_indices[i + 1] - sub_indices[i] == 1) nodes[4 * nodei + i] = sub_indices[i][0];
        for (int i = 0; i < 4; i++) {
            const int cnt = (int)(sub_indices[i + 1] - sub_indices[i]);
            if (cnt != 1) {
                const int child = (int)(nodes.size() / 4);
                nodes[4 * nodei + i] = (int32_t)((uint32_t)child | FWN_INTERNAL_BIT);
                nodes.resize(nodes.size() + 4, FWN_EMPTY);
                init_node(child, sub_boxes[i], sub_indices[i], cnt);
            }
        }
    }
};

struct V3f { float x, y, z; };
inline V3f operator-(V3f a, V3f b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3f operator+(V3f a, V3f b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3f operator*(V3f a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3f operator*(float s, V3f a) { return {s * a.x, s * a.y, s * a.z}; }
inline V3f operator*(V3f a, V3f b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3f operator/(V3f a, float s) { s = 1 / s; return {a.x * s, a.y * s, a.z * s}; }      // UT_FixedVector::operator/(T): the reciprocal, then three products (:3419-3434)
inline V3f cross(V3f a, V3f b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float comp(const V3f &a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
inline float &comp(V3f &a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
inline float length2(V3f a) { return a.x * a.x + a.y * a.y + a.z * a.z; }

struct LocalData {           // UT_SolidAngle::init, struct LocalData (:6580-6611)
    Box3 box;
    V3f averageP, areaP, N;
    float area;
    V3f NijDiag;
    float Nxy, Nyx, Nyz, Nzy, Nzx, Nxz;
    V3f NijkDiag;
    float sumPermuteNxyz, n2Nxxy_Nyxx, n2Nxxz_Nzxx, n2Nyyz_Nzyy, n2Nyyx_Nxyy, n2Nzzx_Nxzz, n2Nzzy_Nyzz;
};

// compute_integrals lambda (:6725-6810): a, b, c sorted along axis i
inline void compute_integrals(const V3f &a, const V3f &b, const V3f &c, const V3f &P, float *integral_ii, float *integral_ij, float *integral_ik, const int i) {
    const V3f oab = b - a, oac = c - a, ocb = b - c;
    const float t = comp(oab, i) / comp(oac, i);
    const int j = (i == 2) ? 0 : (i + 1);
    const int k = (j == 2) ? 0 : (j + 1);
    const float jdiff = t * comp(oac, j) - comp(oab, j);
    const float kdiff = t * comp(oac, k) - comp(oab, k);
    V3f cross_a, cross_c;
    cross_a.x = (jdiff * comp(oab, k) - kdiff * comp(oab, j)); cross_a.y = kdiff * comp(oab, i); cross_a.z = jdiff * comp(oab, i);
    cross_c.x = (jdiff * comp(ocb, k) - kdiff * comp(ocb, j)); cross_c.y = kdiff * comp(ocb, i); cross_c.z = jdiff * comp(ocb, i);
    const float area_scale_a = std::sqrt(length2(cross_a)), area_scale_c = std::sqrt(length2(cross_c));
    const float Pai = comp(a, i) - comp(P, i), Pci = comp(c, i) - comp(P, i);
    const float oabi = comp(oab, i), ocbi = comp(ocb, i);
    const float int_ii_a = area_scale_a * (0.5f * Pai * Pai + float(2.0 / 3.0) * Pai * oabi + 0.25f * oabi * oabi);
    const float int_ii_c = area_scale_c * (0.5f * Pci * Pci + float(2.0 / 3.0) * Pci * ocbi + 0.25f * ocbi * ocbi);
    *integral_ii = int_ii_a + int_ii_c;
    int jk = j;
    float *integral = integral_ij;
    float diff = jdiff;
    for (;;) {
        if (integral) {
            const float obmidj = comp(b, jk) + 0.5f * diff;
            const float oabmidj = obmidj - comp(a, jk), ocbmidj = obmidj - comp(c, jk);
            const float Paj = comp(a, jk) - comp(P, jk), Pcj = comp(c, jk) - comp(P, jk);
            const float int_ij_a = area_scale_a * (0.5f * Pai * Paj + float(1.0 / 3.0) * Pai * oabmidj + float(1.0 / 3.0) * Paj * oabi + 0.25f * oabi * oabmidj);
            const float int_ij_c = area_scale_c * (0.5f * Pci * Pcj + float(1.0 / 3.0) * Pci * ocbmidj + float(1.0 / 3.0) * Pcj * ocbi + 0.25f * ocbi * ocbmidj);
            *integral = int_ij_a + int_ij_c;
        }
        if (jk == k) break;
        jk = k;
        integral = integral_ik;
        diff = kdiff;
    }
}

struct Precompute {
    FwnTree &tr;
    const Box3 *tri_boxes;
    V3f vert(int i) const { return {tr.pos[3 * i], tr.pos[3 * i + 1], tr.pos[3 * i + 2]}; }

    // PrecomputeFunctors::item (:6643-6870)
    void item(int itemi, LocalData &d) const {
        const V3f a = vert(tr.tri[3 * itemi]), b = vert(tr.tri[3 * itemi + 1]), c = vert(tr.tri[3 * itemi + 2]);
        const V3f ab = b - a, ac = c - a;
        d.box = tri_boxes[itemi];
        const V3f N = 0.5f * cross(ab, ac);
        const float area2 = length2(N), area = std::sqrt(area2);
        const V3f P = (a + b + c) / 3.0f;
        d.averageP = P; d.areaP = P * area; d.N = N; d.area = area;
        d.NijDiag = {0, 0, 0};
        d.Nxy = d.Nyx = d.Nyz = d.Nzy = d.Nzx = d.Nxz = 0;
        d.NijkDiag = {0, 0, 0};
        d.sumPermuteNxyz = d.n2Nxxy_Nyxx = d.n2Nxxz_Nzxx = d.n2Nyyz_Nzyy = d.n2Nyyx_Nxyy = d.n2Nzzx_Nxzz = d.n2Nzzy_Nyzz = 0;
        if (area == 0) return;
        const V3f n = N / area;
        const V3f values[3] = {a, b, c};
        float span[3];
        int order[3][3];
        for (int ax = 0; ax < 3; ax++) {
            int *o = order[ax];
            o[0] = 0; o[1] = 1; o[2] = 2;
            if (comp(a, ax) > comp(b, ax)) std::swap(o[0], o[1]);
            if (comp(values[o[0]], ax) > comp(c, ax)) std::swap(o[0], o[2]);
            if (comp(values[o[1]], ax) > comp(values[o[2]], ax)) std::swap(o[1], o[2]);
            span[ax] = comp(values[o[2]], ax) - comp(values[o[0]], ax);
        }
        const float dx = span[0], dy = span[1], dz = span[2];
        float ixx = 0, ixy = 0, iyy = 0, iyz = 0, izz = 0, izx = 0;
        if (dx > 0) compute_integrals(values[order[0][0]], values[order[0][1]], values[order[0][2]], P, &ixx, ((dx >= dy && dy > 0) ? &ixy : nullptr), ((dx >= dz && dz > 0) ? &izx : nullptr), 0);
        if (dy > 0) compute_integrals(values[order[1][0]], values[order[1][1]], values[order[1][2]], P, &iyy, ((dy >= dz && dz > 0) ? &iyz : nullptr), ((dx < dy && dx > 0) ? &ixy : nullptr), 1);
        if (dz > 0) compute_integrals(values[order[2][0]], values[order[2][1]], values[order[2][2]], P, &izz, ((dx < dz && dx > 0) ? &izx : nullptr), ((dy < dz && dy > 0) ? &iyz : nullptr), 2);
        d.NijkDiag = V3f{ixx, iyy, izz} * n;
        d.sumPermuteNxyz = 2 * (n.x * iyz + n.y * izx + n.z * ixy);
        const float Nxxy = n.x * ixy, Nxxz = n.x * izx, Nyyz = n.y * iyz, Nyyx = n.y * ixy, Nzzx = n.z * izx, Nzzy = n.z * iyz;
        d.n2Nxxy_Nyxx = 2 * Nxxy + n.y * ixx;
        d.n2Nxxz_Nzxx = 2 * Nxxz + n.z * ixx;
        d.n2Nyyz_Nzyy = 2 * Nyyz + n.z * iyy;
        d.n2Nyyx_Nxyy = 2 * Nyyx + n.x * iyy;
        d.n2Nzzx_Nxzz = 2 * Nzzx + n.x * izz;
        d.n2Nzzy_Nyzz = 2 * Nzzy + n.y * izz;
    }
    // PrecomputeFunctors::post (:6872-7080)
    void post(int nodei, LocalData &out, int nchildren, const LocalData *ch) const {
        float *rec = tr.box.data() + (size_t)FWN_BOX_FLOATS * nodei;
        std::memset(rec, 0, FWN_BOX_FLOATS * sizeof(float));
        V3f N = ch[0].N, areaP = ch[0].areaP;
        float area = ch[0].area;
        for (int i = 0; i < nchildren; i++) {
            rec[FWN_N + i] = ch[i].N.x; rec[FWN_N + 4 + i] = ch[i].N.y; rec[FWN_N + 8 + i] = ch[i].N.z;
            rec[FWN_P + i] = ch[i].averageP.x; rec[FWN_P + 4 + i] = ch[i].averageP.y; rec[FWN_P + 8 + i] = ch[i].averageP.z;
            if (i > 0) { N = N + ch[i].N; areaP = areaP + ch[i].areaP; area += ch[i].area; }
        }
        out.N = N; out.areaP = areaP; out.area = area;
        Box3 box = ch[0].box;
        for (int i = 1; i < nchildren; i++) box.combine(ch[i].box);
        V3f averageP;
        if (area > 0) averageP = areaP / area;
        else averageP = 0.5f * (V3f{box.v[0][0], box.v[1][0], box.v[2][0]} + V3f{box.v[0][1], box.v[1][1], box.v[2][1]});
        out.averageP = averageP;
        out.box = box;
        for (int i = 0; i < nchildren; i++) {
            const Box3 &lb = ch[i].box;
            const V3f lp = ch[i].averageP;
            const V3f m = {std::max(lp.x - lb.v[0][0], lb.v[0][1] - lp.x), std::max(lp.y - lb.v[1][0], lb.v[1][1] - lp.y), std::max(lp.z - lb.v[2][0], lb.v[2][1] - lp.z)};
            rec[FWN_MAXP2 + i] = length2(m);
        }
        for (int i = nchildren; i < 4; i++) rec[FWN_MAXP2 + i] = std::numeric_limits<float>::infinity();
        out.NijDiag = ch[0].NijDiag;
        out.Nxy = out.Nyx = out.Nyz = out.Nzy = out.Nzx = out.Nxz = 0;
        out.NijkDiag = ch[0].NijkDiag;
        out.sumPermuteNxyz = ch[0].sumPermuteNxyz;
        out.n2Nxxy_Nyxx = ch[0].n2Nxxy_Nyxx; out.n2Nxxz_Nzxx = ch[0].n2Nxxz_Nzxx; out.n2Nyyz_Nzyy = ch[0].n2Nyyz_Nzyy;
        out.n2Nyyx_Nxyy = ch[0].n2Nyyx_Nxyy; out.n2Nzzx_Nxzz = ch[0].n2Nzzx_Nxzz; out.n2Nzzy_Nyzz = ch[0].n2Nzzy_Nyzz;
        for (int i = 1; i < nchildren; i++) {
            out.NijDiag = out.NijDiag + ch[i].NijDiag;
            out.NijkDiag = out.NijkDiag + ch[i].NijkDiag;
            out.sumPermuteNxyz += ch[i].sumPermuteNxyz;
            out.n2Nxxy_Nyxx += ch[i].n2Nxxy_Nyxx; out.n2Nxxz_Nzxx += ch[i].n2Nxxz_Nzxx; out.n2Nyyz_Nzyy += ch[i].n2Nyyz_Nzyy;
            out.n2Nyyx_Nxyy += ch[i].n2Nyyx_Nxyy; out.n2Nzzx_Nxzz += ch[i].n2Nzzx_Nxzz; out.n2Nzzy_Nyzz += ch[i].n2Nzzy_Nyzz;
        }
        for (int i = 0; i < nchildren; i++) {
            rec[FWN_NII + i] = ch[i].NijDiag.x; rec[FWN_NII + 4 + i] = ch[i].NijDiag.y; rec[FWN_NII + 8 + i] = ch[i].NijDiag.z;
            rec[FWN_NXY + i] = ch[i].Nxy + ch[i].Nyx; rec[FWN_NYZ + i] = ch[i].Nyz + ch[i].Nzy; rec[FWN_NZX + i] = ch[i].Nzx + ch[i].Nxz;
            rec[FWN_NIII + i] = ch[i].NijkDiag.x; rec[FWN_NIII + 4 + i] = ch[i].NijkDiag.y; rec[FWN_NIII + 8 + i] = ch[i].NijkDiag.z;
            rec[FWN_SUMPERM + i] = ch[i].sumPermuteNxyz;
            rec[FWN_2NXXY + i] = ch[i].n2Nxxy_Nyxx; rec[FWN_2NXXZ + i] = ch[i].n2Nxxz_Nzxx; rec[FWN_2NYYZ + i] = ch[i].n2Nyyz_Nzyy;
            rec[FWN_2NYYX + i] = ch[i].n2Nyyx_Nxyy; rec[FWN_2NZZX + i] = ch[i].n2Nzzx_Nxzz; rec[FWN_2NZZY + i] = ch[i].n2Nzzy_Nyzz;
        }
        for (int i = 0; i < nchildren; i++) {
            const LocalData &cd = ch[i];
            const V3f dsp = cd.averageP - out.averageP;
            const V3f Nc = cd.N;
            out.NijDiag = out.NijDiag + Nc * dsp;
            const float Nxy = cd.Nxy + Nc.x * dsp.y, Nyx = cd.Nyx + Nc.y * dsp.x, Nyz = cd.Nyz + Nc.y * dsp.z, Nzy = cd.Nzy + Nc.z * dsp.y,
                        Nzx = cd.Nzx + Nc.z * dsp.x, Nxz = cd.Nxz + Nc.x * dsp.z;
            out.Nxy += Nxy; out.Nyx += Nyx; out.Nyz += Nyz; out.Nzy += Nzy; out.Nzx += Nzx; out.Nxz += Nxz;
            out.NijkDiag = out.NijkDiag + (2.0f * dsp * cd.NijDiag + dsp * dsp * cd.N);
            out.sumPermuteNxyz += (dsp.x * (Nyz + Nzy) + dsp.y * (Nzx + Nxz) + dsp.z * (Nxy + Nyx));
            out.n2Nxxy_Nyxx += 2 * (dsp.y * cd.NijDiag.x + dsp.x * cd.Nxy + Nc.x * dsp.x * dsp.y) + 2 * cd.Nyx * dsp.x + Nc.y * dsp.x * dsp.x;
            out.n2Nxxz_Nzxx += 2 * (dsp.z * cd.NijDiag.x + dsp.x * cd.Nxz + Nc.x * dsp.x * dsp.z) + 2 * cd.Nzx * dsp.x + Nc.z * dsp.x * dsp.x;
            out.n2Nyyz_Nzyy += 2 * (dsp.z * cd.NijDiag.y + dsp.y * cd.Nyz + Nc.y * dsp.y * dsp.z) + 2 * cd.Nzy * dsp.y + Nc.z * dsp.y * dsp.y;
            out.n2Nyyx_Nxyy += 2 * (dsp.x * cd.NijDiag.y + dsp.y * cd.Nyx + Nc.y * dsp.y * dsp.x) + 2 * cd.Nxy * dsp.y + Nc.x * dsp.y * dsp.y;
            out.n2Nzzx_Nxzz += 2 * (dsp.x * cd.NijDiag.z + dsp.z * cd.Nzx + Nc.z * dsp.z * dsp.x) + 2 * cd.Nxz * dsp.z + Nc.x * dsp.z * dsp.z;
            out.n2Nzzy_Nyzz += 2 * (dsp.y * cd.NijDiag.z + dsp.z * cd.Nzy + Nc.z * dsp.z * dsp.y) + 2 * cd.Nyz * dsp.z + Nc.y * dsp.z * dsp.z;
        }
    }
    // BVH<N>::traverseHelper (:4626-4653), post-order
    void traverse(int nodei, LocalData &for_parent) const {
        LocalData local[4];
        int s;
        for (s = 0; s < 4; ++s) {
            const int32_t ci = tr.child[4 * nodei + s];
            if (ci < 0) {
                if (ci == FWN_EMPTY) break;
                traverse((int)((uint32_t)ci & ~FWN_INTERNAL_BIT), local[s]);
            } else item(ci, local[s]);
        }
        post(nodei, for_parent, s, local);
    }
};

}  // namespace fwn_detail

// vertices: nV x 3 doubles (cast to float like fast_winding_number.cpp does), faces: nF x 3
inline void fwn_build(const double *vertices, int nV, const int32_t *faces, int nF, FwnTree &tr) {
    using namespace fwn_detail;
    tr.pos.resize((size_t)3 * nV);
    for (int i = 0; i < 3 * nV; i++) tr.pos[i] = (float)vertices[i];
    tr.tri.assign(faces, faces + (size_t)3 * nF);
    std::vector<Box3> boxes(nF);
    for (int f = 0; f < nF; f++) {
        Box3 &b = boxes[f];
        for (int a = 0; a < 3; a++) b.v[a][0] = b.v[a][1] = tr.pos[3 * faces[3 * f] + a];
        for (int k = 1; k < 3; k++)
            for (int a = 0; a < 3; a++) {
                const float p = tr.pos[3 * faces[3 * f + k] + a];
                b.v[a][0] = std::min(b.v[a][0], p);
                b.v[a][1] = std::max(b.v[a][1], p);
            }
    }
    // boxes with NaN / infinite coordinates are left out (utExcludeNaNInfBoxIndices :4512-4551)
    std::vector<int> indices;
    indices.reserve(nF);
    for (int f = 0; f < nF; f++) {
        bool bad = false;
        for (int a = 0; a < 3; a++) bad |= !std::isfinite(boxes[f].v[a][0]) || !std::isfinite(boxes[f].v[a][1]);
        if (!bad) indices.push_back(f);
    }
    tr.child.clear();
    tr.box.clear();
    if (indices.empty()) return;
    Box3 full = boxes[indices[0]];
    for (size_t i = 1; i < indices.size(); i++) full.combine(boxes[indices[i]]);
    Builder B;
    B.boxes = boxes.data();
    B.nodes.assign(4, FWN_EMPTY);
    B.init_node(0, full, indices.data(), (int)indices.size());
    tr.child = B.nodes;
    tr.box.assign((size_t)FWN_BOX_FLOATS * tr.n_nodes(), 0.f);
    Precompute P{tr, boxes.data()};
    LocalData root;
    P.traverse(0, root);
}

// depth of the hierarchy (root = 1): bounds the device traversal stack
inline int fwn_depth(const FwnTree &tr) {
    if (tr.child.empty()) return 0;
    std::vector<std::pair<int, int>> stack{{0, 1}};
    int depth = 0;
    while (!stack.empty()) {
        const auto [node, d] = stack.back();
        stack.pop_back();
        depth = std::max(depth, d);
        for (int s = 0; s < 4; s++) {
            const int32_t ci = tr.child[4 * node + s];
            if (ci != FWN_EMPTY && ci < 0) stack.push_back({(int)((uint32_t)ci & ~FWN_INTERNAL_BIT), d + 1});
        }
    }
    return depth;
}

// Host evaluation of the solid angle (the device walks the same records: csrc/dev_mesh.hpp); used by the tests.
// UT_SolidAngle::computeSolidAngle (:7119-7258): returns the solid angle, w = result / (4 pi).
inline float fwn_tri_solid_angle(const float *a, const float *b, const float *c, const float *q) {     // UTsignedSolidAngleTri (:6044-6082)
    using namespace fwn_detail;
    V3f qa{a[0] - q[0], a[1] - q[1], a[2] - q[2]}, qb{b[0] - q[0], b[1] - q[1], b[2] - q[2]}, qc{c[0] - q[0], c[1] - q[1], c[2] - q[2]};
    const float al = std::sqrt(length2(qa)), bl = std::sqrt(length2(qb)), cl = std::sqrt(length2(qc));
    if (al == 0 || bl == 0 || cl == 0) return 0.f;
    qa = qa / al; qb = qb / bl; qc = qc / cl;
    const V3f cr = cross(qb - qa, qc - qa);
    const float numerator = qa.x * cr.x + qa.y * cr.y + qa.z * cr.z;
    if (numerator == 0) return 0.f;
    const float denominator = 1.0f + (qa.x * qb.x + qa.y * qb.y + qa.z * qb.z) + (qa.x * qc.x + qa.y * qc.y + qa.z * qc.z) + (qb.x * qc.x + qb.y * qc.y + qb.z * qc.z);
    return 2.0f * std::atan2(numerator, denominator);
}
inline float fwn_node_eval(const FwnTree &tr, int nodei, const float *q, float acc2) {
    const float *r = tr.box.data() + (size_t)FWN_BOX_FLOATS * nodei;
    float approx[4];
    unsigned descend = 0;
    for (int l = 0; l < 4; l++) {
        float qx = q[0] - r[FWN_P + l], qy = q[1] - r[FWN_P + 4 + l], qz = q[2] - r[FWN_P + 8 + l];
        const float ql2 = qx * qx + qy * qy + qz * qz;
        const bool desc = ql2 <= r[FWN_MAXP2 + l] * acc2;
        const float m2 = 1.0f / ql2, m1 = std::sqrt(m2);
        qx *= m1; qy *= m1; qz *= m1;
        float om = -m2 * (qx * r[FWN_N + l] + qy * r[FWN_N + 4 + l] + qz * r[FWN_N + 8 + l]);
        const float q2x = qx * qx, q2y = qy * qy, q2z = qz * qz;
        const float m3 = m2 * m1;
        const float om1 = m3 * (r[FWN_NII + l] + r[FWN_NII + 4 + l] + r[FWN_NII + 8 + l]
                                - 3.0f * ((q2x * r[FWN_NII + l] + q2y * r[FWN_NII + 4 + l] + q2z * r[FWN_NII + 8 + l]) + qx * qy * r[FWN_NXY + l] + qx * qz * r[FWN_NZX + l] + qy * qz * r[FWN_NYZ + l]));
        om += om1;
        const float q3x = q2x * qx, q3y = q2y * qy, q3z = q2z * qz;
        const float m4 = m2 * m2;
        const float t0x = r[FWN_2NYYX + l] + r[FWN_2NZZX + l], t0y = r[FWN_2NZZY + l] + r[FWN_2NXXY + l], t0z = r[FWN_2NXXZ + l] + r[FWN_2NYYZ + l];
        const float t1x = qy * r[FWN_2NXXY + l] + qz * r[FWN_2NXXZ + l], t1y = qz * r[FWN_2NYYZ + l] + qx * r[FWN_2NYYX + l], t1z = qx * r[FWN_2NZZX + l] + qy * r[FWN_2NZZY + l];
        const float om2 = m4 * (1.5f * (qx * (3.0f * r[FWN_NIII + l] + t0x) + qy * (3.0f * r[FWN_NIII + 4 + l] + t0y) + qz * (3.0f * r[FWN_NIII + 8 + l] + t0z))
                                - 7.5f * ((q3x * r[FWN_NIII + l] + q3y * r[FWN_NIII + 4 + l] + q3z * r[FWN_NIII + 8 + l]) + qx * qy * qz * r[FWN_SUMPERM + l] + (q2x * t1x + q2y * t1y + q2z * t1z)));
        om += om2;
        const bool use = std::isfinite(om) && !desc;
        approx[l] = use ? om : 0.f;
        if (!use) descend |= 1u << l;
    }
    float sum = approx[0];
    for (int l = 1; l < 4; l++) sum += approx[l];
    if (!descend) return sum;
    float child_val[4] = {0, 0, 0, 0};
    int s;
    for (s = 0; s < 4; ++s) {
        if ((descend >> s) & 1) {
            const int32_t ci = tr.child[4 * nodei + s];
            if (ci < 0) {
                if (ci == FWN_EMPTY) { descend &= (1u << s) - 1; break; }
                child_val[s] = fwn_node_eval(tr, (int)((uint32_t)ci & ~FWN_INTERNAL_BIT), q, acc2);
            } else {
                const int32_t *t = tr.tri.data() + 3 * ci;
                child_val[s] = fwn_tri_solid_angle(&tr.pos[3 * t[0]], &tr.pos[3 * t[1]], &tr.pos[3 * t[2]], q);
            }
        }
    }
    float dsum = (descend & 1) ? child_val[0] : 0.f;
    for (int i = 1; i < s; ++i) dsum += ((descend >> i) & 1) ? child_val[i] : 0.f;
    return sum + dsum;
}
inline float fwn_solid_angle(const FwnTree &tr, const float q[3], float accuracy_scale) {
    if (tr.child.empty()) return 0.f;
    return fwn_node_eval(tr, 0, q, accuracy_scale * accuracy_scale);
}

}  // namespace isdf_host

#ifdef __clang__
#pragma clang fp contract(fast)    // back to the compiler default for the including file
#endif
