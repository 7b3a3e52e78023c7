// Builds the vendored HDK fast-winding-number core (src/utils/include/igl/FastWindingNumberForSoups.h, included
// from /root/reference, not copied) behind a C ABI.  The thin libigl wrapper fast_winding_number.cpp is NOT in
// /root/reference; what it does per its header doc and call sites (Shape.cpp:86,110) is restated
// here: cast V to float, flatten F, init(order), w = computeSolidAngle(q, accuracy_scale) / (4*pi).
// TEST INFRASTRUCTURE ONLY.
#include <vector>
#include <cstring>
// test infrastructure only: look inside UT_SolidAngle (tree + per-node expansion data) to validate the product's own
// restatement of the HDK hierarchy (csrc/fwn_host.hpp) node by node
#define private public
#define protected public
#include <igl/FastWindingNumberForSoups.h>
#undef private
#undef protected
// the definition lives in libigl's absent default_num_threads.cpp; BVH build is single-threaded here
namespace igl { unsigned int default_num_threads(unsigned int) { return 1; } }
#include <cmath>
using namespace igl::FastWindingNumber::HDK_Sample;
namespace {
struct Bvh {
    UT_SolidAngle<float, float> sa;
    std::vector<UT_Vector3T<float>> U;
    std::vector<int> F;
};
}
extern "C" {
void *ref_fwn_create(const double *V, int nV, const int *F, int nF, int order) {
    Bvh *b = new Bvh();
    b->U.resize(nV);
    for (int i = 0; i < nV; i++) { b->U[i][0] = (float)V[3 * i]; b->U[i][1] = (float)V[3 * i + 1]; b->U[i][2] = (float)V[3 * i + 2]; }
    b->F.assign(F, F + 3 * (size_t)nF);
    b->sa.init(nF, b->F.data(), nV, b->U.data(), order);
    return b;
}
void ref_fwn_destroy(void *h) { delete (Bvh *)h; }
void ref_fwn_eval(void *h, const double *Q, int n, double accuracy, double *W) {
    Bvh *b = (Bvh *)h;
    for (int i = 0; i < n; i++) {
        UT_Vector3T<float> q;
        q[0] = (float)Q[3 * i]; q[1] = (float)Q[3 * i + 1]; q[2] = (float)Q[3 * i + 2];
        W[i] = b->sa.computeSolidAngle(q, (float)accuracy) / (4.0 * M_PI);
    }
}
int ref_fwn_num_nodes(void *h) { return ((Bvh *)h)->sa.myTree.getNumNodes(); }
// child[4] per node: leaf = triangle index, -1 = empty, internal = node number | 0x80000000
void ref_fwn_dump_nodes(void *h, int *out) {
    Bvh *b = (Bvh *)h;
    const int n = b->sa.myTree.getNumNodes();
    const auto *nodes = b->sa.myTree.getNodes();
    for (int i = 0; i < n; i++) for (int c = 0; c < 4; c++) out[4 * i + c] = (int)nodes[i].child[c];
}
// BoxData per node as 92 floats in declaration order, 4 lanes (children) each:
// maxPDist2 | averageP xyz | N xyz | NijDiag xyz | Nxy+Nyx | Nyz+Nzy | Nzx+Nxz | NijkDiag xyz | sumPermuteNxyz |
// 2Nxxy+Nyxx | 2Nxxz+Nzxx | 2Nyyz+Nzyy | 2Nyyx+Nxyy | 2Nzzx+Nxzz | 2Nzzy+Nyzz
int ref_fwn_boxdata_floats(void) { return (int)(sizeof(UT_SolidAngle<float, float>::BoxData) / sizeof(float)); }
void ref_fwn_dump_boxdata(void *h, float *out) {
    Bvh *b = (Bvh *)h;
    const int n = b->sa.myTree.getNumNodes();
    std::memcpy(out, b->sa.myData.get(), (size_t)n * sizeof(UT_SolidAngle<float, float>::BoxData));
}
}
