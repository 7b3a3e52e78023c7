// Builds the vendored HDK fast-winding-number core (src/utils/include/igl/FastWindingNumberForSoups.h, included
// from /root/reference, not copied) behind a C ABI.  The thin libigl wrapper fast_winding_number.cpp is NOT in
// /root/reference; what it does per its header doc and call sites (Shape.cpp:86,110) is restated
// here: cast V to float, flatten F, init(order), w = computeSolidAngle(q, accuracy_scale) / (4*pi).
// TEST INFRASTRUCTURE ONLY.
#include <vector>
#include <igl/FastWindingNumberForSoups.h>
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
}
