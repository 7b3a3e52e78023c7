// Parameter block of the device half of the objective callback (csrc/minco_dev.hip).  Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>

namespace isdf {

constexpr int CB_MAX_N = 400;             // pieces per trajectory the device callback takes (LDS: 19 N doubles, + 4 N for the staged inputs up to 320 pieces); beyond it MINCO runs on the host
constexpr int CB_SPLIT_MAX_N = 320;       // up to here the three axes of a row sit in three wavefront groups (3 x 320 threads)
constexpr int CB_AUTO_HOST_MAX_N = 64;    // minco mode 0: single-trajectory fused steps up to this many pieces keep MINCO on the host

struct CbDev {
    int N, nb, n_out;           // pieces per trajectory, trajectories of the launch, packed sweep blocks to add per trajectory
    int res_stride;             // doubles per trajectory in `res`: cost | g (N + 3 (N - 1)) | energy, swept, integral, rho * sum(T)
    const double *x;            // [nb][N + 3 (N - 1)] decision variables: tau | inner waypoints (point-major)
    const double *ends;         // [nb][18]: head (pos, vel, acc), tail (pos, vel, acc)
    double *T;                  // [nb * N]          } the sweeps' inputs
    double *coeffs;             // [nb][6N x 3 col-major]
    double *u;                  // [nb][N + 1][3][2] junction states (v, a)
    double *epart;              // [nb][1 + 19 N] energy | its partial by T | by the coefficients (the sweeps' packed layout)
    const double *sweep;        // [n_out][nb][1 + 19 N] what the sweeps accumulated
    double rho;
    double *res;                // [nb][res_stride]
    unsigned long long *flag;   // [nb] completion words (res and flag host-mapped), or null
    unsigned long long seq;
    const unsigned long long *stats;   // the sweeps' statistics words ([4] = overflow), or null
};

void launch_cb_pre(const CbDev &P, hipStream_t st);
void launch_cb_post(const CbDev &P, hipStream_t st);

}  // namespace isdf
