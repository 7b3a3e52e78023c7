// Device MINCO for batches (csrc/minco_dev.hip): parameter block and launchers.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>

namespace isdf {

struct MincoDevParams {
    int N;                  // pieces per trajectory; variables per trajectory n = N + 3 (N - 1)
    double rho;
    const double *x;        // [n_active][n]      tau | inner waypoints (waypoint-major), the optimizer's variables
    const int *ids;         // [n_active]         trajectory number of every active slot (selects its boundary states)
    const double *cond;     // [n_traj][18]       head (3x3 column-major: pos | vel | acc) | tail, per trajectory
    double *T;              // [n_active][N]      out (forward): durations, the sweep's input
    double *coeffs;         // [n_active][18 N]   out (forward): coefficients, 6N x 3 column-major per trajectory, the sweep's input
    double *lu;             // [n_active][13 * 6N] out (forward) / in (backward): the band factors
    const double *out;      // [n_active][1 + 19 N] in (backward): the sweep's [cost | gradT | gradC]
    double *res;            // [n_active][1 + n]  out (backward): cost | g
};

size_t minco_forward_lds(int N);
size_t minco_backward_lds(int N);
void launch_minco_forward(const MincoDevParams &P, int n_active, hipStream_t stream);
void launch_minco_backward(const MincoDevParams &P, int n_active, hipStream_t stream);

}  // namespace isdf
