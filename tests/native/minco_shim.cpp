// Test shim: exposes the PRODUCT's host MINCO (implicit-sdf-planner_amd/csrc/minco_host.hpp) to the CPU-only tests
// without needing a HIP device (the C-ABI needs one to create a context).  Built by tests/test_minco_host.py with g++.
#include "minco_host.hpp"
extern "C" {
void shim_minco(int N, const double *head9, const double *tail9, const double *waypoints, const double *T,
                double *coeffs, double *energy, double *gdC, double *gdT) {
    isdf_host::MincoS3 m;
    m.set_conditions(head9, tail9, N);
    m.set_parameters(waypoints, T);
    for (int i = 0; i < 18 * N; i++) coeffs[i] = m.c[i];
    *energy = m.energy(gdC, gdT);
}
void shim_minco_propagate(int N, const double *head9, const double *tail9, const double *waypoints, const double *T,
                          const double *gradC, const double *gradT_partial, double *gradP, double *gradT) {
    isdf_host::MincoS3 m;
    m.set_conditions(head9, tail9, N);
    m.set_parameters(waypoints, T);
    m.propagate_grad(gradC, gradT_partial, gradP, gradT);
}
double shim_tau_to_T(double tau) { return isdf_host::tau_to_T(tau); }
double shim_T_to_tau(double T) { return isdf_host::T_to_tau(T); }
double shim_grad_T_to_tau(double tau, double g) { return isdf_host::grad_T_to_tau(tau, g); }
}
