// Device half of the objective callback as kernels of its own (the bodies and what they compute: csrc/minco_dev_body.hpp): one
// workgroup per trajectory, launched either side of the sweeps - the form of every callback whose sweep is not one fused launch
// (swept-volume ctx, long trajectories, the batched optimiser's rounds, sharded ranks).
#include "isdf_ctx.hpp"
#include "minco_dev_body.hpp"

namespace isdf {

template <bool SPLIT> __global__ __launch_bounds__(SPLIT ? 1024 : 448) void cb_pre_kernel(const CbDev P) {
    extern __shared__ double s_mem[];
    __shared__ double s_part[16];
    cbd::cb_pre_body<SPLIT>(P, (int)blockIdx.x, SPLIT ? (int)blockDim.x / 3 : (int)blockDim.x, s_mem, s_part);
}
template <bool SPLIT> __global__ __launch_bounds__(SPLIT ? 1024 : 448) void cb_post_kernel(const CbDev P) {
    extern __shared__ double s_mem[];
    __shared__ double s_part[16];
    cbd::cb_post_body<SPLIT>(P, (int)blockIdx.x, (int)gridDim.x, SPLIT ? (int)blockDim.x / 3 : (int)blockDim.x, s_mem, s_part);
}

static bool cb_split(int N) { return N <= CB_SPLIT_MAX_N; }
static int cb_threads(int N) { return ((N + 63) / 64) * 64 * (cb_split(N) ? 3 : 1); }

void launch_cb_pre(const CbDev &P, hipStream_t st) {
    const size_t lds = cbd::cb_lds_doubles(P.N, cb_split(P.N)) * sizeof(double);
    if (cb_split(P.N)) hipLaunchKernelGGL(cb_pre_kernel<true>, dim3(P.nb), dim3(cb_threads(P.N)), lds, st, P);
    else hipLaunchKernelGGL(cb_pre_kernel<false>, dim3(P.nb), dim3(cb_threads(P.N)), lds, st, P);
}
void launch_cb_post(const CbDev &P, hipStream_t st) {
    const size_t lds = cbd::cb_lds_doubles(P.N, cb_split(P.N)) * sizeof(double);
    if (cb_split(P.N)) hipLaunchKernelGGL(cb_post_kernel<true>, dim3(P.nb), dim3(cb_threads(P.N)), lds, st, P);
    else hipLaunchKernelGGL(cb_post_kernel<false>, dim3(P.nb), dim3(cb_threads(P.N)), lds, st, P);
}

}  // namespace isdf
