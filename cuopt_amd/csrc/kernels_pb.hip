// gfx950 kernels of the pb layout (one translation unit per layout: a kernel change recompiles this file only).
// Launched from pdlp_device.hip through the prototypes of pdlp_kernel_decls.hpp.
#include <hip/hip_runtime.h>

#include "pdlp_kernel_decls.hpp"
#include "spmv_pb.hpp"

// gather-free twins: phase P (one kernel, the gathered vector chosen on the device like the other layouts do) and phase R with
// the same epilogues
template <int THREADS>
__global__ void __launch_bounds__(THREADS)
k_pb_products(PbView V, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ v0, const double* __restrict__ v1, int mode, int in_loop)
{
  extern __shared__ __attribute__((aligned(16))) double pb_lds[];
  if (in_loop && !loop_active(ctl)) return;
  // mode 0: v0.  1: cur ? v0 : v1 (the trial iterate of a ping-pong pair).  2: cur ? v1 : v0 (the current one).
  const double* vec = v0;
  if (mode != 0) {
    const bool cur = ctl->cur != 0;
    vec            = (cur == (mode == 1)) ? v0 : v1;
  }
  pb_products_block<THREADS>(V, vec, pb_lds);
}

__global__ void __launch_bounds__(kPbThreads)
k_pb_a_dual(PbView V, const pdlpdev_ctl* __restrict__ ctl, double* __restrict__ y0, double* __restrict__ y1, const double* __restrict__ lo,
            const double* __restrict__ hi, double* __restrict__ sumy, double* __restrict__ part, double* __restrict__ ycopy,
            const p2pdev::Push* __restrict__ push)
{
  extern __shared__ __attribute__((aligned(16))) double pb_lds[];
  if (!loop_active(ctl)) return;
  const int cur = ctl->cur;
  DualEpilogue e{cur ? y1 : y0, cur ? y0 : y1, lo, hi, sumy, ctl->sigma, ctl->step_size, ctl->pending_avg != 0, ycopy, push};
  pb_rows_block(V, e, part, pb_lds);
  if (push) p2pdev::count_exchange(push);
}

__global__ void __launch_bounds__(kPbThreads)
k_pb_at_step(PbView V, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ x0, const double* __restrict__ x1,
             double* __restrict__ aty0, double* __restrict__ aty1, double* __restrict__ part)
{
  extern __shared__ __attribute__((aligned(16))) double pb_lds[];
  if (!loop_active(ctl)) return;
  const int cur = ctl->cur;
  StepEpilogue e{cur ? x1 : x0, cur ? x0 : x1, cur ? aty1 : aty0, cur ? aty0 : aty1};
  pb_rows_block(V, e, part, pb_lds);
}

__global__ void __launch_bounds__(kPbThreads)
k_pb_at_cur(PbView V, const pdlpdev_ctl* __restrict__ ctl, double* __restrict__ aty0, double* __restrict__ aty1,
            double* __restrict__ out_override, int use_next)
{
  extern __shared__ __attribute__((aligned(16))) double pb_lds[];
  const int cur = ctl->cur ^ (use_next ? 1 : 0);
  StoreEpilogue e{out_override ? out_override : (cur ? aty1 : aty0)};
  pb_rows_block(V, e, nullptr, pb_lds);
}

__global__ void __launch_bounds__(kPbThreads) k_pb_plain(PbView V, double* __restrict__ out)
{
  extern __shared__ __attribute__((aligned(16))) double pb_lds[];
  StoreEpilogue e{out};
  pb_rows_block(V, e, nullptr, pb_lds);
}

__global__ void __launch_bounds__(kPbThreads)
k_pb_eval_primal(PbView V, const pdlpdev_ctl* __restrict__ ctl, int which, const double* __restrict__ y0, const double* __restrict__ y1,
                 const double* __restrict__ avgy, const double* __restrict__ dr, const double* __restrict__ lo_u,
                 const double* __restrict__ hi_u, double eps_rel, double* __restrict__ linf_rows, double* __restrict__ ax_out,
                 double* __restrict__ part)
{
  extern __shared__ __attribute__((aligned(16))) double pb_lds[];
  const int cur    = ctl->cur;
  const double* yv = which == PDLPDEV_AVERAGE ? avgy : (cur ? y1 : y0);
  EvalPrimalEpilogue e{yv, dr, lo_u, hi_u, eps_rel, linf_rows, ax_out};
  pb_rows_block(V, e, part, pb_lds);
}

__global__ void __launch_bounds__(kPbThreads)
k_pb_eval_dual(PbView V, const pdlpdev_ctl* __restrict__ ctl, int which, const double* __restrict__ x0, const double* __restrict__ x1,
               const double* __restrict__ avgx, EvalDualCore core, double* __restrict__ part)
{
  extern __shared__ __attribute__((aligned(16))) double pb_lds[];
  const int cur = ctl->cur;
  core.xhat     = which == PDLPDEV_AVERAGE ? avgx : (cur ? x1 : x0);
  EvalDualEpilogue e{core};
  pb_rows_block(V, e, part, pb_lds);
}

// explicit instantiations (the launch sites live in another translation unit)
template __global__ void k_pb_products<512>(PbView V, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ v0, const double* __restrict__ v1, int mode, int in_loop);
template __global__ void k_pb_products<1024>(PbView V, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ v0, const double* __restrict__ v1, int mode, int in_loop);
