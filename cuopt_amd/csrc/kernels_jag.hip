// gfx950 kernels of the jag layout (one translation unit per layout: a kernel change recompiles this file only).
// Launched from pdlp_device.hip through the prototypes of pdlp_kernel_decls.hpp.
#include <hip/hip_runtime.h>

#include "pdlp_kernel_decls.hpp"
#include "spmv_jag.hpp"

// jagged-layout twins of (2) and (3) and of the plain / ping-pong SpMV: same epilogues, LDS column sets
template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64)
k_jag_a_dual(JagView J, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ xbar,
             double* __restrict__ y0, double* __restrict__ y1, const double* __restrict__ lo,
             const double* __restrict__ hi, double* __restrict__ sumy, double* __restrict__ part, double* __restrict__ ycopy,
             const p2pdev::Push* __restrict__ push)
{
  if (!loop_active(ctl)) return;
  const int cur = ctl->cur;
  DualEpilogue e{cur ? y1 : y0, cur ? y0 : y1, lo, hi, sumy, ctl->sigma, ctl->step_size,
                 ctl->pending_avg != 0, ycopy, push};
  jag_block<decltype(e), WAVES>(J, xbar, e, part);
  if (push) p2pdev::count_exchange(push);
}

template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64)
k_jag_at_step(JagView J, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ y0,
              const double* __restrict__ y1, const double* __restrict__ x0,
              const double* __restrict__ x1, double* __restrict__ aty0, double* __restrict__ aty1,
              double* __restrict__ part)
{
  if (!loop_active(ctl)) return;
  const int cur = ctl->cur;
  StepEpilogue e{cur ? x1 : x0, cur ? x0 : x1, cur ? aty1 : aty0, cur ? aty0 : aty1};
  jag_block<decltype(e), WAVES>(J, cur ? y0 : y1 /* y' */, e, part);
}

template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64)
k_jag_at_cur(JagView J, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ y0,
             const double* __restrict__ y1, double* __restrict__ aty0, double* __restrict__ aty1,
             double* __restrict__ out_override, int use_next)
{
  const int cur = ctl->cur ^ (use_next ? 1 : 0);
  StoreEpilogue e{out_override ? out_override : (cur ? aty1 : aty0)};
  jag_block<decltype(e), WAVES>(J, cur ? y1 : y0, e, nullptr);
}

template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64)
k_jag_plain(JagView J, const double* __restrict__ vec, double* __restrict__ out)
{
  StoreEpilogue e{out};
  jag_block<decltype(e), WAVES>(J, vec, e, nullptr);
}

template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64)
k_jag_eval_primal(JagView J, const pdlpdev_ctl* __restrict__ ctl, int which,
                  const double* __restrict__ x0, const double* __restrict__ x1,
                  const double* __restrict__ avgx, const double* __restrict__ y0,
                  const double* __restrict__ y1, const double* __restrict__ avgy,
                  const double* __restrict__ dr, const double* __restrict__ lo_u,
                  const double* __restrict__ hi_u, double eps_rel, double* __restrict__ linf_rows,
                  double* __restrict__ ax_out, double* __restrict__ part)
{
  const int cur = ctl->cur;
  const double* xv = which == PDLPDEV_AVERAGE ? avgx : (cur ? x1 : x0);
  const double* yv = which == PDLPDEV_AVERAGE ? avgy : (cur ? y1 : y0);
  EvalPrimalEpilogue e{yv, dr, lo_u, hi_u, eps_rel, linf_rows, ax_out};
  jag_block<decltype(e), WAVES>(J, xv, e, part);
}

template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64)
k_jag_eval_dual(JagView J, const pdlpdev_ctl* __restrict__ ctl, int which,
                const double* __restrict__ x0, const double* __restrict__ x1,
                const double* __restrict__ avgx, const double* __restrict__ y0,
                const double* __restrict__ y1, const double* __restrict__ avgy, EvalDualCore core,
                double* __restrict__ part)
{
  const int cur = ctl->cur;
  core.xhat     = which == PDLPDEV_AVERAGE ? avgx : (cur ? x1 : x0);
  const double* yv = which == PDLPDEV_AVERAGE ? avgy : (cur ? y1 : y0);
  EvalDualEpilogue e{core};
  jag_block<decltype(e), WAVES>(J, yv, e, part);
}

// explicit instantiations (the launch sites live in another translation unit)
template __global__ void k_jag_a_dual<8>(JagView J, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ xbar,
             double* __restrict__ y0, double* __restrict__ y1, const double* __restrict__ lo,
             const double* __restrict__ hi, double* __restrict__ sumy, double* __restrict__ part, double* __restrict__ ycopy,
             const p2pdev::Push* __restrict__ push);
template __global__ void k_jag_a_dual<16>(JagView J, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ xbar,
             double* __restrict__ y0, double* __restrict__ y1, const double* __restrict__ lo,
             const double* __restrict__ hi, double* __restrict__ sumy, double* __restrict__ part, double* __restrict__ ycopy,
             const p2pdev::Push* __restrict__ push);
template __global__ void k_jag_at_step<8>(JagView J, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ y0,
              const double* __restrict__ y1, const double* __restrict__ x0,
              const double* __restrict__ x1, double* __restrict__ aty0, double* __restrict__ aty1,
              double* __restrict__ part);
template __global__ void k_jag_at_step<16>(JagView J, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ y0,
              const double* __restrict__ y1, const double* __restrict__ x0,
              const double* __restrict__ x1, double* __restrict__ aty0, double* __restrict__ aty1,
              double* __restrict__ part);
template __global__ void k_jag_at_cur<8>(JagView J, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ y0,
             const double* __restrict__ y1, double* __restrict__ aty0, double* __restrict__ aty1,
             double* __restrict__ out_override, int use_next);
template __global__ void k_jag_at_cur<16>(JagView J, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ y0,
             const double* __restrict__ y1, double* __restrict__ aty0, double* __restrict__ aty1,
             double* __restrict__ out_override, int use_next);
template __global__ void k_jag_plain<8>(JagView J, const double* __restrict__ vec, double* __restrict__ out);
template __global__ void k_jag_plain<16>(JagView J, const double* __restrict__ vec, double* __restrict__ out);
template __global__ void k_jag_eval_primal<8>(JagView J, const pdlpdev_ctl* __restrict__ ctl, int which,
                  const double* __restrict__ x0, const double* __restrict__ x1,
                  const double* __restrict__ avgx, const double* __restrict__ y0,
                  const double* __restrict__ y1, const double* __restrict__ avgy,
                  const double* __restrict__ dr, const double* __restrict__ lo_u,
                  const double* __restrict__ hi_u, double eps_rel, double* __restrict__ linf_rows,
                  double* __restrict__ ax_out, double* __restrict__ part);
template __global__ void k_jag_eval_primal<16>(JagView J, const pdlpdev_ctl* __restrict__ ctl, int which,
                  const double* __restrict__ x0, const double* __restrict__ x1,
                  const double* __restrict__ avgx, const double* __restrict__ y0,
                  const double* __restrict__ y1, const double* __restrict__ avgy,
                  const double* __restrict__ dr, const double* __restrict__ lo_u,
                  const double* __restrict__ hi_u, double eps_rel, double* __restrict__ linf_rows,
                  double* __restrict__ ax_out, double* __restrict__ part);
template __global__ void k_jag_eval_dual<8>(JagView J, const pdlpdev_ctl* __restrict__ ctl, int which,
                const double* __restrict__ x0, const double* __restrict__ x1,
                const double* __restrict__ avgx, const double* __restrict__ y0,
                const double* __restrict__ y1, const double* __restrict__ avgy, EvalDualCore core,
                double* __restrict__ part);
template __global__ void k_jag_eval_dual<16>(JagView J, const pdlpdev_ctl* __restrict__ ctl, int which,
                const double* __restrict__ x0, const double* __restrict__ x1,
                const double* __restrict__ avgx, const double* __restrict__ y0,
                const double* __restrict__ y1, const double* __restrict__ avgy, EvalDualCore core,
                double* __restrict__ part);
