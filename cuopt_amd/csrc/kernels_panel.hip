// gfx950 kernels of the panel layout (one translation unit per layout: a kernel change recompiles this file only).
// Launched from pdlp_device.hip through the prototypes of pdlp_kernel_decls.hpp.
#include <hip/hip_runtime.h>

#include "pdlp_kernel_decls.hpp"
#include "spmv_panel.hpp"

// panel-layout twins of (2) and (3): same epilogues, slab-major gather (pdlp_kernels.hpp)
template <bool SEG>
__global__ void __launch_bounds__(kPanelThreads)
k_panel_a_dual(PanelView P, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ xbar,
               double* __restrict__ y0, double* __restrict__ y1, const double* __restrict__ lo,
               const double* __restrict__ hi, double* __restrict__ sumy, double* __restrict__ part, double* __restrict__ ycopy,
               const p2pdev::Push* __restrict__ push)
{
  if (!loop_active(ctl)) return;
  const int cur = ctl->cur;
  DualEpilogue e{cur ? y1 : y0, cur ? y0 : y1, lo, hi, sumy, ctl->sigma, ctl->step_size,
                 ctl->pending_avg != 0, ycopy, push};
  panel_block<SEG>(P, xbar, e, part);
  if (push) p2pdev::count_exchange(push);
}

template <bool SEG>
__global__ void __launch_bounds__(kPanelThreads)
k_panel_at_step(PanelView P, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ y0,
                const double* __restrict__ y1, const double* __restrict__ x0,
                const double* __restrict__ x1, double* __restrict__ aty0, double* __restrict__ aty1,
                double* __restrict__ part)
{
  if (!loop_active(ctl)) return;
  const int cur = ctl->cur;
  StepEpilogue e{cur ? x1 : x0, cur ? x0 : x1, cur ? aty1 : aty0, cur ? aty0 : aty1};
  panel_block<SEG>(P, cur ? y0 : y1 /* y' */, e, part);
}

// panel twin of k_spmv_at_cur (A^T y of the iterate / of the trial iterate, optionally into `out_override`)
template <bool SEG>
__global__ void __launch_bounds__(kPanelThreads)
k_panel_at_cur(PanelView P, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ y0,
               const double* __restrict__ y1, double* __restrict__ aty0, double* __restrict__ aty1,
               double* __restrict__ out_override, int use_next)
{
  const int cur = ctl->cur ^ (use_next ? 1 : 0);
  StoreEpilogue e{out_override ? out_override : (cur ? aty1 : aty0)};
  panel_block<SEG>(P, cur ? y1 : y0, e, nullptr);
}

template <bool SEG>
__global__ void __launch_bounds__(kPanelThreads)
k_panel_plain(PanelView P, const double* __restrict__ vec, double* __restrict__ out)
{
  StoreEpilogue e{out};
  panel_block<SEG>(P, vec, e, nullptr);
}

template <bool SEG>
__global__ void __launch_bounds__(kPanelThreads)
k_panel_eval_primal(PanelView P, const pdlpdev_ctl* __restrict__ ctl, int which,
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
  panel_block<SEG>(P, xv, e, part);
}

template <bool SEG>
__global__ void __launch_bounds__(kPanelThreads)
k_panel_eval_dual(PanelView P, const pdlpdev_ctl* __restrict__ ctl, int which,
                  const double* __restrict__ x0, const double* __restrict__ x1,
                  const double* __restrict__ avgx, const double* __restrict__ y0,
                  const double* __restrict__ y1, const double* __restrict__ avgy, EvalDualCore core,
                  double* __restrict__ part)
{
  const int cur = ctl->cur;
  core.xhat     = which == PDLPDEV_AVERAGE ? avgx : (cur ? x1 : x0);
  const double* yv = which == PDLPDEV_AVERAGE ? avgy : (cur ? y1 : y0);
  EvalDualEpilogue e{core};
  panel_block<SEG>(P, yv, e, part);
}

// explicit instantiations (the launch sites live in another translation unit)
template __global__ void k_panel_a_dual<true>(PanelView P, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ xbar,
               double* __restrict__ y0, double* __restrict__ y1, const double* __restrict__ lo,
               const double* __restrict__ hi, double* __restrict__ sumy, double* __restrict__ part, double* __restrict__ ycopy,
               const p2pdev::Push* __restrict__ push);
template __global__ void k_panel_a_dual<false>(PanelView P, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ xbar,
               double* __restrict__ y0, double* __restrict__ y1, const double* __restrict__ lo,
               const double* __restrict__ hi, double* __restrict__ sumy, double* __restrict__ part, double* __restrict__ ycopy,
               const p2pdev::Push* __restrict__ push);
template __global__ void k_panel_at_step<true>(PanelView P, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ y0,
                const double* __restrict__ y1, const double* __restrict__ x0,
                const double* __restrict__ x1, double* __restrict__ aty0, double* __restrict__ aty1,
                double* __restrict__ part);
template __global__ void k_panel_at_step<false>(PanelView P, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ y0,
                const double* __restrict__ y1, const double* __restrict__ x0,
                const double* __restrict__ x1, double* __restrict__ aty0, double* __restrict__ aty1,
                double* __restrict__ part);
template __global__ void k_panel_at_cur<true>(PanelView P, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ y0,
               const double* __restrict__ y1, double* __restrict__ aty0, double* __restrict__ aty1,
               double* __restrict__ out_override, int use_next);
template __global__ void k_panel_at_cur<false>(PanelView P, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ y0,
               const double* __restrict__ y1, double* __restrict__ aty0, double* __restrict__ aty1,
               double* __restrict__ out_override, int use_next);
template __global__ void k_panel_plain<true>(PanelView P, const double* __restrict__ vec, double* __restrict__ out);
template __global__ void k_panel_plain<false>(PanelView P, const double* __restrict__ vec, double* __restrict__ out);
template __global__ void k_panel_eval_primal<true>(PanelView P, const pdlpdev_ctl* __restrict__ ctl, int which,
                    const double* __restrict__ x0, const double* __restrict__ x1,
                    const double* __restrict__ avgx, const double* __restrict__ y0,
                    const double* __restrict__ y1, const double* __restrict__ avgy,
                    const double* __restrict__ dr, const double* __restrict__ lo_u,
                    const double* __restrict__ hi_u, double eps_rel, double* __restrict__ linf_rows,
                    double* __restrict__ ax_out, double* __restrict__ part);
template __global__ void k_panel_eval_primal<false>(PanelView P, const pdlpdev_ctl* __restrict__ ctl, int which,
                    const double* __restrict__ x0, const double* __restrict__ x1,
                    const double* __restrict__ avgx, const double* __restrict__ y0,
                    const double* __restrict__ y1, const double* __restrict__ avgy,
                    const double* __restrict__ dr, const double* __restrict__ lo_u,
                    const double* __restrict__ hi_u, double eps_rel, double* __restrict__ linf_rows,
                    double* __restrict__ ax_out, double* __restrict__ part);
template __global__ void k_panel_eval_dual<true>(PanelView P, const pdlpdev_ctl* __restrict__ ctl, int which,
                  const double* __restrict__ x0, const double* __restrict__ x1,
                  const double* __restrict__ avgx, const double* __restrict__ y0,
                  const double* __restrict__ y1, const double* __restrict__ avgy, EvalDualCore core,
                  double* __restrict__ part);
template __global__ void k_panel_eval_dual<false>(PanelView P, const pdlpdev_ctl* __restrict__ ctl, int which,
                  const double* __restrict__ x0, const double* __restrict__ x1,
                  const double* __restrict__ avgx, const double* __restrict__ y0,
                  const double* __restrict__ y1, const double* __restrict__ avgy, EvalDualCore core,
                  double* __restrict__ part);
