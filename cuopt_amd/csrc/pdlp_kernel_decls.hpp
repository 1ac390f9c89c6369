// Prototypes of the SpMV kernels, which are DEFINED one layout per translation unit (kernels_<layout>.hip) and launched from the
// core (pdlp_device.hip); the templated ones are instantiated explicitly where they are defined.  Generated once from the
// definitions when the file was split (round 4); keep the two in step.
#pragma once
#include "pdlp_epilogues.hpp"

__global__ void __launch_bounds__(kBlock)
k_spmv_a_dual(int nb, const int32_t* __restrict__ rb, const int32_t* __restrict__ off,
              const int32_t* __restrict__ idx, const double* __restrict__ val,
              const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ xbar,
              double* __restrict__ y0, double* __restrict__ y1, const double* __restrict__ lo,
              const double* __restrict__ hi, double* __restrict__ sumy, double* __restrict__ part, double* __restrict__ ycopy,
              const p2pdev::Push* __restrict__ push, const double* __restrict__ dadd);
__global__ void __launch_bounds__(kBlock)
k_spmv_at_step(int nb, const int32_t* __restrict__ rb, const int32_t* __restrict__ off,
               const int32_t* __restrict__ idx, const double* __restrict__ val,
               const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ y0,
               const double* __restrict__ y1, const double* __restrict__ x0,
               const double* __restrict__ x1, double* __restrict__ aty0, double* __restrict__ aty1,
               double* __restrict__ part, const double* __restrict__ dadd);
__global__ void __launch_bounds__(kBlock)
k_spmv_plain(int nb, const int32_t* __restrict__ rb, const int32_t* __restrict__ off,
             const int32_t* __restrict__ idx, const double* __restrict__ val,
             const double* __restrict__ vec, double* __restrict__ out, const double* __restrict__ dadd);
__global__ void __launch_bounds__(kBlock)
k_spmv_at_cur(int nb, const int32_t* __restrict__ rb, const int32_t* __restrict__ off,
              const int32_t* __restrict__ idx, const double* __restrict__ val,
              const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ y0,
              const double* __restrict__ y1, double* __restrict__ aty0, double* __restrict__ aty1,
              double* __restrict__ out_override, int use_next, const double* __restrict__ dadd);
__global__ void __launch_bounds__(kBlock)
k_eval_primal(int nb, const int32_t* __restrict__ rb, const int32_t* __restrict__ off,
              const int32_t* __restrict__ idx, const double* __restrict__ val,
              const pdlpdev_ctl* __restrict__ ctl, int which, const double* __restrict__ x0,
              const double* __restrict__ x1, const double* __restrict__ avgx,
              const double* __restrict__ y0, const double* __restrict__ y1,
              const double* __restrict__ avgy, const double* __restrict__ dr,
              const double* __restrict__ lo_u, const double* __restrict__ hi_u, double eps_rel,
              double* __restrict__ linf_rows, double* __restrict__ ax_out, double* __restrict__ part, const double* __restrict__ dadd);
__global__ void __launch_bounds__(kBlock)
k_eval_dual(int nb, const int32_t* __restrict__ rb, const int32_t* __restrict__ off,
            const int32_t* __restrict__ idx, const double* __restrict__ val,
            const pdlpdev_ctl* __restrict__ ctl, int which, const double* __restrict__ x0,
            const double* __restrict__ x1, const double* __restrict__ avgx,
            const double* __restrict__ y0, const double* __restrict__ y1,
            const double* __restrict__ avgy, EvalDualCore core, double* __restrict__ part, const double* __restrict__ dadd);
template <bool SEG>
__global__ void __launch_bounds__(kPanelThreads)
k_panel_a_dual(PanelView P, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ xbar,
               double* __restrict__ y0, double* __restrict__ y1, const double* __restrict__ lo,
               const double* __restrict__ hi, double* __restrict__ sumy, double* __restrict__ part, double* __restrict__ ycopy,
               const p2pdev::Push* __restrict__ push);
extern template __global__ void k_panel_a_dual<true>(PanelView P, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ xbar,
               double* __restrict__ y0, double* __restrict__ y1, const double* __restrict__ lo,
               const double* __restrict__ hi, double* __restrict__ sumy, double* __restrict__ part, double* __restrict__ ycopy,
               const p2pdev::Push* __restrict__ push);
extern template __global__ void k_panel_a_dual<false>(PanelView P, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ xbar,
               double* __restrict__ y0, double* __restrict__ y1, const double* __restrict__ lo,
               const double* __restrict__ hi, double* __restrict__ sumy, double* __restrict__ part, double* __restrict__ ycopy,
               const p2pdev::Push* __restrict__ push);
template <bool SEG>
__global__ void __launch_bounds__(kPanelThreads)
k_panel_at_step(PanelView P, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ y0,
                const double* __restrict__ y1, const double* __restrict__ x0,
                const double* __restrict__ x1, double* __restrict__ aty0, double* __restrict__ aty1,
                double* __restrict__ part);
extern template __global__ void k_panel_at_step<true>(PanelView P, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ y0,
                const double* __restrict__ y1, const double* __restrict__ x0,
                const double* __restrict__ x1, double* __restrict__ aty0, double* __restrict__ aty1,
                double* __restrict__ part);
extern template __global__ void k_panel_at_step<false>(PanelView P, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ y0,
                const double* __restrict__ y1, const double* __restrict__ x0,
                const double* __restrict__ x1, double* __restrict__ aty0, double* __restrict__ aty1,
                double* __restrict__ part);
template <bool SEG>
__global__ void __launch_bounds__(kPanelThreads)
k_panel_at_cur(PanelView P, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ y0,
               const double* __restrict__ y1, double* __restrict__ aty0, double* __restrict__ aty1,
               double* __restrict__ out_override, int use_next);
extern template __global__ void k_panel_at_cur<true>(PanelView P, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ y0,
               const double* __restrict__ y1, double* __restrict__ aty0, double* __restrict__ aty1,
               double* __restrict__ out_override, int use_next);
extern template __global__ void k_panel_at_cur<false>(PanelView P, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ y0,
               const double* __restrict__ y1, double* __restrict__ aty0, double* __restrict__ aty1,
               double* __restrict__ out_override, int use_next);
template <bool SEG>
__global__ void __launch_bounds__(kPanelThreads)
k_panel_plain(PanelView P, const double* __restrict__ vec, double* __restrict__ out);
extern template __global__ void k_panel_plain<true>(PanelView P, const double* __restrict__ vec, double* __restrict__ out);
extern template __global__ void k_panel_plain<false>(PanelView P, const double* __restrict__ vec, double* __restrict__ out);
template <bool SEG>
__global__ void __launch_bounds__(kPanelThreads)
k_panel_eval_primal(PanelView P, const pdlpdev_ctl* __restrict__ ctl, int which,
                    const double* __restrict__ x0, const double* __restrict__ x1,
                    const double* __restrict__ avgx, const double* __restrict__ y0,
                    const double* __restrict__ y1, const double* __restrict__ avgy,
                    const double* __restrict__ dr, const double* __restrict__ lo_u,
                    const double* __restrict__ hi_u, double eps_rel, double* __restrict__ linf_rows,
                    double* __restrict__ ax_out, double* __restrict__ part);
extern template __global__ void k_panel_eval_primal<true>(PanelView P, const pdlpdev_ctl* __restrict__ ctl, int which,
                    const double* __restrict__ x0, const double* __restrict__ x1,
                    const double* __restrict__ avgx, const double* __restrict__ y0,
                    const double* __restrict__ y1, const double* __restrict__ avgy,
                    const double* __restrict__ dr, const double* __restrict__ lo_u,
                    const double* __restrict__ hi_u, double eps_rel, double* __restrict__ linf_rows,
                    double* __restrict__ ax_out, double* __restrict__ part);
extern template __global__ void k_panel_eval_primal<false>(PanelView P, const pdlpdev_ctl* __restrict__ ctl, int which,
                    const double* __restrict__ x0, const double* __restrict__ x1,
                    const double* __restrict__ avgx, const double* __restrict__ y0,
                    const double* __restrict__ y1, const double* __restrict__ avgy,
                    const double* __restrict__ dr, const double* __restrict__ lo_u,
                    const double* __restrict__ hi_u, double eps_rel, double* __restrict__ linf_rows,
                    double* __restrict__ ax_out, double* __restrict__ part);
template <bool SEG>
__global__ void __launch_bounds__(kPanelThreads)
k_panel_eval_dual(PanelView P, const pdlpdev_ctl* __restrict__ ctl, int which,
                  const double* __restrict__ x0, const double* __restrict__ x1,
                  const double* __restrict__ avgx, const double* __restrict__ y0,
                  const double* __restrict__ y1, const double* __restrict__ avgy, EvalDualCore core,
                  double* __restrict__ part);
extern template __global__ void k_panel_eval_dual<true>(PanelView P, const pdlpdev_ctl* __restrict__ ctl, int which,
                  const double* __restrict__ x0, const double* __restrict__ x1,
                  const double* __restrict__ avgx, const double* __restrict__ y0,
                  const double* __restrict__ y1, const double* __restrict__ avgy, EvalDualCore core,
                  double* __restrict__ part);
extern template __global__ void k_panel_eval_dual<false>(PanelView P, const pdlpdev_ctl* __restrict__ ctl, int which,
                  const double* __restrict__ x0, const double* __restrict__ x1,
                  const double* __restrict__ avgx, const double* __restrict__ y0,
                  const double* __restrict__ y1, const double* __restrict__ avgy, EvalDualCore core,
                  double* __restrict__ part);
template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64)
k_jag_a_dual(JagView J, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ xbar,
             double* __restrict__ y0, double* __restrict__ y1, const double* __restrict__ lo,
             const double* __restrict__ hi, double* __restrict__ sumy, double* __restrict__ part, double* __restrict__ ycopy,
             const p2pdev::Push* __restrict__ push);
extern template __global__ void k_jag_a_dual<8>(JagView J, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ xbar,
             double* __restrict__ y0, double* __restrict__ y1, const double* __restrict__ lo,
             const double* __restrict__ hi, double* __restrict__ sumy, double* __restrict__ part, double* __restrict__ ycopy,
             const p2pdev::Push* __restrict__ push);
extern template __global__ void k_jag_a_dual<16>(JagView J, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ xbar,
             double* __restrict__ y0, double* __restrict__ y1, const double* __restrict__ lo,
             const double* __restrict__ hi, double* __restrict__ sumy, double* __restrict__ part, double* __restrict__ ycopy,
             const p2pdev::Push* __restrict__ push);
template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64)
k_jag_at_step(JagView J, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ y0,
              const double* __restrict__ y1, const double* __restrict__ x0,
              const double* __restrict__ x1, double* __restrict__ aty0, double* __restrict__ aty1,
              double* __restrict__ part);
extern template __global__ void k_jag_at_step<8>(JagView J, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ y0,
              const double* __restrict__ y1, const double* __restrict__ x0,
              const double* __restrict__ x1, double* __restrict__ aty0, double* __restrict__ aty1,
              double* __restrict__ part);
extern template __global__ void k_jag_at_step<16>(JagView J, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ y0,
              const double* __restrict__ y1, const double* __restrict__ x0,
              const double* __restrict__ x1, double* __restrict__ aty0, double* __restrict__ aty1,
              double* __restrict__ part);
template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64)
k_jag_at_cur(JagView J, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ y0,
             const double* __restrict__ y1, double* __restrict__ aty0, double* __restrict__ aty1,
             double* __restrict__ out_override, int use_next);
extern template __global__ void k_jag_at_cur<8>(JagView J, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ y0,
             const double* __restrict__ y1, double* __restrict__ aty0, double* __restrict__ aty1,
             double* __restrict__ out_override, int use_next);
extern template __global__ void k_jag_at_cur<16>(JagView J, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ y0,
             const double* __restrict__ y1, double* __restrict__ aty0, double* __restrict__ aty1,
             double* __restrict__ out_override, int use_next);
template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64)
k_jag_plain(JagView J, const double* __restrict__ vec, double* __restrict__ out);
extern template __global__ void k_jag_plain<8>(JagView J, const double* __restrict__ vec, double* __restrict__ out);
extern template __global__ void k_jag_plain<16>(JagView J, const double* __restrict__ vec, double* __restrict__ out);
template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64)
k_jag_eval_primal(JagView J, const pdlpdev_ctl* __restrict__ ctl, int which,
                  const double* __restrict__ x0, const double* __restrict__ x1,
                  const double* __restrict__ avgx, const double* __restrict__ y0,
                  const double* __restrict__ y1, const double* __restrict__ avgy,
                  const double* __restrict__ dr, const double* __restrict__ lo_u,
                  const double* __restrict__ hi_u, double eps_rel, double* __restrict__ linf_rows,
                  double* __restrict__ ax_out, double* __restrict__ part);
extern template __global__ void k_jag_eval_primal<8>(JagView J, const pdlpdev_ctl* __restrict__ ctl, int which,
                  const double* __restrict__ x0, const double* __restrict__ x1,
                  const double* __restrict__ avgx, const double* __restrict__ y0,
                  const double* __restrict__ y1, const double* __restrict__ avgy,
                  const double* __restrict__ dr, const double* __restrict__ lo_u,
                  const double* __restrict__ hi_u, double eps_rel, double* __restrict__ linf_rows,
                  double* __restrict__ ax_out, double* __restrict__ part);
extern template __global__ void k_jag_eval_primal<16>(JagView J, const pdlpdev_ctl* __restrict__ ctl, int which,
                  const double* __restrict__ x0, const double* __restrict__ x1,
                  const double* __restrict__ avgx, const double* __restrict__ y0,
                  const double* __restrict__ y1, const double* __restrict__ avgy,
                  const double* __restrict__ dr, const double* __restrict__ lo_u,
                  const double* __restrict__ hi_u, double eps_rel, double* __restrict__ linf_rows,
                  double* __restrict__ ax_out, double* __restrict__ part);
template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64)
k_jag_eval_dual(JagView J, const pdlpdev_ctl* __restrict__ ctl, int which,
                const double* __restrict__ x0, const double* __restrict__ x1,
                const double* __restrict__ avgx, const double* __restrict__ y0,
                const double* __restrict__ y1, const double* __restrict__ avgy, EvalDualCore core,
                double* __restrict__ part);
extern template __global__ void k_jag_eval_dual<8>(JagView J, const pdlpdev_ctl* __restrict__ ctl, int which,
                const double* __restrict__ x0, const double* __restrict__ x1,
                const double* __restrict__ avgx, const double* __restrict__ y0,
                const double* __restrict__ y1, const double* __restrict__ avgy, EvalDualCore core,
                double* __restrict__ part);
extern template __global__ void k_jag_eval_dual<16>(JagView J, const pdlpdev_ctl* __restrict__ ctl, int which,
                const double* __restrict__ x0, const double* __restrict__ x1,
                const double* __restrict__ avgx, const double* __restrict__ y0,
                const double* __restrict__ y1, const double* __restrict__ avgy, EvalDualCore core,
                double* __restrict__ part);
template <int THREADS>
__global__ void __launch_bounds__(THREADS)
k_pb_products(PbView V, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ v0, const double* __restrict__ v1, int mode, int in_loop);
extern template __global__ void k_pb_products<512>(PbView V, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ v0, const double* __restrict__ v1, int mode, int in_loop);
extern template __global__ void k_pb_products<1024>(PbView V, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ v0, const double* __restrict__ v1, int mode, int in_loop);
// (phase R: WIDE = the wide-bin skeleton, 1024 threads; otherwise the image-in-LDS skeleton, 512)
template <bool WIDE>
__global__ void __launch_bounds__(WIDE ? kPbwThreads : kPbThreads)
k_pb_a_dual(PbView V, const pdlpdev_ctl* __restrict__ ctl, double* __restrict__ y0, double* __restrict__ y1, const double* __restrict__ lo,
            const double* __restrict__ hi, double* __restrict__ sumy, double* __restrict__ part, double* __restrict__ ycopy,
            const p2pdev::Push* __restrict__ push);
extern template __global__ void k_pb_a_dual<false>(PbView V, const pdlpdev_ctl* __restrict__ ctl, double* __restrict__ y0, double* __restrict__ y1, const double* __restrict__ lo,
            const double* __restrict__ hi, double* __restrict__ sumy, double* __restrict__ part, double* __restrict__ ycopy,
            const p2pdev::Push* __restrict__ push);
extern template __global__ void k_pb_a_dual<true>(PbView V, const pdlpdev_ctl* __restrict__ ctl, double* __restrict__ y0, double* __restrict__ y1, const double* __restrict__ lo,
            const double* __restrict__ hi, double* __restrict__ sumy, double* __restrict__ part, double* __restrict__ ycopy,
            const p2pdev::Push* __restrict__ push);
template <bool WIDE>
__global__ void __launch_bounds__(WIDE ? kPbwThreads : kPbThreads)
k_pb_at_step(PbView V, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ x0, const double* __restrict__ x1,
             double* __restrict__ aty0, double* __restrict__ aty1, double* __restrict__ part);
extern template __global__ void k_pb_at_step<false>(PbView V, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ x0, const double* __restrict__ x1,
             double* __restrict__ aty0, double* __restrict__ aty1, double* __restrict__ part);
extern template __global__ void k_pb_at_step<true>(PbView V, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ x0, const double* __restrict__ x1,
             double* __restrict__ aty0, double* __restrict__ aty1, double* __restrict__ part);
template <bool WIDE>
__global__ void __launch_bounds__(WIDE ? kPbwThreads : kPbThreads)
k_pb_at_cur(PbView V, const pdlpdev_ctl* __restrict__ ctl, double* __restrict__ aty0, double* __restrict__ aty1,
            double* __restrict__ out_override, int use_next);
extern template __global__ void k_pb_at_cur<false>(PbView V, const pdlpdev_ctl* __restrict__ ctl, double* __restrict__ aty0, double* __restrict__ aty1,
            double* __restrict__ out_override, int use_next);
extern template __global__ void k_pb_at_cur<true>(PbView V, const pdlpdev_ctl* __restrict__ ctl, double* __restrict__ aty0, double* __restrict__ aty1,
            double* __restrict__ out_override, int use_next);
template <bool WIDE>
__global__ void __launch_bounds__(WIDE ? kPbwThreads : kPbThreads)
k_pb_plain(PbView V, double* __restrict__ out);
extern template __global__ void k_pb_plain<false>(PbView V, double* __restrict__ out);
extern template __global__ void k_pb_plain<true>(PbView V, double* __restrict__ out);
template <bool WIDE>
__global__ void __launch_bounds__(WIDE ? kPbwThreads : kPbThreads)
k_pb_eval_primal(PbView V, const pdlpdev_ctl* __restrict__ ctl, int which, const double* __restrict__ y0, const double* __restrict__ y1,
                 const double* __restrict__ avgy, const double* __restrict__ dr, const double* __restrict__ lo_u,
                 const double* __restrict__ hi_u, double eps_rel, double* __restrict__ linf_rows, double* __restrict__ ax_out,
                 double* __restrict__ part);
extern template __global__ void k_pb_eval_primal<false>(PbView V, const pdlpdev_ctl* __restrict__ ctl, int which, const double* __restrict__ y0, const double* __restrict__ y1,
                 const double* __restrict__ avgy, const double* __restrict__ dr, const double* __restrict__ lo_u,
                 const double* __restrict__ hi_u, double eps_rel, double* __restrict__ linf_rows, double* __restrict__ ax_out,
                 double* __restrict__ part);
extern template __global__ void k_pb_eval_primal<true>(PbView V, const pdlpdev_ctl* __restrict__ ctl, int which, const double* __restrict__ y0, const double* __restrict__ y1,
                 const double* __restrict__ avgy, const double* __restrict__ dr, const double* __restrict__ lo_u,
                 const double* __restrict__ hi_u, double eps_rel, double* __restrict__ linf_rows, double* __restrict__ ax_out,
                 double* __restrict__ part);
template <bool WIDE>
__global__ void __launch_bounds__(WIDE ? kPbwThreads : kPbThreads)
k_pb_eval_dual(PbView V, const pdlpdev_ctl* __restrict__ ctl, int which, const double* __restrict__ x0, const double* __restrict__ x1,
               const double* __restrict__ avgx, EvalDualCore core, double* __restrict__ part);
extern template __global__ void k_pb_eval_dual<false>(PbView V, const pdlpdev_ctl* __restrict__ ctl, int which, const double* __restrict__ x0, const double* __restrict__ x1,
               const double* __restrict__ avgx, EvalDualCore core, double* __restrict__ part);
extern template __global__ void k_pb_eval_dual<true>(PbView V, const pdlpdev_ctl* __restrict__ ctl, int which, const double* __restrict__ x0, const double* __restrict__ x1,
               const double* __restrict__ avgx, EvalDualCore core, double* __restrict__ part);
__global__ void __launch_bounds__(kBlock)
k_dense_rows(DenseView D, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ v0, const double* __restrict__ v1, int mode, int in_loop);
__global__ void __launch_bounds__(kBlock)
k_dense_rows_finish(DenseView D, int nrows, const pdlpdev_ctl* __restrict__ ctl, int in_loop, double* __restrict__ add);
__global__ void __launch_bounds__(kBlock)
k_dense_cols(DenseView D, int n, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ v0, const double* __restrict__ v1, int mode,
             int in_loop, double* __restrict__ add);
