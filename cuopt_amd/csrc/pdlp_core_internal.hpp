// What the core's translation units (pdlp_device.hip, pdlp_eval.hip) share beyond the public device ABI: read-backs, the plain products'
// launch sites, and the element-wise kernels both of them enqueue.  Defined in pdlp_device.hip.
#pragma once
#include "pdlp_ctx.hpp"

#define LAUNCH_CHECK() HIP_TRY(hipGetLastError())

extern "C" {
int fetch_scalars(pdlpdev_ctx* ctx, int count);         // scal[0..count) -> scal_h, synchronised
int fetch_ctl(pdlpdev_ctx* ctx, pdlpdev_ctl* out);      // the control block -> ctl_h (and *out)
void launch_plain(pdlpdev_ctx* ctx, int transpose, const double* vec, double* out);   // out = A vec / A^T vec in the side's layout
void launch_at_cur(pdlpdev_ctx* ctx, double* out_override, int use_next);             // A^T y of the current (next) iterate
void dense_part(pdlpdev_ctx* ctx, int transpose, const double* v0, const double* v1, int mode, int in_loop);  // the dense segments' share
}

// partial sums one product's epilogue leaves, by the side's layout
static inline int dual_partials(const pdlpdev_ctx* ctx) { return ctx->pba.on ? ctx->pba.v.B : ctx->ja.on ? ctx->ja.v.nblk + ctx->ja.v.nlong : ctx->pa.on ? ctx->pa.v.W : ctx->a_nb; }
static inline int step_partials(const pdlpdev_ctx* ctx) { return ctx->pbat.on ? ctx->pbat.v.B : ctx->jat.on ? ctx->jat.v.nblk + ctx->jat.v.nlong : ctx->pat.on ? ctx->pat.v.W : ctx->at_nb; }

__global__ void __launch_bounds__(kBlock)
k_flush_average(int n, int m, const pdlpdev_ctl* __restrict__ ctl, const double* __restrict__ x0, const double* __restrict__ x1, const double* __restrict__ y0,
                const double* __restrict__ y1, double* __restrict__ sumx, double* __restrict__ sumy);
__global__ void __launch_bounds__(kBlock) k_finalize(const double* __restrict__ part, int nb, int nq, unsigned op_mask, double* __restrict__ out);
__global__ void __launch_bounds__(kBlock) k_div_inplace(int n, double* __restrict__ v, const double* __restrict__ d);
__global__ void __launch_bounds__(kBlock) k_div_to(int n, double* __restrict__ out, const double* __restrict__ v, const double* __restrict__ d);
__global__ void k_restart_ctl(pdlpdev_ctl* ctl);
__global__ void k_clear_pending(pdlpdev_ctl* ctl);

int sync_panel_values(pdlpdev_ctx* c);  // layout value arrays <- the (scaled) CSR values (pdlp_device.hip)
__global__ void __launch_bounds__(kBlock) k_fill(int64_t n, double* __restrict__ d, double v);
__global__ void __launch_bounds__(kBlock) k_scale_vectors(int n, int m, double* __restrict__ c, double* __restrict__ lb, double* __restrict__ ub,
                                                          const double* __restrict__ dc, double* __restrict__ lo, double* __restrict__ hi,
                                                          const double* __restrict__ dr);
// (pdlp_scaling.hip; the owner-computes set-up scales its column block with them too)
__global__ void __launch_bounds__(kBlock) k_scale_matrix(int rows, const int32_t* __restrict__ off, const int32_t* __restrict__ idx, double* __restrict__ val,
                                                         const double* __restrict__ d_self, const double* __restrict__ d_other);
__global__ void __launch_bounds__(kBlock) k_scale_matrix_long(const int32_t* __restrict__ rows_long, const int32_t* __restrict__ off,
                                                              const int32_t* __restrict__ idx, double* __restrict__ val,
                                                              const double* __restrict__ d_self, const double* __restrict__ d_other);
