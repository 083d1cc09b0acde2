/*
 * ghr_oracle64.c -- TEST INFRASTRUCTURE ONLY: the compositing walk of ghr_oracle.c (K7 forward.cu:287-400, K8
 * backward.cu:403-561) evaluated in IEEE DOUBLE -- the arbiter of tests/test_gpu_fused_fullsize.py's loss leg.
 * Two fp32 implementations of a badly conditioned chain (T <- T / (1 - alpha) at the 0.99 clamp) may each be 1e-3 away
 * from the other and both be right; "which one is closer to the real numbers" is a measurement only with a reference of
 * higher precision.  This file is ghr_oracle.c itself, compiled with float -> double (same statements, same order, same
 * decisions: the callers hand it pixels whose discrete decisions are not within 2e-5 of a threshold in fp32, so the
 * double walk takes the same ones -- and they check n_contrib to be sure).  Only ghro64_render_forward /
 * ghro64_render_backward are meaningful: the projection / binning functions of the file reinterpret float bits and are
 * not exported for use.
 */
#define float double
#define expf exp
#define fminf fmin
#define fmaxf fmax
#define fabsf fabs
#define sqrtf sqrt
#define ceilf ceil
#define floorf floor
#define ghro_preprocess ghro64_unused_preprocess
#define ghro_mark_visible ghro64_unused_mark_visible
#define ghro_scan ghro64_unused_scan
#define ghro_binning ghro64_unused_binning
#define ghro_render_forward ghro64_render_forward
#define ghro_render_backward ghro64_render_backward
#define ghro_cov2d_backward ghro64_unused_cov2d_backward
#define ghro_preprocess_backward ghro64_unused_preprocess_backward
#define ghro_num_threads ghro64_num_threads
#include "ghr_oracle.c"
