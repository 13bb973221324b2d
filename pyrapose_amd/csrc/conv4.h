// Host-side launchers of conv4.hip (one build per plane format, like conv3.hip), called from conv3.hip's dispatch.
#pragma once
#include "conv_common.h"

// the persistent LDS-DMA GEMM of the 1x1 convolutions (igemm4p_kernel): p as dispatch3 fills it; splits > 1 writes partial sums to
// `ws` (the caller runs splitk_finish_kernel); nst = stages of the ring (3 or 4)
void pp4_launch_igemm4p_fmt0(hipStream_t st, IgemmParams& p, const void* ahi, const void* whi, const void* wlo, int w_rows, int w_ld8, void* ohi,
                             void* olo, int splits, float* ws, int n_cu, int nst);
void pp4_launch_igemm4p_fmt1(hipStream_t st, IgemmParams& p, const void* ahi, const void* whi, const void* wlo, int w_rows, int w_ld8, void* ohi,
                             void* olo, int splits, float* ws, int n_cu, int nst);

// the tap-row-reuse weight gradient of 3x3 stride-1 "same" convolutions on plane-stored operands (wgrad3r_kernel): fills the launch
// fields of p and launches, or returns false when the launch does not meet the kernel's conditions
bool pp4_launch_wgrad3r_fmt0(hipStream_t st, Wgrad3Params& p, const void* xhi, const void* xlo, const void* dhi, const void* dlo, float* dw,
                             float* dbias, const int* list, int n_cu, int variant);
bool pp4_launch_wgrad3r_fmt1(hipStream_t st, Wgrad3Params& p, const void* xhi, const void* xlo, const void* dhi, const void* dlo, float* dw,
                             float* dbias, const int* list, int n_cu, int variant);
