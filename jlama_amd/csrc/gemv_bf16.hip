// gemv_bf16.hip -- explicit instantiations of GEMV launchers and, through them, of their kernels (the other files only declare them: jh_launch.h).
#define JH_LAUNCH_INSTANTIATE 1
#include "jh_launch.h"

template int launch_gemv_bf16<PROB_QUANT_BF16, EPI_RESID, false>(const GemvParams&, int, int*, hipStream_t);
template int launch_gemv_bf16<PROB_QUANT_BF16, EPI_STORE, false>(const GemvParams&, int, int*, hipStream_t);
template int launch_gemv_bf16<PROB_RMS_BF16, EPI_SILU_MUL, false>(const GemvParams&, int, int*, hipStream_t);
template int launch_gemv_bf16<PROB_RMS_BF16, EPI_STORE, false>(const GemvParams&, int, int*, hipStream_t);
template int launch_gemv_bf16<PROB_RMS_F32, EPI_STORE, true>(const GemvParams&, int, int*, hipStream_t);
template int launch_gemv_bf16r<PROB_QUANT_BF16, EPI_RESID, false>(const GemvParams&, int*, hipStream_t);
template int launch_gemv_bf16r<PROB_QUANT_BF16, EPI_STORE, false>(const GemvParams&, int*, hipStream_t);
template int launch_gemv_bf16r<PROB_RMS_BF16, EPI_SILU_MUL, false>(const GemvParams&, int*, hipStream_t);
template int launch_gemv_bf16r<PROB_RMS_BF16, EPI_STORE, false>(const GemvParams&, int*, hipStream_t);
template int launch_gemv_bf16r<PROB_RMS_F32, EPI_STORE, true>(const GemvParams&, int*, hipStream_t);
