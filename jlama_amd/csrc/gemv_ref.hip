// gemv_ref.hip -- explicit instantiations of GEMV launchers and, through them, of their kernels (the other files only declare them: jh_launch.h).
#define JH_LAUNCH_INSTANTIATE 1
#include "jh_launch.h"

template int launch_gemv_i8q4_p16<PRO_QUANT_Q8, EPI_RESID>(const GemvParams&, int, hipStream_t);
template int launch_gemv_i8q4_p16<PRO_QUANT_Q8, EPI_STORE>(const GemvParams&, int, hipStream_t);
template int launch_gemv_i8q4_p16<PRO_QUANT_Q8, EPI_TP>(const GemvParams&, int, hipStream_t);
template int launch_gemv_i8q4_p16<PRO_RMS_Q8, EPI_SILU_MUL>(const GemvParams&, int, hipStream_t);
template int launch_gemv_i8q4_p16<PRO_RMS_Q8, EPI_STORE>(const GemvParams&, int, hipStream_t);
template int launch_gemv_f32q4_p16<PRO_RMS_F32>(const GemvParams&, int*, hipStream_t);
template int launch_gemv_t16<PRO_RMS_Q8, EPI_SILU_MUL>(const GemvParams&, hipStream_t);
