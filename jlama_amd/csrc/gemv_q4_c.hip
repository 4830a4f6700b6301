// gemv_q4_c.hip -- explicit instantiations of GEMV launchers and, through them, of their kernels (the other files only declare them: jh_launch.h).
#define JH_LAUNCH_INSTANTIATE 1
#include "jh_launch.h"

template int launch_gemv_i8q4<PRO_RMS_Q8, EPI_SILU_MUL>(const GemvParams&, LaunchCfg, hipStream_t);
template int launch_gemv_i8q4<PRO_RMS_Q8, EPI_STORE>(const GemvParams&, LaunchCfg, hipStream_t);
template int launch_gemv_f32q4<PRO_F32>(const GemvParams&, LaunchCfg, int*, hipStream_t);
template int launch_gemv_f32q4<PRO_RMS_F32>(const GemvParams&, LaunchCfg, int*, hipStream_t);
