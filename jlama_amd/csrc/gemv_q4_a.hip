// gemv_q4_a.hip -- explicit instantiations of GEMV launchers and, through them, of their kernels (the other files only declare them: jh_launch.h).
#define JH_LAUNCH_INSTANTIATE 1
#include "jh_launch.h"

template int launch_gemv_i8q4<PRO_Q8, EPI_RESID>(const GemvParams&, LaunchCfg, hipStream_t);
template int launch_gemv_i8q4<PRO_Q8, EPI_SILU_MUL>(const GemvParams&, LaunchCfg, hipStream_t);
template int launch_gemv_i8q4<PRO_Q8, EPI_STORE>(const GemvParams&, LaunchCfg, hipStream_t);
