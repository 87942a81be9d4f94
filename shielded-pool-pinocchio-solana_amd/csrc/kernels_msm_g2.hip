// The G2 instantiations of the table-walk kernels (kernels_msm.hip), a translation unit of their own so that the G1 walk can be
// compiled with another instruction-scheduling strategy (Makefile).
#define SPP_MSM_TU_G2 1
#include "kernels_msm.hip"
