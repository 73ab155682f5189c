/* systolic kernels of one method: see sa_systolic_kernel.inc */
#include "sa_internal.h"
#define SA_SYS_METHOD SA_METHOD_SW
#define SA_SYS_LAUNCH sa_launch_systolic_sw
#define SA_SYS_WARM sa_warm_systolic_sw
#include "sa_systolic_kernel.inc"
