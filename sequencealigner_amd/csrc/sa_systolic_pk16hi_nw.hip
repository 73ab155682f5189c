/* packed-u16 systolic kernels of one method, 16-lane groups, K = 45..64 (columns 705..1024): see sa_systolic_pk.inc */
#include "sa_internal.h"
#define SA_SYS_METHOD SA_METHOD_NW
#define SA_PK_G 16
#define SA_PK16_PART 1
#define SA_SYS_LAUNCH sa_launch_systolic_pk16hi_nw
#define SA_SYS_WARM sa_warm_systolic_pk16hi_nw
#include "sa_systolic_pk.inc"
