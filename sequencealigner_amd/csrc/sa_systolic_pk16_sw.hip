/* packed-u16 systolic kernels of one method, 16-lane groups: see sa_systolic_pk.inc */
#include "sa_internal.h"
#define SA_SYS_METHOD SA_METHOD_SW
#define SA_PK_G 16
#define SA_PK16_PART 0
#define SA_SYS_LAUNCH sa_launch_systolic_pk16_sw
#define SA_SYS_WARM sa_warm_systolic_pk16_sw
#include "sa_systolic_pk.inc"
