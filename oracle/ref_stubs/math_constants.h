/* oracle/ref_stubs/math_constants.h -- checker build only: the two CUDA constants adcensus.cu uses
 * (doubles in CUDA's header too; narrowed to float at the use sites). */
#ifndef MCREF_MATH_CONSTANTS_H
#define MCREF_MATH_CONSTANTS_H
#define CUDART_INF (__builtin_inf())
#define CUDART_NAN (__builtin_nan(""))
#endif
