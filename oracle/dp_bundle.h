/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle for the raytracer_datapath hot path.
 * Nothing under oracle/ is part of the product: only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may load it, as the checker.
 *
 * Plain-C image of the datapath's IO bundles, one record per clock of the RTL.
 * Every FP field is carried as its raw IEEE-754 binary32 bit pattern
 * (Chisel Bits(32.W)) so NaN payloads and signed zeros survive untouched.
 *
 *   dp_in_bundle   <- EnhancedInputBundle   (reference src/main/scala/raytracer_datapath/IOSpec.scala:106-133)
 *                     Ray :54-71, AABB :73-91, Triangle :93-99
 *   dp_out_bundle  <- EnhancedOutputBundle  (IOSpec.scala:135-173)
 *   dp_accum_state <- the three accumulator registers
 *                     (UnifiedDatapath.scala:1031-1032 angular, :1136 euclidean)
 */
#ifndef DP_BUNDLE_H
#define DP_BUNDLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { DP_OP_TRIANGLE = 0, DP_OP_QUADBOX = 1, DP_OP_EUCLIDEAN = 2, DP_OP_ANGULAR = 3 };

typedef struct dp_f3 { uint32_t x, y, z; } dp_f3;

typedef struct dp_ray {
    dp_f3 origin, dir, inv;
    uint32_t extent;
    uint32_t kx, ky, kz; /* 2 bits each */
    dp_f3 shear;
} dp_ray;

typedef struct dp_aabb { uint32_t x_min, y_min, z_min, x_max, y_max, z_max; } dp_aabb;

typedef struct dp_triangle { dp_f3 A, B, C; } dp_triangle;

typedef struct dp_in_bundle {
    uint32_t opcode; /* 2 bits */
    dp_ray ray;
    dp_aabb aabb[4];
    dp_triangle triangle;
    uint32_t euclidean_a[16];
    uint32_t euclidean_b[16];
    uint32_t euclidean_mask;        /* 16 bits, bit i <-> element i */
    uint32_t euclidean_reset_accum; /* 1 bit */
} dp_in_bundle;

typedef struct dp_out_bundle {
    uint32_t opcode;
    uint32_t tmin_out[4];
    uint32_t isIntersect[4];
    uint32_t boxIndex[4];
    uint32_t t_num, t_denom;
    uint32_t triangle_hit;
    uint32_t euclidean_accumulator;
    uint32_t euclidean_reset_accum;
    uint32_t angular_dot_product;
    uint32_t angular_norm;
    uint32_t angular_reset_accum;
} dp_out_bundle;

/* Accumulator registers as IEEE bit patterns; power-up value +0. */
typedef struct dp_accum_state {
    uint32_t euclidean;
    uint32_t angular_dot;
    uint32_t angular_norm;
} dp_accum_state;

#ifdef __cplusplus
}
#endif
#endif
