// real.hpp -- the solver's real type.  Included by solver.hip AFTER every system header and BEFORE the ABI and the kernels.
//
// GATO_DOUBLE is the reference's USE_DOUBLES (gato/settings.h:7-11: `typedef double T`, double classes in python/bindings.cu:244-252):
// every `float` that follows this header is the solver's real type, and the same sources compile to libgato_hip_f64.so with the same
// entry points on double buffers -- the float64 build of the oracle is made the same way (oracle/Makefile).  Vector types, math functions
// and lane operations go through the aliases and overloads below so that they follow.  The float64 build is the validation mode it is in
// the reference: it runs the SAME kernel plan as fp32 (fused, pair, symmetric storage ...: solver.hip:plan_pcg has no branch on the real
// type), and since every register budget in kernels.hpp is sized for 4-byte reals its kernels spill -- correct, several times slower.
#pragma once

namespace gato {
typedef float f32_t;   // the 4-byte type, for interfaces that are fp32 whatever the real type is (hipEventElapsedTime)
#define GATO_LANE __device__ __forceinline__
// ---- lane operations on 4-byte and on 8-byte reals (an 8-byte value moves as two dwords) -------------------------------------
// v_mov_b32_dpp: lanes the row / bank masks disable keep `old`
template<int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf> GATO_LANE float dpp_mov(float old, float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, ROW_MASK, BANK_MASK, false));
}
template<int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf> GATO_LANE double dpp_mov(double old, double v)
{
    const unsigned long long o = __builtin_bit_cast(unsigned long long, old), x = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)o, (int)(unsigned)x, CTRL, ROW_MASK, BANK_MASK, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)(o >> 32), (int)(unsigned)(x >> 32), CTRL, ROW_MASK, BANK_MASK, false);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
template<int CTRL, class R> GATO_LANE R dpp_get(R v) { return dpp_mov<CTRL>(R(0), v); }   // every lane enabled
// v_readlane_b32: `lane` wavefront-uniform
GATO_LANE float lane_read(float v, int lane) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane)); }
GATO_LANE double lane_read(double v, int lane)
{
    const unsigned long long x = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)x, lane), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(x >> 32), lane);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// ds_bpermute_b32: the value of lane byte_addr / 4
GATO_LANE float lane_permute(int byte_addr, float v) { return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(byte_addr, __builtin_bit_cast(int, v))); }
GATO_LANE double lane_permute(int byte_addr, double v)
{
    const unsigned long long x = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute(byte_addr, (int)(unsigned)x), hi = (unsigned)__builtin_amdgcn_ds_bpermute(byte_addr, (int)(unsigned)(x >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// v_rcp_f32 (1 ulp); the float64 build divides
GATO_LANE float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
GATO_LANE double fast_rcp(double x) { return 1.0 / x; }
}  // namespace gato

#ifdef GATO_DOUBLE
#define float double
#define sincosf sincos
#define fabsf fabs
#define fmaxf fmax
#define fminf fmin
#define sqrtf sqrt
#define logf log
#define __builtin_fmaf __builtin_fma
#endif

namespace gato {
#ifdef GATO_DOUBLE
typedef float real2 __attribute__((ext_vector_type(2)));
typedef float real4 __attribute__((ext_vector_type(4)));
GATO_LANE real4 make_real4(float a, float b, float c, float d) { return real4{a, b, c, d}; }
GATO_LANE real2 make_real2(float a, float b) { return real2{a, b}; }
#else
// the fp32 build keeps HIP's own vector types: with ext_vector_type aliases the symmetric-storage PCG kernel, which sits at 256
// registers, spilled 340 instead of 36 bytes per lane (C3: 581 vs 416 us per launch)
typedef ::float2 real2;
typedef ::float4 real4;
GATO_LANE real4 make_real4(float a, float b, float c, float d) { return make_float4(a, b, c, d); }
GATO_LANE real2 make_real2(float a, float b) { return make_float2(a, b); }
#endif
constexpr bool kDouble = sizeof(float) == 8;
}  // namespace gato
