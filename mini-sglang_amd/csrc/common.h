// Shared device/host helpers for the gfx950 kernels.  wave = 64 lanes, hard-coded.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/msgl_hip.h"

namespace msgl {

constexpr int kWave = 64;

// ---- host-side error plumbing -------------------------------------------------
void set_error(const char* fmt, ...);  // defined in host.cpp

#define MSGL_REQUIRE(cond, ...)       \
  do {                                \
    if (!(cond)) {                    \
      ::msgl::set_error(__VA_ARGS__); \
      return MSGL_EINVAL;             \
    }                                 \
  } while (0)

#define MSGL_CHECK_LAUNCH(name)                                          \
  do {                                                                   \
    hipError_t e_ = hipGetLastError();                                   \
    if (e_ != hipSuccess) {                                              \
      ::msgl::set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
      return MSGL_ELAUNCH;                                               \
    }                                                                    \
  } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

int device_cu_count();  // cached, host.cpp

// A kernel that needs more than 64 KB of dynamic LDS has to ask for it with hipFuncSetAttribute, and the attribute belongs
// to the function object of the CURRENT device: one flag per (call site, device), safe under concurrent callers.
struct PerDeviceOnce {
  unsigned done = 0;  // bit d = the attribute is set on device d (devices above 31 ask every time)
  template <typename F>
  bool ensure(F&& set_attribute) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    const unsigned bit = dev >= 0 && dev < 32 ? 1u << dev : 0u;
    if (bit && (__atomic_load_n(&done, __ATOMIC_ACQUIRE) & bit)) return true;
    if (!set_attribute()) return false;
    if (bit) __atomic_fetch_or(&done, bit, __ATOMIC_RELEASE);
    return true;
  }
};

// ---- 16-bit float element traits ---------------------------------------------
struct BF16 {};  // storage: uint16_t, upper half of an fp32
struct FP16 {};  // storage: IEEE half

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_v;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_v;

template <typename T>
struct Elem;

template <>
struct Elem<BF16> {
  // unpack the two 16-bit halves of a dword to fp32 (exact)
  static __device__ __forceinline__ float lo(uint32_t u) { return __uint_as_float(u << 16); }
  static __device__ __forceinline__ float hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
  // round-to-nearest-even fp32 -> bf16 bits (NaN kept quiet)
  static __device__ __forceinline__ uint32_t bits(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
  }
  // two fp32 -> packed bf16 in ONE instruction (v_cvt_pk_bf16_f32: round-to-nearest-even, NaN kept quiet), the same
  // rounding as bits() above; the integer form costs ~6 VALU operations per element
  static __device__ __forceinline__ uint32_t pack(float a, float b) {
    const bf16x2_v v = {static_cast<__bf16>(a), static_cast<__bf16>(b)};
    return __builtin_bit_cast(uint32_t, v);
  }
  // acc += a.lo*b.lo + a.hi*b.hi  (fp32 accumulate, v_dot2c_f32_bf16)
  static __device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float acc) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_v, a),
                                           __builtin_bit_cast(bf16x2_v, b), acc, false);
  }
  // a.lo*b.lo + a.hi*b.hi with an inline-constant 0 accumulator (the VOP3P form: the accumulate-in-place
  // v_dot2c the compiler prefers needs a register zeroed by an extra v_mov first)
  static __device__ __forceinline__ float dot2_first(uint32_t a, uint32_t b) {
    float d;
    asm("v_dot2_f32_bf16 %0, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b));
    return d;
  }
};

template <>
struct Elem<FP16> {
  static __device__ __forceinline__ float lo(uint32_t u) {
    return (float)__builtin_bit_cast(_Float16, (uint16_t)(u & 0xffffu));
  }
  static __device__ __forceinline__ float hi(uint32_t u) {
    return (float)__builtin_bit_cast(_Float16, (uint16_t)(u >> 16));
  }
  static __device__ __forceinline__ uint32_t bits(float f) {
    return (uint32_t)__builtin_bit_cast(uint16_t, (_Float16)f);
  }
  static __device__ __forceinline__ uint32_t pack(float a, float b) { return bits(a) | (bits(b) << 16); }
  static __device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float acc) {
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_v, a), __builtin_bit_cast(f16x2_v, b), acc,
                                  false);
  }
  static __device__ __forceinline__ float dot2_first(uint32_t a, uint32_t b) { return dot2(a, b, 0.f); }
};

// ---- cross-lane helpers (DPP: full-rate, no LDS) -------------------------------
// dpp_ctrl encodings (gfx9): quad_perm = 0x00..0xff, row_shr:n = 0x110+n,
// row_ror:n = 0x120+n, row_mirror = 0x140, row_half_mirror = 0x141.
template <int kCtrl>
__device__ __forceinline__ float dpp_mov(float x) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), kCtrl, 0xf, 0xf, false));
}
// same, but with an undefined "old" operand: every lane is written (all rows / banks enabled and each
// control used here has a source lane for every lane), so no zero has to be materialised for it
template <int kCtrl>
__device__ __forceinline__ float dpp_get(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), kCtrl, 0xf, 0xf, true));
}
constexpr int kDppXor1 = 0xB1;         // quad_perm [1,0,3,2]
constexpr int kDppXor2 = 0x4E;         // quad_perm [2,3,0,1]
constexpr int kDppHalfMirror = 0x141;  // i -> 7-i  within each 8 lanes
constexpr int kDppRowMirror = 0x140;   // i -> 15-i within each 16 lanes
constexpr int kDppRor8 = 0x128;        // row_ror:8: i -> i^8 within each 16 lanes

// all-reduce (sum) over each aligned group of 8 / 16 lanes
__device__ __forceinline__ float row8_sum(float x) {
  x += dpp_mov<kDppXor1>(x);
  x += dpp_mov<kDppXor2>(x);
  x += dpp_mov<kDppHalfMirror>(x);
  return x;
}
__device__ __forceinline__ float row16_sum(float x) {
  x = row8_sum(x);
  x += dpp_mov<kDppRowMirror>(x);
  return x;
}
__device__ __forceinline__ float wave_sum(float x) {
  x = row16_sum(x);
  x += __shfl_xor(x, 16, 64);
  x += __shfl_xor(x, 32, 64);
  return x;
}
__device__ __forceinline__ float wave_max(float x) {
  x = fmaxf(x, dpp_mov<kDppXor1>(x));
  x = fmaxf(x, dpp_mov<kDppXor2>(x));
  x = fmaxf(x, dpp_mov<kDppHalfMirror>(x));
  x = fmaxf(x, dpp_mov<kDppRowMirror>(x));
  x = fmaxf(x, __shfl_xor(x, 16, 64));
  x = fmaxf(x, __shfl_xor(x, 32, 64));
  return x;
}

// silu(g) * u in fp32: the one expression shared by the activation kernels and the fused projection epilogue
// (csrc/gemm_g3.hip) so that they agree bit for bit.  The empty asm pins the fp32 product in a register: without it
// hipcc fuses "multiply, then round to fp16" into one v_fma_mixlo_f16 where the second factor came from an fp16 value
// (single rounding) but not elsewhere (fp32 product, then rounded again): 1-ulp differences between call sites.
__device__ __forceinline__ float silu_mul_f32(float g, float u) {
  float p = (g / (1.0f + __expf(-g))) * u;
  asm("" : "+v"(p));
  return p;
}

__device__ __forceinline__ int sgpr(int x) { return __builtin_amdgcn_readfirstlane(x); }

// 16-byte vector used for every wide load/store
struct __attribute__((aligned(16))) U4 {
  uint32_t x, y, z, w;
};

__device__ __forceinline__ U4 ldg16(const void* p) { return *reinterpret_cast<const U4*>(p); }
__device__ __forceinline__ void stg16(void* p, const U4& v) { *reinterpret_cast<U4*>(p) = v; }

// eight 16-bit elements <-> fp32 (exact / round-to-nearest-even): the one pair every row kernel and every fused
// prologue uses, so that a fused form rounds where the kernel it replaces rounded
template <typename T>
__device__ __forceinline__ void unpack8(const U4& u, float (&f)[8]) {
  f[0] = Elem<T>::lo(u.x); f[1] = Elem<T>::hi(u.x);
  f[2] = Elem<T>::lo(u.y); f[3] = Elem<T>::hi(u.y);
  f[4] = Elem<T>::lo(u.z); f[5] = Elem<T>::hi(u.z);
  f[6] = Elem<T>::lo(u.w); f[7] = Elem<T>::hi(u.w);
}
template <typename T>
__device__ __forceinline__ U4 pack8(const float (&f)[8]) {
  U4 u;
  u.x = Elem<T>::pack(f[0], f[1]); u.y = Elem<T>::pack(f[2], f[3]);
  u.z = Elem<T>::pack(f[4], f[5]); u.w = Elem<T>::pack(f[6], f[7]);
  return u;
}

}  // namespace msgl
