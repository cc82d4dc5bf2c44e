// pixel.h -- pixel types and the arithmetic they are filtered in.
//
// The reference evaluates every scan in the pixel type P, with the float coefficients cast
// to P (lib/recfilter.cpp:324,335,338; lib/split.cpp:836,977,1107).  Floating pixels use
// their own type; integer pixels use unsigned 32-bit wrap-around arithmetic, which is the
// same ring the reference's int16/int32 expressions live in once the store truncates.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace rf {

template <typename P> struct PixelTraits;

template <> struct PixelTraits<float> {
    using Acc = float;
    static __host__ __device__ inline Acc load(float v) { return v; }
    static __host__ __device__ inline float store(Acc v) { return v; }
    static inline Acc coef_from_double(double c) { return (float)c; }
    static constexpr bool is_integer = false;
};
template <> struct PixelTraits<double> {
    using Acc = double;
    static __host__ __device__ inline Acc load(double v) { return v; }
    static __host__ __device__ inline double store(Acc v) { return v; }
    static inline Acc coef_from_double(double c) { return c; }
    static constexpr bool is_integer = false;
};
template <> struct PixelTraits<int32_t> {
    using Acc = uint32_t;
    static __host__ __device__ inline Acc load(int32_t v) { return (uint32_t)v; }
    static __host__ __device__ inline int32_t store(Acc v) { return (int32_t)v; }
    static constexpr bool is_integer = true;
};
template <> struct PixelTraits<int16_t> {
    using Acc = uint32_t;
    static __host__ __device__ inline Acc load(int16_t v) { return (uint32_t)(int32_t)v; }
    static __host__ __device__ inline int16_t store(Acc v) { return (int16_t)(uint16_t)v; }
    static constexpr bool is_integer = true;
};

}  // namespace rf
