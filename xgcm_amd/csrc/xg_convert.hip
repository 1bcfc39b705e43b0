// xg_convert.hip -- element type conversion between the storage dtype of an array and the dtype the kernels compute in
// Part of libxgcm_hip.so; compiled ONCE (type-independent; the float64 pass of the build).
//
// numpy keeps integer arrays integral through diff / min / max / cumsum / pad (xgcm/gridops.py:23-24,123-126,172-175,
// 227-278; xgcm/padding.py:610-615: `np.pad` keeps the dtype) and wraps modulo 2^bits.  The integer kernels (*_i64) run on
// two's-complement int64 lanes; this unit widens the narrower / unsigned / bool storage types on the way in and narrows on
// the way out, and it is the `int -> float64` promotion numpy performs before `int * metric` (xgcm/grid.py:804-808,830-832,
// 1600).  One flat kernel: 4 consecutive elements per thread, thread id == memory order.

#include "xg_common.hpp"

#ifdef XG_PRIMARY

namespace {

template <int T> struct TypeOf;
template <> struct TypeOf<XG_T_BOOL> { typedef uint8_t type; };
template <> struct TypeOf<XG_T_I8> { typedef int8_t type; };
template <> struct TypeOf<XG_T_I16> { typedef int16_t type; };
template <> struct TypeOf<XG_T_I32> { typedef int32_t type; };
template <> struct TypeOf<XG_T_I64> { typedef int64_t type; };
template <> struct TypeOf<XG_T_U8> { typedef uint8_t type; };
template <> struct TypeOf<XG_T_U16> { typedef uint16_t type; };
template <> struct TypeOf<XG_T_U32> { typedef uint32_t type; };
template <> struct TypeOf<XG_T_U64> { typedef uint64_t type; };
template <> struct TypeOf<XG_T_F32> { typedef float type; };
template <> struct TypeOf<XG_T_F64> { typedef double type; };
template <> struct TypeOf<XG_T_F16> { typedef _Float16 type; };  // storage only: float16 arrays compute on float32 lanes

constexpr bool is_float_type(int t) { return t == XG_T_F32 || t == XG_T_F64 || t == XG_T_F16; }
constexpr bool is_unsigned_type(int t) { return t == XG_T_BOOL || (t >= XG_T_U8 && t <= XG_T_U64); }
inline int type_bytes(int t) {
  switch (t) {
    case XG_T_BOOL: case XG_T_I8: case XG_T_U8: return 1;
    case XG_T_I16: case XG_T_U16: case XG_T_F16: return 2;
    case XG_T_I32: case XG_T_U32: case XG_T_F32: return 4;
    default: return 8;
  }
}

// the 64-bit pattern `w` as the integer type `via` would hold it (wrap modulo 2^bits, then sign- or zero-extend)
__device__ __forceinline__ int64_t wrap_as(int64_t w, int via) {
  switch (via) {
    case XG_T_BOOL: return w != 0;
    case XG_T_I8: return (int64_t)(int8_t)w;
    case XG_T_I16: return (int64_t)(int16_t)w;
    case XG_T_I32: return (int64_t)(int32_t)w;
    case XG_T_U8: return (int64_t)(uint8_t)w;
    case XG_T_U16: return (int64_t)(uint16_t)w;
    case XG_T_U32: return (int64_t)(uint32_t)w;
    default: return w;  // 64-bit types, -1
  }
}

// one element; `logical` = the integer type whose VALUE the 64-bit pattern stands for (signedness of int -> float)
template <int S, int D>
__device__ __forceinline__ typename TypeOf<D>::type convert1(typename TypeOf<S>::type x, int via, int logical, double scale,
                                                             int flip) {
  typedef typename TypeOf<D>::type DT;
  if constexpr (is_float_type(S)) {
    // float32 / float16 -> float16 straight from the float (v_cvt_f16_f32: one rounding, numpy's astype)
    if constexpr (D == XG_T_F16 && S != XG_T_F64) return (DT)((DT)(float)x * (DT)scale);
    const double f = (double)x;
    if constexpr (is_float_type(D)) return (DT)((DT)f * (DT)scale);
    else if constexpr (D == XG_T_BOOL) return (DT)(f != 0.0);
    else {
      const int64_t w = (D == XG_T_U64) ? (int64_t)(uint64_t)f : (int64_t)f;  // C truncation toward zero, like astype
      return (DT)w;
    }
  } else {
    int64_t w = (S == XG_T_BOOL) ? (int64_t)(x != 0) : (int64_t)x;  // sign- / zero-extension of the storage type
    w = wrap_as(w, via);
    if constexpr (is_float_type(D)) {
      const DT f = (logical == XG_T_U64) ? (DT)(uint64_t)w : (DT)w;  // round to nearest, like numpy's astype
      return (DT)(f * (DT)scale);
    } else if constexpr (D == XG_T_BOOL) {
      return (DT)(w != 0);
    } else {
      if (flip) w ^= (int64_t)0x8000000000000000ull;
      return (DT)w;  // truncation to the destination width = wrap modulo 2^bits
    }
  }
}

constexpr int CV = 4;  // elements per thread

template <int S, int D>
__global__ __launch_bounds__(BLOCK) void k_convert(const typename TypeOf<S>::type* __restrict__ src,
                                                   typename TypeOf<D>::type* __restrict__ dst, u64 n, int via, int logical,
                                                   double scale, int flip, int vec) {
  typedef typename TypeOf<S>::type ST;
  typedef typename TypeOf<D>::type DT;
  typedef ST sv __attribute__((ext_vector_type(CV)));
  typedef DT dvv __attribute__((ext_vector_type(CV)));
  const u64 i0 = ((u64)blockIdx.x * BLOCK + threadIdx.x) * CV;
  if (i0 >= n) return;
  if (vec && i0 + CV <= n) {
    const sv x = *reinterpret_cast<const sv*>(src + i0);
    dvv y;
#pragma unroll
    for (int k = 0; k < CV; ++k) y[k] = convert1<S, D>(x[k], via, logical, scale, flip);
    *reinterpret_cast<dvv*>(dst + i0) = y;
  } else {
    for (int k = 0; k < CV && i0 + k < n; ++k) dst[i0 + k] = convert1<S, D>(src[i0 + k], via, logical, scale, flip);
  }
}

template <int S, int D>
int launch(const void* src, void* dst, u64 n, int via, int logical, double scale, int flip, hipStream_t st) {
  typedef typename TypeOf<S>::type ST;
  typedef typename TypeOf<D>::type DT;
  const int vec = ((reinterpret_cast<uintptr_t>(src) % (sizeof(ST) * CV)) == 0 &&
                   (reinterpret_cast<uintptr_t>(dst) % (sizeof(DT) * CV)) == 0) ? 1 : 0;
  const u64 per = (u64)BLOCK * CV;
  const u64 chunk = (u64)0x7fffff00ull * per;  // elements per launch (grid limit)
  for (u64 off = 0; off < n; off += chunk) {
    const u64 m = (n - off < chunk) ? n - off : chunk;
    const u32 nblk = (u32)((m + per - 1) / per);
    hipLaunchKernelGGL((k_convert<S, D>), dim3(nblk), dim3(BLOCK), 0, st, reinterpret_cast<const ST*>(src) + off,
                       reinterpret_cast<DT*>(dst) + off, m, via, logical, scale, flip, vec);
  }
  return 0;
}

template <int S>
int launch_dst(int D, const void* src, void* dst, u64 n, int via, int logical, double scale, int flip, hipStream_t st) {
  switch (D) {
    case XG_T_BOOL: return launch<S, XG_T_BOOL>(src, dst, n, via, logical, scale, flip, st);
    case XG_T_I8: return launch<S, XG_T_I8>(src, dst, n, via, logical, scale, flip, st);
    case XG_T_I16: return launch<S, XG_T_I16>(src, dst, n, via, logical, scale, flip, st);
    case XG_T_I32: return launch<S, XG_T_I32>(src, dst, n, via, logical, scale, flip, st);
    case XG_T_I64: return launch<S, XG_T_I64>(src, dst, n, via, logical, scale, flip, st);
    case XG_T_U8: return launch<S, XG_T_U8>(src, dst, n, via, logical, scale, flip, st);
    case XG_T_U16: return launch<S, XG_T_U16>(src, dst, n, via, logical, scale, flip, st);
    case XG_T_U32: return launch<S, XG_T_U32>(src, dst, n, via, logical, scale, flip, st);
    case XG_T_U64: return launch<S, XG_T_U64>(src, dst, n, via, logical, scale, flip, st);
    case XG_T_F32: return launch<S, XG_T_F32>(src, dst, n, via, logical, scale, flip, st);
    case XG_T_F16: return launch<S, XG_T_F16>(src, dst, n, via, logical, scale, flip, st);
    default: return launch<S, XG_T_F64>(src, dst, n, via, logical, scale, flip, st);
  }
}

}  // namespace

extern "C" int xg_convert(const void* src, int src_type, void* dst, int dst_type, uint64_t n, int via_type, double scale,
                          int flags, void* stream) {
  if (src_type < XG_T_BOOL || src_type > XG_T_F16 || dst_type < XG_T_BOOL || dst_type > XG_T_F16)
    return fail(XG_ERR_INVALID, "unknown element type (%d -> %d)", src_type, dst_type);
  if (via_type < -1 || via_type > XG_T_U64) return fail(XG_ERR_INVALID, "via_type %d is not an integer type", via_type);
  if (flags & ~1) return fail(XG_ERR_INVALID, "unknown flags %d", flags);
  if (n == 0) return XG_OK;
  if (!src || !dst) return fail(XG_ERR_INVALID, "NULL array argument");
  if (src == dst && type_bytes(src_type) != type_bytes(dst_type)) return fail(XG_ERR_INVALID, "in-place conversion needs equal element sizes");
  if (is_float_type(src_type) && via_type != -1) return fail(XG_ERR_INVALID, "via_type applies to integer sources only");
  if ((flags & 1) && (type_bytes(src_type) != 8 || type_bytes(dst_type) != 8 || is_float_type(src_type) || is_float_type(dst_type)))
    return fail(XG_ERR_INVALID, "the sign-bit flip maps uint64 order to int64 order: 64-bit integer types on both sides");
  // the integer type whose value the (wrapped) 64-bit pattern stands for when it becomes a float
  const int logical = (via_type != -1) ? via_type : src_type;
  hipStream_t st = (hipStream_t)stream;
  const int flip = flags & 1;
  switch (src_type) {
    case XG_T_BOOL: launch_dst<XG_T_BOOL>(dst_type, src, dst, n, via_type, logical, scale, flip, st); break;
    case XG_T_I8: launch_dst<XG_T_I8>(dst_type, src, dst, n, via_type, logical, scale, flip, st); break;
    case XG_T_I16: launch_dst<XG_T_I16>(dst_type, src, dst, n, via_type, logical, scale, flip, st); break;
    case XG_T_I32: launch_dst<XG_T_I32>(dst_type, src, dst, n, via_type, logical, scale, flip, st); break;
    case XG_T_I64: launch_dst<XG_T_I64>(dst_type, src, dst, n, via_type, logical, scale, flip, st); break;
    case XG_T_U8: launch_dst<XG_T_U8>(dst_type, src, dst, n, via_type, logical, scale, flip, st); break;
    case XG_T_U16: launch_dst<XG_T_U16>(dst_type, src, dst, n, via_type, logical, scale, flip, st); break;
    case XG_T_U32: launch_dst<XG_T_U32>(dst_type, src, dst, n, via_type, logical, scale, flip, st); break;
    case XG_T_U64: launch_dst<XG_T_U64>(dst_type, src, dst, n, via_type, logical, scale, flip, st); break;
    case XG_T_F32: launch_dst<XG_T_F32>(dst_type, src, dst, n, via_type, logical, scale, flip, st); break;
    case XG_T_F16: launch_dst<XG_T_F16>(dst_type, src, dst, n, via_type, logical, scale, flip, st); break;
    default: launch_dst<XG_T_F64>(dst_type, src, dst, n, via_type, logical, scale, flip, st); break;
  }
  XG_LAUNCH_CHECK();
  return XG_OK;
}

#endif  // XG_PRIMARY
