// Shared device/host helpers for the X3D HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/x3d_hip.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

// Kernel-selection A/B switches (X3D_* environment variables): every one of them belongs to an experiment whose outcome
// is recorded next to the call site and in DESIGN.md, so the PRODUCT library reads no environment at all -- x3d_env_int
// is its default there.  A build made with -DX3D_EXPERIMENTS (X3D_EXPERIMENTS=1 python -m x3d-tf_amd.build; tools/ab_*)
// reads the variable on every call, which is what the in-process A/B tools need.
static inline int x3d_env_int(const char* name, int dflt) {
#ifdef X3D_EXPERIMENTS
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
#else
  (void)name;
  return dflt;
#endif
}

// 16-bit storage types (X3D_BF16 / X3D_F16): vector types, the matrix-core instruction and the name used in kernel
// descriptions.  Every 16-bit kernel is a template over H; arithmetic around the matrix cores is fp32 either way.
// 16-bit storage, point counts that are not a multiple of 8 (X3D-S: 13 frames -> 1300 / 325 points per sample in stages
// 4 / 5; X3D-XS: 100): rows then start at any 2-byte address and end inside their last 8-point vector.  The vector form of
// the resident-panel kernel (pw_gemm_bf16.h) takes them all the same: unaligned 16-byte accesses for the whole vectors,
// element accesses for a row's last one, sums masked by element (the scalar form it used to fall back to ran the
// 216 -> 96 @ 13x10x10 layer in 311 us).  X3D_PW_RAGGED=0: A/B hook.
static inline bool pw_ragged_rows(long long P, int dtype_bytes) {
  return dtype_bytes == 2 && (P & 7) != 0 && P >= 8 && x3d_env_int("X3D_PW_RAGGED", 1) != 0;
}

// 8 elements of a row of which only the first nv exist (the row -- P % 8 != 0 points -- ends inside the vector): element
// accesses, zero fill.  The whole vectors of such rows start at any 2-byte address; the compute queues run in unaligned
// access mode, so the 16-byte loads / stores stay (checked on the GPU by the odd-point-count parity cases).
template <typename H, typename V8> __device__ __forceinline__ V8 load8_ragged(const H* p, int nv) {
  V8 v;
#pragma unroll
  for (int e = 0; e < 8; e++) v[e] = (H)0.f;
#pragma unroll
  for (int e = 0; e < 8; e++) if (e < nv) v[e] = p[e];
  return v;
}
// v = the 8 elements that END with a row's last element (loaded from row_end - 8, always inside the tensor); returns
// them moved down by sh places (out[e] = v[e + sh], zero filled): the row's last, partial vector without a load that
// depends on the data path -- sh = 8 - P % 8 is uniform over the launch, so this is a scalar switch around four v_alignbit
template <typename V8> __device__ __forceinline__ V8 shift_down8(const V8& v, int sh) {
  typedef unsigned int u4_ __attribute__((ext_vector_type(4)));
  const u4_ w = __builtin_bit_cast(u4_, v);
  unsigned a0, a1, a2, a3;
  switch (sh >> 1) {
    case 0: a0 = w[0]; a1 = w[1]; a2 = w[2]; a3 = w[3]; break;
    case 1: a0 = w[1]; a1 = w[2]; a2 = w[3]; a3 = 0u; break;
    case 2: a0 = w[2]; a1 = w[3]; a2 = 0u; a3 = 0u; break;
    default: a0 = w[3]; a1 = 0u; a2 = 0u; a3 = 0u; break;
  }
  u4_ o;
  if (sh & 1) {
    o[0] = __builtin_amdgcn_alignbit(a1, a0, 16); o[1] = __builtin_amdgcn_alignbit(a2, a1, 16);
    o[2] = __builtin_amdgcn_alignbit(a3, a2, 16); o[3] = a3 >> 16;
  } else {
    o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3;
  }
  return __builtin_bit_cast(V8, o);
}
template <typename H, typename V8> __device__ __forceinline__ void store8_ragged(H* p, const V8& v, int nv) {
#pragma unroll
  for (int e = 0; e < 8; e++) if (e < nv) p[e] = v[e];
}
template <typename H> struct HV;
template <> struct HV<bf16> {
  typedef bf16x8 x8; typedef bf16x4 x4; typedef bf16x2 x2;
  static constexpr const char* name = "bf16";
};
template <> struct HV<f16> {
  typedef f16x8 x8; typedef f16x4 x4; typedef f16x2 x2;
  static constexpr const char* name = "f16";
};
template <typename T> struct TypeName { static constexpr const char* v = HV<T>::name; };
template <> struct TypeName<float> { static constexpr const char* v = "float"; };
// D[32x32] += A[32x16] * B[16x32]: v_mfma_f32_32x32x16_bf16 / v_mfma_f32_32x32x16_f16 (same operand layout, same rate)
template <typename H>
__device__ __forceinline__ f32x16 mfma16(typename HV<H>::x8 a, typename HV<H>::x8 b, f32x16 c) {
  if constexpr (__is_same(H, bf16)) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
// dtype code of the C ABI -> is it one of the 16-bit storage types
static inline bool x3d_is_half(int dtype) { return dtype == X3D_BF16 || dtype == X3D_F16; }
static inline bool x3d_dtype_ok(int dtype) { return dtype == X3D_F32 || x3d_is_half(dtype); }

#define WAVE 64

// ---------------------------------------------------------------------------------------------
// error reporting across the C ABI: status codes + a thread-local message (x3d_last_error()).
// ---------------------------------------------------------------------------------------------
void x3d_set_error(const char* fmt, ...);

#define X3D_REQUIRE(cond, ...)                 \
  do {                                         \
    if (!(cond)) {                             \
      x3d_set_error(__VA_ARGS__);              \
      return X3D_ERR_INVALID;                  \
    }                                          \
  } while (0)

#define X3D_LAUNCH_CHECK(name)                                               \
  do {                                                                       \
    hipError_t e_ = hipGetLastError();                                       \
    if (e_ != hipSuccess) {                                                  \
      x3d_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));   \
      return X3D_ERR_LAUNCH;                                                 \
    }                                                                        \
  } while (0)

// x3d_dw3d_kernel_name(): when this buffer is set the depthwise launch helpers run their whole dispatch
// but write the chosen instantiation's name here instead of launching it
struct X3dDescribe { char* out; int cap; };
extern thread_local X3dDescribe x3d_describe;
// x3d_pw_wgrad_dw_parts(): with this set a weight-gradient launcher that has the partial-slab form stores the number of
// slabs its launch would write (its point-chunk grid) and returns instead of launching; launchers without the form leave it
extern thread_local int* x3d_parts_query;

// dry-run dispatch (x3d_dw3d_kernel_name / x3d_pw_kernel_name): a launcher that reaches this line with the describe buffer
// set writes the name of the instantiation it chose and returns instead of launching (no HIP call has been made)
#define X3D_DESCRIBE(...)                                              \
  do {                                                                 \
    if (x3d_describe.out) {                                            \
      snprintf(x3d_describe.out, x3d_describe.cap, __VA_ARGS__);       \
      return X3D_OK;                                                   \
    }                                                                  \
  } while (0)

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline long long ceil_div_ll(long long a, long long b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------------------------------------
// storage-type conversion.  Arithmetic is always fp32; T is only the HBM storage type.
// ---------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float to_f(T v);
template <> __device__ __forceinline__ float to_f<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f<bf16>(bf16 v) { return (float)v; }
template <> __device__ __forceinline__ float to_f<f16>(f16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f<bf16>(float v) { return (bf16)v; }
template <> __device__ __forceinline__ f16 from_f<f16>(float v) { return (f16)v; }
// value as it will read back from HBM (used so batch statistics describe the stored tensor)
template <typename T> __device__ __forceinline__ float round_to(float v) { return to_f<T>(from_f<T>(v)); }

// VEC contiguous elements -> fp32 registers (VEC in {1,2,4,8}; caller guarantees alignment)
template <typename T, int VEC> struct VecIO;   // primary template (below): 16-bit types; float specialised
template <int VEC> struct VecIO<float, VEC> {
  static __device__ __forceinline__ void load(const float* p, float (&o)[VEC]) {
    if constexpr (VEC == 8) {
      f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
#pragma unroll
      for (int i = 0; i < 4; i++) { o[i] = a[i]; o[4 + i] = b[i]; }
    } else if constexpr (VEC == 4) {
      f32x4 a = *(const f32x4*)p;
#pragma unroll
      for (int i = 0; i < 4; i++) o[i] = a[i];
    } else if constexpr (VEC == 2) {
      float2 a = *(const float2*)p; o[0] = a.x; o[1] = a.y;
    } else { o[0] = *p; }
  }
  static __device__ __forceinline__ void store(float* p, const float (&v)[VEC]) {
    if constexpr (VEC == 8) {
      f32x4 a, b;
#pragma unroll
      for (int i = 0; i < 4; i++) { a[i] = v[i]; b[i] = v[4 + i]; }
      *(f32x4*)p = a; *(f32x4*)(p + 4) = b;
    } else if constexpr (VEC == 4) {
      f32x4 a;
#pragma unroll
      for (int i = 0; i < 4; i++) a[i] = v[i];
      *(f32x4*)p = a;
    } else if constexpr (VEC == 2) {
      *(float2*)p = make_float2(v[0], v[1]);
    } else { *p = v[0]; }
  }
};
template <typename H, int VEC> struct VecIO {   // the 16-bit storage types
  typedef typename HV<H>::x8 hx8; typedef typename HV<H>::x4 hx4; typedef typename HV<H>::x2 hx2;
  static __device__ __forceinline__ void load(const H* p, float (&o)[VEC]) {
    if constexpr (VEC == 8) {
      hx8 a = *(const hx8*)p;
#pragma unroll
      for (int i = 0; i < 8; i++) o[i] = (float)a[i];
    } else if constexpr (VEC == 4) {
      hx4 a = *(const hx4*)p;
#pragma unroll
      for (int i = 0; i < 4; i++) o[i] = (float)a[i];
    } else if constexpr (VEC == 2) {
      hx2 a = *(const hx2*)p; o[0] = (float)a[0]; o[1] = (float)a[1];
    } else { o[0] = (float)*p; }
  }
  static __device__ __forceinline__ void store(H* p, const float (&v)[VEC]) {
    if constexpr (VEC == 8) {
      hx8 a;
#pragma unroll
      for (int i = 0; i < 8; i++) a[i] = (H)v[i];
      *(hx8*)p = a;
    } else if constexpr (VEC == 4) {
      hx4 a;
#pragma unroll
      for (int i = 0; i < 4; i++) a[i] = (H)v[i];
      *(hx4*)p = a;
    } else if constexpr (VEC == 2) {
      hx2 a; a[0] = (H)v[0]; a[1] = (H)v[1];
      *(hx2*)p = a;
    } else { *p = (H)v[0]; }
  }
};

// largest power-of-two vector (<= cap elements, <= 16 bytes) dividing every extent given and
// compatible with the pointer alignments
static inline int pick_vec(int elem_bytes, long long extent, const void* p0, const void* p1 = nullptr,
                           const void* p2 = nullptr, const void* p3 = nullptr) {
  int v = 16 / elem_bytes;
  const void* ps[4] = {p0, p1, p2, p3};
  while (v > 1) {
    bool ok = (extent % v) == 0;
    for (int i = 0; i < 4 && ok; i++)
      if (ps[i] && ((uintptr_t)ps[i] % (size_t)(v * elem_bytes)) != 0) ok = false;
    if (ok) break;
    v >>= 1;
  }
  return v;
}

// ---------------------------------------------------------------------------------------------
// stride-(1,2,2) 'valid' 1x1x1 shortcut conv: 8 consecutive OUTPUT points (t, ho, wo..) need the even elements of
// their input rows.  The outputs are taken in groups of GV (GV | Wo, so a group never crosses an output row):
// one aligned load of 2*GV input elements per group (16 / 8 / 4 bytes for GV = 4 / 2 / 1).  lo | hi hold the 16
// gathered elements in order, so output j of the vector is element 2*j whatever GV is.
// Caller guarantees: Wo % GV == 0, base 4*GV-byte aligned, all 8 outputs inside the tensor, and either W % (2*GV) == 0
// (aligned loads) or W odd (round 2: 39 -> 20, the X3D-L / XL shortcuts).  With W odd the loads are unaligned -- which the
// compute queues allow -- and the last group of a row would read one element past the row end (W = 2*Wo - 1): that group is
// loaded one element EARLY and moved down by 16 bits, so nothing past the tensor is touched.
// ---------------------------------------------------------------------------------------------
// nout: how many of the 8 outputs exist (a multiple of GV; < 8 only where a row of P % 8 != 0 points ends inside the
// vector): the groups past it are not loaded (zeros).
template <int GV, typename HT>
__device__ __forceinline__ void strided_gather16(const HT* base, long long p, int H, int W, int Ho, int Wo,
                                                 typename HV<HT>::x8& lo, typename HV<HT>::x8& hi, int nout = 8) {
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
  typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
  const int hw = Ho * Wo;                  // p < 2^31 (per-sample point index): 32-bit divisions
  long long t = (int)p / hw;
  const int rem = (int)p - (int)t * hw;
  int ho = rem / Wo, wo = rem - ho * Wo;
  unsigned int w[8];
#pragma unroll
  for (int gi = 0; gi < 8 / GV; gi++) {
    if (gi * GV >= nout) {
#pragma unroll
      for (int j = 0; j < GV; j++) w[gi * GV + j] = 0u;
      continue;
    }
    const bool early = (W & 1) && wo + GV == Wo;    // odd W: the row's last group, loaded from one element earlier
    const HT* src = base + (t * H + (long long)ho * 2) * W + (long long)wo * 2 - (early ? 1 : 0);
    if constexpr (GV == 4) {
      u32x4 v = *(const u32x4*)src;
      if (early) {
        v[0] = __builtin_amdgcn_alignbit(v[1], v[0], 16); v[1] = __builtin_amdgcn_alignbit(v[2], v[1], 16);
        v[2] = __builtin_amdgcn_alignbit(v[3], v[2], 16); v[3] = v[3] >> 16;
      }
      w[gi * 4] = v[0]; w[gi * 4 + 1] = v[1]; w[gi * 4 + 2] = v[2]; w[gi * 4 + 3] = v[3];
    } else if constexpr (GV == 2) {
      u32x2 v = *(const u32x2*)src;
      if (early) { v[0] = __builtin_amdgcn_alignbit(v[1], v[0], 16); v[1] = v[1] >> 16; }
      w[gi * 2] = v[0]; w[gi * 2 + 1] = v[1];
    } else {
      const unsigned int v = *(const unsigned int*)src;
      w[gi] = early ? v >> 16 : v;
    }
    wo += GV;
    if (wo >= Wo) { wo = 0; if (++ho >= Ho) { ho = 0; ++t; } }
  }
  const u32x4 l = {w[0], w[1], w[2], w[3]}, h = {w[4], w[5], w[6], w[7]};
  lo = __builtin_bit_cast(typename HV<HT>::x8, l);
  hi = __builtin_bit_cast(typename HV<HT>::x8, h);
}
// largest group size the gather supports for a stride-2 source of row length W sampled to Wo (0: use the scalar path)
static inline bool strided_odd_enabled() {
  return x3d_env_int("X3D_PW_STRIDED_ODD", 1) != 0;   // A/B hook: 0 = odd input widths on the scalar gather
}
static inline int strided_gather_gv(int W, int Wo, long long P, const void* x) {
  if ((P % 8) != 0 && !pw_ragged_rows(P, 2)) return 0;   // (P % 8 != 0: the RAG instantiations; P % GV == 0 since GV | Wo)
  for (int gv = 4; gv >= 1; gv >>= 1)
    // odd W: the row's last group is loaded one element early -- it must not also be the row's first group (Wo == gv: the
    // load of the tensor's very first row would start 2 bytes before the allocation; 7 -> 4 then takes gv = 2, 3 -> 2 gv = 1)
    if ((Wo % gv) == 0 && ((W % (2 * gv)) == 0 || ((W & 1) && Wo > gv && strided_odd_enabled())) && ((uintptr_t)x % (4 * gv)) == 0) return gv;
  return 0;
}

// ---------------------------------------------------------------------------------------------
// BatchNorm finalize (training): statistics -> (scale, shift), one channel.  ONE definition, used by
// bn_finalize_kernel and by the consumers that fold the finalize into their prologue (x3d_bn_fold), so the
// coefficients are the same bits whichever path computed them.  `writer`: exactly one thread per channel
// and launch publishes scale_shift / mean_invstd and updates the moving statistics.
// ---------------------------------------------------------------------------------------------
// Replicated statistics accumulators (include/x3d_hip.h): STATS_R copies of [C][2] doubles, stats_stride(C) apart.
// A/B switch X3D_XCD_PAD=0: do not pad grid.x to a multiple of the XCD count (see pw_bwd_fused.hip, pw_wgrad_bf16.h)
static inline bool xcd_pad_enabled() {
  return x3d_env_int("X3D_XCD_PAD", 1) != 0;
}

// BatchNorm backward finalize, one channel (x3d_bn_bwd_finalize; include/x3d_hip.h x3d_bn_bwd_fold): ONE definition, used by
// the finalize kernels and by the consumers that derive their coefficient table from the sums themselves, so the table is the
// same bits whichever path computed it.
//   dY = k1*(g - dbeta/M - xhat*dgamma/M), k1 = gamma*invstd, xhat = (y - mean)*invstd
//      = A*g + B*y + C with A = k1, B = -k1*invstd*dgamma/M, C = -k1*dbeta/M - B*mean
__device__ __forceinline__ void bn_bwd_coefs(const double* __restrict__ sums, double count, const float* __restrict__ mi,
                                             const float* __restrict__ gamma, int c, float& A, float& B, float& C, double& dga,
                                             double& dbe) {
  const double mean = mi[c * 2], invstd = mi[c * 2 + 1];
  dbe = sums[c * 2];
  dga = (sums[c * 2 + 1] - mean * dbe) * invstd;
  const double k1 = (double)gamma[c] * invstd;
  const double Bd = -k1 * invstd * dga / count;
  A = (float)k1;
  B = (float)Bd;
  C = (float)(-k1 * dbe / count - Bd * mean);
}
// by-value copy of x3d_bn_bwd_fold for kernel arguments (sums == NULL: off, the table `coef` is read)
struct BnBwdFold { const double* sums; double count; const float* mi; const float* gamma; float* dgamma; float* dbeta; float* coef_out; };
static inline BnBwdFold bn_bwd_fold_arg(const x3d_bn_bwd_fold* f) {
  BnBwdFold o;
  memset(&o, 0, sizeof(o));
  if (f) { o.sums = f->sums; o.count = f->count; o.mi = f->mean_invstd; o.gamma = f->gamma; o.dgamma = f->dgamma; o.dbeta = f->dbeta; o.coef_out = f->coef_out; }
  return o;
}
static inline bool bn_bwd_fold_ok(const x3d_bn_bwd_fold* f) {
  return !f || (f->sums && f->mean_invstd && f->gamma && f->count > 0 && ((f->dgamma != nullptr) == (f->dbeta != nullptr)));
}
// coefficients of channel k: from the table, or from the sums; `publish`: this thread is the ONE thread of the launch that
// handles channel k in workgroup (0, 0, 0) -- it adds the gamma / beta gradients and writes the table (when asked to)
__device__ __forceinline__ void bn_bwd_coef_load(const float* __restrict__ coef, const BnBwdFold& f, int k, bool publish, float& A,
                                                 float& B, float& C) {
  if (f.sums) {
    double dga, dbe;
    bn_bwd_coefs(f.sums, f.count, f.mi, f.gamma, k, A, B, C, dga, dbe);
    if (publish && f.dgamma) {
      f.dgamma[k] += (float)dga;
      f.dbeta[k] += (float)dbe;
      if (f.coef_out) { f.coef_out[k * 4] = A; f.coef_out[k * 4 + 1] = B; f.coef_out[k * 4 + 2] = C; f.coef_out[k * 4 + 3] = 0.f; }
    }
  } else {
    A = coef[k * 4]; B = coef[k * 4 + 1]; C = coef[k * 4 + 2];
  }
}

// CUs of the current device (256 on MI355X; also the answer without a device: dry-run dispatch / x3d_pw_bwd_dw_parts on a
// build host).  The persistent one-workgroup-per-CU kernels size their grids from it.
static inline int x3d_device_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    hipDeviceProp_t prop;
    if (hipGetDeviceCount(&n) == hipSuccess && n > 0 && hipGetDevice(&dev) == hipSuccess &&
        hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      cus = prop.multiProcessorCount;
    else
      cus = 256;
    (void)hipGetLastError();
  }
  return cus;
}
// grid of a persistent kernel over `total` work items on `cus` CUs: an equal share per workgroup, no empty workgroup
static inline void x3d_persistent_grid(long long total, int cus, long long* per_block, long long* grid) {
  *per_block = (total + cus - 1) / cus;
  if (*per_block < 1) *per_block = 1;
  *grid = (total + *per_block - 1) / *per_block;
}

#define STATS_R 32
// copies the producers actually add into (the rest stay zero).  -DSTATS_USED=8: round-6 A/B of one copy per XCD with the finalize
// folded into the consumers (plan option bn_fold), whose per-workgroup sum over the copies is then 8 terms instead of 32
// (tools/ab_stats_used.sh, profiles/r06_ab_stats_used.txt)
#ifndef STATS_USED
#define STATS_USED 32
#endif
static_assert(STATS_USED == 32 || STATS_USED == 8, "STATS_USED: 32 (product) or 8 (experiment)");
__host__ __device__ __forceinline__ long long stats_stride(int C) {
  const long long need = ((long long)C * 2 + 63) & ~63ll;
  return need > 512 ? need : 512;        // >= 4 KB apart: the copies land in different L2 channels
}
// the copy a producer workgroup adds into; `key`: any index that differs between the workgroups that share a channel
__device__ __forceinline__ double* stats_replica(double* stats, int C, unsigned key) {
  return stats + (long long)(key % STATS_USED) * stats_stride(C);
}

// coefficients of one channel from its totals (sum, sum of squares); `writer` publishes them / updates the moving stats
__device__ __forceinline__ void bn_coefs(const x3d_bn_fold& f, int c, double sum1, double sum2, float ga, float be,
                                         bool writer, float& sc, float& sh) {
  const double mean = sum1 / f.count;
  double var = sum2 / f.count - mean * mean;  // biased batch variance (Keras, training)
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)f.eps));
  sc = ga * invstd;
  sh = be - (float)mean * sc;
  if (writer) {
    f.scale_shift[c * 2] = sc;
    f.scale_shift[c * 2 + 1] = sh;
    f.mean_invstd[c * 2] = (float)mean;
    f.mean_invstd[c * 2 + 1] = invstd;
    if (f.update_moving) {
      // Keras momentum convention: moving = moving*momentum + batch*(1-momentum); the moving variance
      // receives the unbiased estimate (TF fused batch norm) [TF-3p]
      const double unb = f.count > 1.0 ? var * (f.count / (f.count - 1.0)) : var;
      f.moving_mean[c] = f.moving_mean[c] * f.momentum + (float)mean * (1.f - f.momentum);
      f.moving_var[c] = f.moving_var[c] * f.momentum + (float)unb * (1.f - f.momentum);
    }
  }
}
// one thread sums the STATS_R copies itself (the folded consumers, x3d_bn_fold): batches of 8 copies = 16 loads in
// flight and 32 VGPRs; all 32 at once would set the register count of the whole consumer kernel.  The copies are
// summed pairwise in the same tree as bn_finalize_kernel's lane reduction, so both paths give the same bits.
__device__ __forceinline__ void bn_fold_channel(const x3d_bn_fold& f, int c, bool writer, float& sc, float& sh, int C) {
  const float ga = f.gamma[c], be = f.beta[c];
  const long long rs = stats_stride(C);
  double q1[STATS_R / 8], q2[STATS_R / 8];
#pragma unroll
  for (int i = 0; i < STATS_R / 8; i++) { q1[i] = 0.0; q2[i] = 0.0; }      // (copies past STATS_USED hold zeros: not read)
#pragma unroll
  for (int r0 = 0; r0 < STATS_USED; r0 += 8) {
    double p1[8], p2[8];
#pragma unroll
    for (int r = 0; r < 8; r++) { p1[r] = f.stats[(r0 + r) * rs + c * 2]; p2[r] = f.stats[(r0 + r) * rs + c * 2 + 1]; }
    // xor-butterfly order over 8: (0+1, 2+3, 4+5, 6+7) -> pairs -> total
    q1[r0 / 8] = ((p1[0] + p1[1]) + (p1[2] + p1[3])) + ((p1[4] + p1[5]) + (p1[6] + p1[7]));
    q2[r0 / 8] = ((p2[0] + p2[1]) + (p2[2] + p2[3])) + ((p2[4] + p2[5]) + (p2[6] + p2[7]));
  }
  const double sum1 = (q1[0] + q1[1]) + (q1[2] + q1[3]), sum2 = (q2[0] + q2[1]) + (q2[2] + q2[3]);
  bn_coefs(f, c, sum1, sum2, ga, be, writer, sc, sh);
}
static inline bool bn_fold_valid(const x3d_bn_fold* f) {
  return f && f->stats && f->gamma && f->beta && f->scale_shift && f->mean_invstd && f->count > 0 &&
         (!f->update_moving || (f->moving_mean && f->moving_var));
}

// ---------------------------------------------------------------------------------------------
// activations
// ---------------------------------------------------------------------------------------------
// v_exp_f32 + v_rcp_f32 (1 ulp) instead of the ~10-instruction IEEE division sequence `1.0f / x` compiles to:
// swish sits in the load path of every `c` conv (forward, wgrad) and in the dgrad epilogue
__device__ __forceinline__ float sigmoidf_(float v) { return __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
__device__ __forceinline__ float swishf_(float v) { return v * sigmoidf_(v); }
__device__ __forceinline__ float swish_grad_(float v) {
  float s = sigmoidf_(v);
  return s * (1.0f + v * (1.0f - s));
}

// swish backward of the `c` conv's input on FOLDED per-row coefficients (the forward prologue folds the SE gate the same way:
// (s*g)*x + t*g):  u = su*b + tu ;  sigmoid(u) = 1 / (1 + 2^(nu*b + nt)),  (nu, nt) = -log2(e) * (su, tu) ;
// xh = swish(u) = u*sg ;  d = swish'(u) = sg + xh*(1 - sg).  11 VALU + 2 transcendental per element where
// (s*b + t)*g -> __expf -> s*(1 + u*(1 - s)) took 15 + 2: the epilogues that run it are VALU-bound (1.6 G elements per X3D-M step)
struct SwishCoef { float su, tu, nu, nt; };
__device__ __forceinline__ SwishCoef swish_coef(float s, float t, float g) {
  SwishCoef c;
  c.su = s * g; c.tu = t * g;
  c.nu = -1.4426950408889634f * c.su; c.nt = -1.4426950408889634f * c.tu;
  return c;
}
__device__ __forceinline__ void swish_bwd_(const SwishCoef& c, float b, float& xh, float& d) {
  const float u = fmaf(c.su, b, c.tu);
  const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(fmaf(c.nu, b, c.nt)));
  xh = u * sg;
  d = fmaf(xh, 1.0f - sg, sg);
}

// v[e] = act(v[e]) for a whole register vector with ONE uniform branch on the (runtime, kernel-uniform) activation.
// Written per element -- `if (act == RELU) u = max(u, 0); else if (act == SWISH) u = swish(u);` inside the unrolled
// loop -- the compiler kept a scalar compare + branch (+ s_nop) PER ELEMENT: 63 branches per staged chunk, +20 us on a
// 30 us GEMM (216->96 @ 14x14: 30.4 us without prologue, 52.2 us with a plain affine one, 35 us after this).
template <int N>
__device__ __forceinline__ void act_vec(float (&v)[N], int act) {
  if (act == X3D_ACT_SWISH) {
#pragma unroll
    for (int e = 0; e < N; e++) v[e] = swishf_(v[e]);
  } else {
    const float floor_ = act == X3D_ACT_RELU ? 0.f : -INFINITY;   // max(u, -inf) = u: no branch for "none" either
#pragma unroll
    for (int e = 0; e < N; e++) v[e] = fmaxf(v[e], floor_);
  }
}

// ---------------------------------------------------------------------------------------------
// reductions: 64-wide wavefront shuffles, then LDS across the waves of a block
// ---------------------------------------------------------------------------------------------
// DPP (data-parallel primitive) adds: cross-lane operands inside the VALU instruction, no LDS round trip.
// __shfl_xor compiles to ds_bpermute_b32 + s_waitcnt; the 29 wave sums at the end of the depthwise backward
// were 174 of those (r01c ISA).  Controls: quad_perm 0x00-0xFF, row_half_mirror 0x141, row_mirror 0x140,
// row_bcast15 0x142, row_bcast31 0x143; a lane whose row is masked off adds 0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_get(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
// every lane ends with the sum over its row of 16 lanes
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_get<0xB1, 0xF>(v);    // quad_perm [1,0,3,2]
  v += dpp_get<0x4E, 0xF>(v);    // quad_perm [2,3,0,1]
  v += dpp_get<0x141, 0xF>(v);   // row_half_mirror
  v += dpp_get<0x140, 0xF>(v);   // row_mirror
  return v;
}
// sum over the 64 lanes, valid in lane 63 only (no v_readlane / SGPR round trip: for many sums in a row)
__device__ __forceinline__ float wave_sum_lane63(float v) {
  v = row16_sum(v);
  v += dpp_get<0x142, 0xA>(v);   // row_bcast15 into rows 1 and 3
  v += dpp_get<0x143, 0xC>(v);   // row_bcast31 into rows 2 and 3
  return v;
}
// End of a one-wave depthwise backward (dw_mx.hip, dw_mxg.hip): 27 weight-gradient sums + the two BatchNorm-backward sums, each
// valid in lane 63 (wave_sum_lane63).  Lane 63 used to issue them as 29 single-lane atomic instructions -- 13 824 waves x 29
// lone dwords per launch, which the memory-side atomic unit takes one request at a time (~9 % of the launch).
// Here lane k takes sum k (a scalar read of lane 63 + one select each) and the wave issues TWO atomic instructions: 27 floats
// at 108 contiguous bytes, two doubles.
__device__ __forceinline__ void dw_flush_sums29(const float (&red)[29], int lane, float* dw27, double* sums2) {
  float mine = 0.f;
#pragma unroll
  for (int k = 0; k < 29; k++) {
    const float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, red[k]), 63));
    mine = lane == k ? b : mine;
  }
  if (lane < 27) atomicAdd(&dw27[lane], mine);
  else if (lane < 29) atomicAdd(&sums2[lane - 27], (double)mine);
}
// sum over the 64 lanes, returned to every lane (as a wave-uniform value)
__device__ __forceinline__ float wave_sum(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wave_sum_lane63(v)), 63));
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// sum over the 32 lanes that share (lane >> 5)
__device__ __forceinline__ float half_wave_sum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// the same sum on DPP adds only (the form above is five ds_bpermute round trips PER VALUE, each waited for: the 32 sums at the end
// of an fp32 pointwise tile walk were ~18,000 cycles of a 64,000-cycle workgroup, in-kernel stamps profiles/r06_f32p_stamps.txt):
// valid in the UPPER 16 lanes of each half (lane & 16), i.e. lanes 16-31 hold the sum of lanes 0-31, lanes 48-63 that of 32-63
__device__ __forceinline__ float half_wave_sum_hi(float v) {
  v = row16_sum(v);
  v += dpp_get<0x142, 0xA>(v);   // row_bcast15 into rows 1 and 3: lane 15 of the row below
  return v;
}

// block-wide sum of NV floats per thread; result valid in thread 0.  scratch: >= NV * (blockDim/64) floats.
template <int NV>
__device__ __forceinline__ void block_sum(float (&v)[NV], float* scratch) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int i = 0; i < NV; i++) v[i] = wave_sum(v[i]);
  if (nw == 1) return;
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < NV; i++) scratch[wid * NV + i] = v[i];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < NV; i++) {
      float s = 0.f;
      for (int w = 0; w < nw; w++) s += scratch[w * NV + i];
      v[i] = s;
    }
  }
}

__device__ __forceinline__ void atomic_add_d(double* p, double v) { atomicAdd(p, v); }
