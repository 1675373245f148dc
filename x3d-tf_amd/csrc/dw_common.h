// Channelwise 3x3x3 convolution (stride (1,s,s), TF-SAME padding): forward and fused backward.
//
// HBM-bound: 27 FMAs per output against 2-10 bytes, so the design goal is "read every input element
// once, write every output once, never wait for a load".  One workgroup owns one (n, c, H-tile) and
// streams the T planes of that channel through LDS:
//   global --(vector load into registers, issued ONE PLANE AHEAD)--> folded BN+ReLU, once per element
//          --> LDS plane with zero halo (= the TF-SAME / temporal zero padding)
// The staging map (which global vector lands where in LDS) is the same for every plane, so it is computed
// once per thread before the T loop.  Each thread owns a strip of SW consecutive outputs of one row and
// keeps THREE partial output planes (t-1, t, t+1) in registers: a staged plane is read from LDS once and
// scattered into the three temporal taps, so no plane is ever re-read and only one plane lives in LDS.
// Per-channel BatchNorm statistics and the squeeze-excite pool are reduced in the epilogue
// (wave shuffles -> LDS -> one fp64 atomic per workgroup).
#pragma once
#include <stdlib.h>

#include "common.h"

struct DwGeom {
  int N, C, T, H, W, Ho, Wo, S;
  int ph, pw;        // TF-SAME pad_before along H, W
  int TH;            // output rows per tile
  int ntile_h;
  int nstrips;       // strips of SW outputs per row
  int RIN;           // staged input rows  = (TH-1)*S + 3
  int LP;            // LDS pitch (floats) = (nstrips*SW-1)*S + 3
  int vec;           // staging vector width along W (elements)
};

// ---- a staged vector: up to 16 raw bytes held in registers between the load and the LDS write
struct Raw { uint32_t w[4]; };

template <typename T>
__device__ __forceinline__ void raw_load(Raw& r, const T* p, int vec) {
  const int bytes = vec * (int)sizeof(T);
  if (bytes == 16) { const uint4 v = *(const uint4*)p; r.w[0] = v.x; r.w[1] = v.y; r.w[2] = v.z; r.w[3] = v.w; }
  else if (bytes == 8) { const uint2 v = *(const uint2*)p; r.w[0] = v.x; r.w[1] = v.y; }
  else if (bytes == 4) { r.w[0] = *(const uint32_t*)p; }
  else { r.w[0] = *(const uint16_t*)p; }
}
template <typename T> __device__ __forceinline__ float raw_get(const Raw& r, int e);
template <> __device__ __forceinline__ float raw_get<float>(const Raw& r, int e) { return __uint_as_float(r.w[e]); }
template <> __device__ __forceinline__ float raw_get<bf16>(const Raw& r, int e) {
  return __uint_as_float(((r.w[e >> 1] >> (16 * (e & 1))) & 0xffffu) << 16);
}
template <> __device__ __forceinline__ float raw_get<f16>(const Raw& r, int e) {   // v_cvt_f32_f16 (with op_sel for the high half)
  return (float)__builtin_bit_cast(f16, (unsigned short)(r.w[e >> 1] >> (16 * (e & 1))));
}

// ---- argument blocks shared by the streaming kernels (dw_fwd.hip / dw_bwd.hip) and their deep-prefetch
// variants for small planes (dw_pd.hip)
struct DwFwdArgs {
  DwGeom g;
  const void* x; const float* w; void* y;
  const float* ss; int act;
  double* stats; double* pool;
  x3d_bn_fold bn;   // bn.stats != nullptr: scale/shift of the prologue from these statistics (BN finalize folded in)
  int exp;          // timing hooks (X3D_DW_FWD_EXP, experiments builds only; 0 in the product): 1 = skip the tap sums, 2 = no global access
};
struct DwBwdArgs {
  DwGeom g;
  const void* dv; const void* braw; const float* coef_nc;
  const void* araw; const float* ss_a; const float* w;
  void* ga; double* a_sums; float* dw;
  int LPB, RB;  // dB plane pitch / rows
  int vecB;     // staging vector width for the dv / braw planes
  int exp;      // timing hooks of the deep-prefetch kernel (X3D_DW_PD_EXP, experiments builds only; 0 in the product):
                // 1 = skip the tap sums, 2 = every global access out of range, 4 = skip the emit arithmetic
};
// prefetch depth (planes in flight per workgroup) of the deep-prefetch variants for strips of SW outputs:
// 4 for SW <= 2 (rows of < 20 outputs), else 1 (one-plane-ahead kernels).  X3D_DW_PD=1 switches them off (A/B hook).
int dw_pick_pd(int SW);
// deep-prefetch launchers (dw_pd.hip); return false when the shape is not covered (caller falls back)
bool dw_fwd_pd_launch(const DwFwdArgs& a, int dtype, int S, int SW, int cv, int pd, unsigned grid, int bd,
                      size_t lds, hipStream_t st);
bool dw_bwd_pd_launch(const DwBwdArgs& a, int dtype, int S, int SW, int cv, int pd, unsigned grid, int bd,
                      size_t lds, hipStream_t st);

// stride 2, strips of two outputs, aligned 16-byte staging vectors (dw_s2.hip): register roles + packed FMAs
// stride 1, strips of four outputs, rows of whole 16-byte vectors in 16-bit storage (dw_s1.hip): the same form
bool dw_bwd_s1_launch(const DwBwdArgs& a, int dtype, int SW, int cv, unsigned grid, int bd, hipStream_t st);
bool dw_bwd_s2_launch(const DwBwdArgs& a, int dtype, int SW, int cv, int pd, unsigned grid, int bd, size_t lds, hipStream_t st);

// packed variant for small stride-1 planes (dw_pk.hip): several samples of one channel per 512-thread workgroup
bool dw_bwd_pk_launch(const DwBwdArgs& a, int dtype, int S, int SW, hipStream_t st);
bool dw_fwd_pk_launch(const DwFwdArgs& a, int dtype, int S, int SW, hipStream_t st);
// matrix-core variant for 14x14 stride-1 planes in 16-bit storage (dw_mx.hip): the tap sums as Toeplitz products on MFMA
bool dw_fwd_mx_launch(const DwFwdArgs& a, int dtype, int S, hipStream_t st);
bool dw_bwd_mx_launch(const DwBwdArgs& a, int dtype, int S, hipStream_t st);
bool dw_bwd_mxw_launch(const DwBwdArgs& a, int dtype, int S, hipStream_t st);   // rows of 18 .. 30 elements (28 x 28, 20 x 20), H-tiled
bool dw_bwd_mxg_launch(const DwBwdArgs& a, int dtype, int S, hipStream_t st);   // ragged rows of 26 columns and more (39, 78 ...), H- and W-tiled (dw_mxg.hip)

// ---- bounds-checked buffer accesses of BYTES (2/4/8/16) per lane: an out-of-range offset (voff + soff >= the
// resource's num_records) loads zeros / drops the store WITHOUT touching memory, so the instruction itself can be
// unconditional.  That matters for more than the branch: vmcnt retires in order and the compiler must assume a
// conditional memory operation was not issued, which turns every wait behind one into vmcnt(0).
typedef __attribute__((ext_vector_type(4))) unsigned int dw_u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int dw_u32x2;
#define DW_OOB 0x40000000
template <int BYTES>
__device__ __forceinline__ void raw_bload(Raw& r, __amdgpu_buffer_rsrc_t rs, int voff, int soff) {
  if constexpr (BYTES == 16) { const dw_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0); r.w[0] = v[0]; r.w[1] = v[1]; r.w[2] = v[2]; r.w[3] = v[3]; }
  else if constexpr (BYTES == 8) { const dw_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0); r.w[0] = v[0]; r.w[1] = v[1]; }
  else if constexpr (BYTES == 4) { r.w[0] = __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0); }
  else { r.w[0] = __builtin_amdgcn_raw_buffer_load_b16(rs, voff, soff, 0); }
}
template <int BYTES>
__device__ __forceinline__ void raw_bstore(const Raw& r, __amdgpu_buffer_rsrc_t rs, int voff, int soff) {
  if constexpr (BYTES == 16) { const dw_u32x4 v = {r.w[0], r.w[1], r.w[2], r.w[3]}; __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff, soff, 0); }
  else if constexpr (BYTES == 8) { const dw_u32x2 v = {r.w[0], r.w[1]}; __builtin_amdgcn_raw_buffer_store_b64(v, rs, voff, soff, 0); }
  else if constexpr (BYTES == 4) { __builtin_amdgcn_raw_buffer_store_b32(r.w[0], rs, voff, soff, 0); }
  else { __builtin_amdgcn_raw_buffer_store_b16((unsigned short)r.w[0], rs, voff, soff, 0); }
}
// N fp32 values -> N elements of storage type T packed into a Raw
template <typename T, int N> __device__ __forceinline__ void raw_pack(Raw& r, const float (&v)[N]) {
  if constexpr (sizeof(T) == 4) {
#pragma unroll
    for (int i = 0; i < N; i++) r.w[i] = __float_as_uint(v[i]);
  } else {
#pragma unroll
    for (int i = 0; i < (N + 1) / 2; i++) {
      const uint32_t lo = __builtin_bit_cast(unsigned short, (T)v[2 * i]);
      const uint32_t hi = (2 * i + 1 < N) ? (uint32_t)__builtin_bit_cast(unsigned short, (T)v[2 * i + 1]) : 0u;
      r.w[i] = lo | (hi << 16);
    }
  }
}

template <typename T> struct MaxVec { static constexpr int v = 16 / sizeof(T); };

// staging map of one plane tile: vector i of this thread reads goff[i] (elements from the plane origin,
// -1 = nothing to load) and writes lds[loff[i] ...]
template <int NSV>
struct StageMap {
  int goff[NSV], loff[NSV];
  __device__ __forceinline__ void build(int RIN, int LP, int row0, int H, int W, int pw, int vec) {
    const int nvr = W / vec, total = RIN * nvr;
#pragma unroll
    for (int i = 0; i < NSV; i++) {
      const int v = threadIdx.x + i * blockDim.x;
      goff[i] = -1; loff[i] = 0;
      if (v < total) {
        const int lr = v / nvr, jv = v - lr * nvr;
        const int hi = row0 + lr;
        if (hi >= 0 && hi < H) { goff[i] = hi * W + jv * vec; loff[i] = lr * LP + pw + jv * vec; }
      }
    }
  }
};

// Ragged planes (a row is not a whole number of aligned vectors: W = 39, 78, 91 ...).  A tile covers whole rows, so
// its valid rows are ONE contiguous run of elements in memory: the run is cut into vectors of `vec` elements whatever
// the row length (the loads are then unaligned, which the compute queues' unaligned access mode allows; the last
// vector is moved back to end with the run so that nothing past the tensor is touched).  A vector may cross from one
// row to the next: elements k >= wrap[i] land `skip` = LP - W floats further on in LDS.  Needs W >= vec.
template <bool B, typename A, typename C> struct DwSel { typedef A type; };
template <typename A, typename C> struct DwSel<false, A, C> { typedef C type; };
template <int NSV>
struct FlatMap {
  int goff[NSV], loff[NSV], wrap[NSV];
  __device__ __forceinline__ void build(int RIN, int LP, int row0, int H, int W, int pw, int vec) {
    const int rlo = row0 > 0 ? row0 : 0, rhi = (row0 + RIN < H) ? row0 + RIN : H;
    const int span = (rhi - rlo) * W, nvec = (span + vec - 1) / vec;
#pragma unroll
    for (int i = 0; i < NSV; i++) {
      const int v = threadIdx.x + i * blockDim.x;
      goff[i] = -1; loff[i] = 0; wrap[i] = vec;
      if (v < nvec) {
        int e = v * vec;
        if (e > span - vec) e = span - vec;
        const int lr = e / W, col = e - lr * W;
        goff[i] = rlo * W + e; loff[i] = (rlo - row0 + lr) * LP + pw + col; wrap[i] = W - col;
      }
    }
  }
};
// LDS commit of one ragged vector: element k of the vector -> d[k] or, past the row end, d[k + skip]
template <typename T, int VEC, typename F>
__device__ __forceinline__ void flat_commit(float* d, int wrap, int skip, const Raw& r, F f) {
  float* d2 = d + skip;
#pragma unroll
  for (int k = 0; k < VEC; k++) (k >= wrap ? d2 : d)[k] = f(raw_get<T>(r, k));
}
template <typename T, int VEC, typename F>
__device__ __forceinline__ void flat_commit2(float* d, int wrap, int skip, const Raw& r0, const Raw& r1, F f) {
  float* d2 = d + skip;
#pragma unroll
  for (int k = 0; k < VEC; k++) (k >= wrap ? d2 : d)[k] = f(raw_get<T>(r0, k), raw_get<T>(r1, k));
}
// staging vectors per thread of the ragged map
static int dw_nsv_flat(int rows, int W, int vec, int bd) { return ceil_div(ceil_div(rows * W, vec), bd); }
// ragged staging width for rows of at least `wmin` elements: 16 bytes, else 8 bytes (16-bit types), else 0 = not covered
static int dw_flat_vec(int elem_bytes, int wmin) {
  if (wmin >= 16 / elem_bytes) return 16 / elem_bytes;
  if (elem_bytes == 2 && wmin >= 4) return 4;
  return 0;
}

// generic (no prefetch) staging for tiles with more vectors per thread than the register budget
template <typename T, typename F>
__device__ __forceinline__ void stage_direct(const T* src, float* lds, int RIN, int LP, int row0, int H, int W,
                                             int pw, int vec, F f) {
  const int nvr = W / vec, total = RIN * nvr;
  for (int v = threadIdx.x; v < total; v += blockDim.x) {
    const int lr = v / nvr, jv = v - lr * nvr;
    const int hi = row0 + lr;
    if (hi >= 0 && hi < H) {
      Raw r;
      raw_load<T>(r, src + (long long)hi * W + jv * vec, vec);
      float* d = lds + lr * LP + pw + jv * vec;
#pragma unroll
      for (int e = 0; e < MaxVec<T>::v; e++) if (e < vec) d[e] = f(raw_get<T>(r, e));
    }
  }
}
template <typename T, typename F>
__device__ __forceinline__ void stage_direct2(const T* s0, const T* s1, float* lds, int RIN, int LP, int row0, int H,
                                              int W, int pw, int vec, F f) {
  const int nvr = W / vec, total = RIN * nvr;
  for (int v = threadIdx.x; v < total; v += blockDim.x) {
    const int lr = v / nvr, jv = v - lr * nvr;
    const int hi = row0 + lr;
    if (hi >= 0 && hi < H) {
      Raw r0, r1;
      raw_load<T>(r0, s0 + (long long)hi * W + jv * vec, vec);
      raw_load<T>(r1, s1 + (long long)hi * W + jv * vec, vec);
      float* d = lds + lr * LP + pw + jv * vec;
#pragma unroll
      for (int e = 0; e < MaxVec<T>::v; e++) if (e < vec) d[e] = f(raw_get<T>(r0, e), raw_get<T>(r1, e));
    }
  }
}


// A thread's window of N consecutive LDS floats starting at a multiple of AL floats (AL = 4, 2 or 1; the pitches
// are multiples of 4).  Read as aligned ds_read_b128 / b64 pieces: with scalar ds_read_b32 the lanes of a wave sit
// 4*SW bytes apart and collide 4-way on the 32 banks (SW = 4), which made the stage-2/3 depthwise kernels LDS bound.
template <int N, int AL>
__device__ __forceinline__ void lds_window(const float* __restrict__ p, float (&w)[N]) {
  int i = 0;
  if constexpr (AL >= 4) {
#pragma unroll
    for (; i + 4 <= N; i += 4) {
      const f32x4 v = *(const f32x4*)(p + i);
      w[i] = v[0]; w[i + 1] = v[1]; w[i + 2] = v[2]; w[i + 3] = v[3];
    }
  }
  if constexpr (AL >= 2) {
#pragma unroll
    for (; i + 2 <= N; i += 2) {
      const float2 v = *(const float2*)(p + i);
      w[i] = v.x; w[i + 1] = v.y;
    }
  }
#pragma unroll
  for (; i < N; i++) w[i] = p[i];
}

// ---------------------------------------------------------------------------------------------
// Packed fp32 arithmetic for the FORWARD tap loop.  v_pk_fma_f32 retires two FMAs per lane and instruction on an
// even-aligned register pair, op_sel picking either half of each source for each half of the result.  Pairs that
// need no data movement:
//   * over the TEMPORAL taps: one staged value feeds three output planes, (out[t-1], out[t]) += (w_kt2, w_kt1) * v
//     with the weight pair in an SGPR pair and v broadcast -- 3 FMAs in 2 instructions, any strip width;
//   * the third plane over adjacent outputs (i, i+1) of the strip when the two inputs are an aligned register pair
//     of the window (stride 1, even i + kw): 2 FMAs in 1 instruction.
// Every product is added to the same accumulator in the same order as the scalar code: results are bit-identical.
// Measured (tools/ab_dw.py, r01h): forward stride-1 layers 4-16 % faster.  tools/micro/pk_rate.hip: a packed FMA does
// not have twice the FMA throughput of v_fma_f32 on MI355X (1.15-1.2x in dependent chains), so the gain is issue
// slots, not arithmetic rate; in the fused backward the pair constraints cost ~20 VGPRs and a wave of occupancy
// and it got slower (570 -> 690 us at 56x56), so the backward kernels keep scalar FMAs.
// ---------------------------------------------------------------------------------------------
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f bc2(float v) { return (v2f){v, v}; }

// One window row: acc01[i] += wp[kw] * x (the pair of planes), acc2[i] += w3[kw] * x (third plane, held as pairs
// over i), x = win[i*S + kw], kw ascending.
template <int S, int SW, int WIN>
__device__ __forceinline__ void dw_taps_row(const float (&win)[WIN], const v2f (&wp)[3], const float (&w3)[3],
                                            v2f (&acc01)[SW], v2f (&acc2p)[(SW + 1) / 2]) {
#pragma unroll
  for (int kw = 0; kw < 3; kw++) {
#pragma unroll
    for (int i = 0; i < SW; i++) acc01[i] = pk_fma(wp[kw], bc2(win[i * S + kw]), acc01[i]);
#pragma unroll
    for (int i = 0; i < SW; i += 2) {
      const int col = i * S + kw;
      if (S == 1 && i + 1 < SW && (col & 1) == 0) {
        acc2p[i / 2] = pk_fma(bc2(w3[kw]), (v2f){win[col], win[col + 1]}, acc2p[i / 2]);
      } else {
        acc2p[i / 2].x = __builtin_fmaf(w3[kw], win[col], acc2p[i / 2].x);
        if (i + 1 < SW) acc2p[i / 2].y = __builtin_fmaf(w3[kw], win[col + S], acc2p[i / 2].y);
      }
    }
  }
}
// rotate the three planes after a staged plane: fin <- plane (t-1) (complete), pair <- (plane t, plane t+1), third <- 0
template <int SW>
__device__ __forceinline__ void dw_rotate(float (&fin)[SW], v2f (&acc01)[SW], v2f (&acc2p)[(SW + 1) / 2]) {
#pragma unroll
  for (int i = 0; i < SW; i++) {
    fin[i] = acc01[i].x;
    acc01[i] = (v2f){acc01[i].y, (i & 1) ? acc2p[i / 2].y : acc2p[i / 2].x};
  }
#pragma unroll
  for (int j = 0; j < (SW + 1) / 2; j++) acc2p[j] = (v2f){0.f, 0.f};
}

// ---------------------------------------------------------------------------------------------
// Two-element dot products for the fused BACKWARD with 16-bit storage: v_dot2c_f32_bf16 / v_dot2c_f32_f16 retire two
// multiply-adds (16-bit operands, fp32 accumulator) per lane and instruction at the FULL VALU rate on gfx950
// (tools/micro/dot2_rate.hip: 135-144 lanes per CU and ns against 113 for dependent v_fma_f32; v_cvt_pk_bf16_f32, which
// forms a pair from two fp32 registers, runs at the v_fma_f32 rate).  The stride-1 fused backward is VALU-issue bound at
// 54 FMAs + ~30 other instructions per output, so the tap loops pair their products: the operands are
// rounded to the storage type first (what the matrix-core pointwise kernels do with theirs, and what the reference's
// mixed-precision policy does with the whole convolution), the sums stay fp32.
// ---------------------------------------------------------------------------------------------
template <typename T> struct Dot2 { static constexpr bool ok = false; };
template <> struct Dot2<bf16> {
  static constexpr bool ok = true;
  static __device__ __forceinline__ uint32_t pk(float lo, float hi) {
    bf16x2 q; q[0] = (bf16)lo; q[1] = (bf16)hi;
    return __builtin_bit_cast(uint32_t, q);
  }
  static __device__ __forceinline__ float dot(uint32_t a, uint32_t b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a), __builtin_bit_cast(bf16x2, b), c, false);
  }
};
template <> struct Dot2<f16> {
  static constexpr bool ok = true;
  static __device__ __forceinline__ uint32_t pk(float lo, float hi) {
    f16x2 q; q[0] = (f16)lo; q[1] = (f16)hi;
    return __builtin_bit_cast(uint32_t, q);
  }
  static __device__ __forceinline__ float dot(uint32_t a, uint32_t b, float c) {
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, a), __builtin_bit_cast(f16x2, b), c, false);
  }
};
template <> struct Dot2<float> {
  static constexpr bool ok = false;
  static __device__ __forceinline__ uint32_t pk(float, float) { return 0u; }
  static __device__ __forceinline__ float dot(uint32_t, uint32_t, float c) { return c; }
};
// X3D_DW_DOT=1 (bf16) / 2 (bf16 and fp16): the dot2 tap loops of dw3d_bwd_pk_kernel (A/B hook, default OFF).  Measured with
// tools/ab_dot.py (variants alternating inside one process, 216 ch x 64 clips of 16x14x14): scalar FMAs 110.6 us, dW on
// dot2 120.9, dA on dot2 112.4, both 113.0 on one box; 128.6 / 133.4 / 123.5 / 122.6 on another -- 13 % fewer VALU
// instructions buy nothing.  tools/micro/vgpr_banks.hip says why the count is the wrong measure on gfx950: a v_fmac_f32 /
// v_fma_f32 / v_mul / v_add / v_and / v_mov whose operands are all VGPRs (or inline constants) retires in ~3.0-3.4 clocks
// per wave64 instruction at four waves per SIMD, while the same instruction with an SGPR operand, v_dot2c_f32_bf16,
// v_pk_fma_f32, v_cvt_pk_bf16_f32, v_max_f32, v_lshlrev_b32 and every DPP form take ~4.9: a dot2 (2 MACs, 4.9 clocks) is no
// cheaper than two all-VGPR FMAs, and the pair conversions come on top.
static bool dw_use_dot(int dtype) {
  const int e = x3d_env_int("X3D_DW_DOT", 0);   // (tools/ab_dot.py switches it inside one process: X3D_EXPERIMENTS build)
  return e != 0 && dtype != X3D_F32 && (e == 2 || dtype == X3D_BF16);
}

// tile geometry shared by forward and backward
static int dw_geom(DwGeom& g, int N, int C, int T, int H, int W, int stride, int SW, int elem_bytes,
                   const void* p0, const void* p1, const void* p2, int* block_dim, size_t* lds_floats) {
  g.N = N; g.C = C; g.T = T; g.H = H; g.W = W; g.S = stride;
  g.Ho = ceil_div(H, stride); g.Wo = ceil_div(W, stride);
  const int tot_h = (g.Ho - 1) * stride + 3 - H, tot_w = (g.Wo - 1) * stride + 3 - W;
  g.ph = (tot_h > 0 ? tot_h : 0) / 2;
  g.pw = (tot_w > 0 ? tot_w : 0) / 2;
  g.nstrips = ceil_div(g.Wo, SW);
  int bd = 256;
  const int items = g.Ho * g.nstrips;
  if (items <= 64) bd = 64;
  else if (items <= 128) bd = 128;
  const int bd_env = x3d_env_int("X3D_DW_BD", 0);   // A/B hook: cap the workgroup size (64 / 128): more, smaller H-tiles
  if (bd_env >= 64 && bd_env < bd) bd = bd_env;
  if (g.nstrips > bd) return -1;
  int th = bd / g.nstrips;
  if (th > g.Ho) th = g.Ho;
  g.ntile_h = ceil_div(g.Ho, th);
  g.TH = ceil_div(g.Ho, g.ntile_h);
  g.RIN = (g.TH - 1) * stride + 3;
  g.LP = ((g.nstrips * SW - 1) * stride + 3 + 3) & ~3;   // multiple of 4 floats: window reads are 16/8-byte aligned vectors
  g.vec = pick_vec(elem_bytes, W, p0, p1, p2);
  *block_dim = bd;
  *lds_floats = (size_t)g.RIN * g.LP;
  return 0;
}

static int dw_pick_sw(int Wo) {
  const int e = x3d_env_int("X3D_DW_SW14", 0);    // experiment hook: strip width for 10 <= Wo < 20
  if (e > 0 && Wo >= 10 && Wo < 20) return e;
  const int e2 = x3d_env_int("X3D_DW_SW28", 0);   // experiment hook: strip width for Wo >= 20
  if (e2 > 0 && Wo >= 20) return e2;
  return Wo >= 20 ? 4 : (Wo >= 10 ? 2 : 1);
}

// staging vectors per thread for a [rows][W] plane tile
static int dw_nsv(int rows, int W, int vec, int bd) { return ceil_div(rows * (W / vec), bd); }

