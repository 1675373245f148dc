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
  static const char* e = getenv("X3D_DW_SW14");   // experiment hook: strip width for 10 <= Wo < 20
  if (e && Wo >= 10 && Wo < 20) return atoi(e);
  return Wo >= 20 ? 4 : (Wo >= 10 ? 2 : 1);
}

// staging vectors per thread for a [rows][W] plane tile
static int dw_nsv(int rows, int W, int vec, int bd) { return ceil_div(rows * (W / vec), bd); }

