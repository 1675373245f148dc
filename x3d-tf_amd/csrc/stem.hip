// X3D stem (reference model.py:134-210): conv_s = 1x3x3 stride (1,2,2) conv with symmetric pad
// (0,1,1), Cin=3 -> C1; conv_t = KTx1x1 temporal depthwise conv with pad (KT/2,0,0).  Cin = 3 and
// K = 27 pad to one 32x32 MFMA tile: bf16 storage runs conv_s forward and its weight gradient on the matrix cores
// (im2col tile built in LDS from 16-byte loads); fp32 storage and odd widths use the direct kernels.
#include <stdlib.h>

#include "common.h"

// ------------------------------------------------------------------------------------------------
// conv_s forward: one thread = one output position (t, ho, wo), all COUT channels in registers.
// ------------------------------------------------------------------------------------------------
template <typename T, int CIN, int COUT>
__global__ __launch_bounds__(256) void stem_s_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                         T* __restrict__ y, int Tn, int H, int W, int Ho, int Wo) {
  const int n = blockIdx.z, t = blockIdx.y;
  const int pos = blockIdx.x * blockDim.x + threadIdx.x;
  if (pos >= Ho * Wo) return;
  const int ho = pos / Wo, wo = pos - ho * Wo;
  float in[CIN][3][3];
#pragma unroll
  for (int ci = 0; ci < CIN; ci++) {
    const T* xp = x + (((long long)n * CIN + ci) * Tn + t) * H * W;
#pragma unroll
    for (int kh = 0; kh < 3; kh++) {
      const int hi = ho * 2 + kh - 1;
#pragma unroll
      for (int kw = 0; kw < 3; kw++) {
        const int wi = wo * 2 + kw - 1;
        in[ci][kh][kw] = (hi >= 0 && hi < H && wi >= 0 && wi < W) ? to_f<T>(xp[(long long)hi * W + wi]) : 0.f;
      }
    }
  }
  const long long oplane = (long long)Ho * Wo;
  T* yp = y + (((long long)n * COUT) * Tn + t) * oplane + pos;
#pragma unroll 4
  for (int co = 0; co < COUT; co++) {
    const float* wc = w + co * CIN * 9;
    float acc = 0.f;
#pragma unroll
    for (int ci = 0; ci < CIN; ci++)
#pragma unroll
      for (int kh = 0; kh < 3; kh++)
#pragma unroll
        for (int kw = 0; kw < 3; kw++) acc += wc[ci * 9 + kh * 3 + kw] * in[ci][kh][kw];
    yp[(long long)co * Tn * oplane] = from_f<T>(acc);
  }
}


// x staging of the matrix-core stem kernels when x is channels-last ([N][T][H][W][3], 16-bit): thread (q = segment of the step,
// kh, v) owns the 8 input pixels 2 wo0 + 8 v .. of input row 2 ho + kh - 1 -- 48 contiguous bytes, three 16-byte loads --
// and commits its three channels to the im2col tile exactly as the channel-planar form does (even pixels -> tap kw = 1,
// odd -> kw = 2 and, one point later, kw = 0; the pixel left of the segment for point 0 of kw = 0).
template <typename HT>
struct StemNhwcStage {
  typedef typename HV<HT>::x8 hx8; typedef typename HV<HT>::x4 hx4;
  int q, kh, v;
  bool on;
  hx8 r[3];
  HT rl[3];
  __device__ __forceinline__ void roles(int tid) {
    q = tid / 48;
    const int rem = tid - q * 48;
    kh = rem >> 4; v = rem & 15;
    on = tid < 192;
  }
  __device__ __forceinline__ void issue(const HT* __restrict__ x, int s0, int seg_end, int nws, int Ho, int Tn, int H, int W) {
#pragma unroll
    for (int i = 0; i < 3; i++) {
#pragma unroll
      for (int e = 0; e < 8; e++) r[i][e] = (HT)0.f;
      rl[i] = (HT)0.f;
    }
    const int seg = s0 + q;
    if (!on || seg >= seg_end) return;
    const int ws = seg % nws;
    int tmp = seg / nws;
    const int ho = tmp % Ho; tmp /= Ho;
    const int t = tmp % Tn;
    const int n = tmp / Tn;
    const int wo0 = ws * 64;
    const int hi = 2 * ho + kh - 1;
    if (hi >= 0 && hi < H && 2 * wo0 + 8 * v < W) {
      const HT* src = x + ((((long long)n * Tn + t) * H + hi) * W + 2 * wo0 + 8 * v) * 3;
      r[0] = *(const hx8*)src; r[1] = *(const hx8*)(src + 8); r[2] = *(const hx8*)(src + 16);
      if (v == 0 && wo0 > 0) { rl[0] = src[-3]; rl[1] = src[-2]; rl[2] = src[-1]; }
    }
  }
  __device__ __forceinline__ HT at(int i) const { return i < 8 ? r[0][i] : (i < 16 ? r[1][i - 8] : r[2][i - 16]); }
  __device__ __forceinline__ void commit(HT* Bs, int LP) const {
    if (!on) return;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const int tap1 = c * 9 + kh * 3 + 1;
      hx4 ev, od;
#pragma unroll
      for (int e = 0; e < 4; e++) { ev[e] = at(3 * (2 * e) + c); od[e] = at(3 * (2 * e + 1) + c); }
      *(hx4*)&Bs[tap1 * LP + q * 64 + 4 * v] = ev;          // kw = 1: wi = 2wo
      *(hx4*)&Bs[(tap1 + 1) * LP + q * 64 + 4 * v] = od;    // kw = 2: wi = 2wo + 1
      HT* k0 = &Bs[(tap1 - 1) * LP + q * 64 + 4 * v + 1];    // kw = 0: wi = 2wo - 1  (one point later)
#pragma unroll
      for (int e = 0; e < 4; e++)
        if (4 * v + 1 + e < 64) k0[e] = od[e];
      if (v == 0) Bs[(tap1 - 1) * LP + q * 64] = rl[c];
    }
  }
};

// ---- bf16 fast path of the stem forward on the matrix cores -----------------------------------------------
// The direct kernel above is VALU bound (648 FMAs + 27 two-byte loads per output position: 0.56 ms on X3D-M B=64
// against ~0.23 ms of HBM time).  Here the im2col tile [tap][point] is built in LDS exactly as in the weight-gradient
// kernel below (16-byte loads of the input rows, even/odd de-interleave = the three kw taps) and
//   Y[co][p] = sum_tap W[co][tap] * im2col[tap][p]
// is two k-steps of v_mfma_f32_32x32x16_bf16 per 32 points: A = the 32x32 weight tile (bf16, held in registers for
// the whole kernel), B = the im2col tile read transposed (ds_read_b64_tr_b16).  One wave = one 64-column segment of
// an output row; its 24x64 result goes through a private LDS slab so that the stores are 16-byte row pieces.
// NHWC = true: x is the caller's channels-last clip batch [N][T][H][W][3] itself (the reference's input layout): a thread
// loads the 8 pixels x 3 channels of its vector as three 16-byte loads and separates the channels in registers, so the
// stand-alone layout pass (x3d_nthwc_to_ncthw: 2 x 308 MB per X3D-M step) is not needed.  Staging roles then are
// (segment, kh, vector) -- 4 x 3 x 16 = 192 threads, each committing its three channels.
template <typename HT, int SEGS, bool NHWC>
__global__ __launch_bounds__(256) void stem_s_fwd_bf16_kernel(const HT* __restrict__ x, const float* __restrict__ w,
                                                              HT* __restrict__ y, int Cin, int Cout, int Tn, int H,
                                                              int W, int Ho, int Wo, int nws, int total_segs,
                                                              int segs_per_block) {
  typedef typename HV<HT>::x8 hx8; typedef typename HV<HT>::x4 hx4; typedef typename HV<HT>::x2 hx2;
  static_assert(SEGS == 4, "one wave per segment");
  constexpr int LP = SEGS * 64 + 32;        // 576 B pitch = 64 mod 256: the transposed read is conflict-free
  constexpr int OP = 64 + 4;                // fp32 slab pitch
  __shared__ __attribute__((aligned(16))) HT Bs[32 * LP];        // im2col [tap][point]
  __shared__ __attribute__((aligned(16))) float Os[SEGS * 32 * OP]; // per-wave output slab [co][64 points]
  typedef __attribute__((ext_vector_type(4))) short s16x4_s;
  typedef __attribute__((ext_vector_type(8))) short s16x8_s;
  typedef s16x4_s __attribute__((address_space(3))) * lds_s16x4_ptr;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, half = lane >> 5;
  const int ntap = Cin * 9;
  // segment indices fit 32 bits (host check): the (n, t, ho, ws) decomposition is four 32-bit divisions per segment
  // instead of 64-bit ones (each a ~100-instruction sequence, eight of them per iteration)
  const int seg_begin = blockIdx.x * segs_per_block;
  const int seg_end = min(seg_begin + segs_per_block, total_segs);
  for (int i = tid; i < 32 * LP / 8; i += 256) {
    hx8 z;
#pragma unroll
    for (int e = 0; e < 8; e++) z[e] = (HT)0.f;
    ((hx8*)Bs)[i] = z;
  }
  // A operand: row co = lane & 31, taps 8*half + 16*ks .. +7 (zero beyond Cout / ntap)
  hx8 afrag[2];
#pragma unroll
  for (int ks = 0; ks < 2; ks++)
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const int tap = ks * 16 + 8 * half + e;
      afrag[ks][e] = (HT)((r < Cout && tap < ntap) ? w[r * ntap + tap] : 0.f);
    }

  const int xrow = tid >> 4, xv = tid & 15;
  const int xci = xrow / 3, xkh = xrow - xci * 3;
  const bool xrow_ok = xrow < Cin * 3;
  hx8 rx[SEGS];
  HT rl[SEGS];
  StemNhwcStage<HT> st3;
  st3.roles(tid);
  auto issue = [&](int s0) {
    if constexpr (NHWC) { st3.issue(x, s0, seg_end, nws, Ho, Tn, H, W); return; }
#pragma unroll
    for (int q = 0; q < SEGS; q++) {
      const int seg = s0 + q;
      hx8 z;
#pragma unroll
      for (int e = 0; e < 8; e++) z[e] = (HT)0.f;
      rx[q] = z; rl[q] = (HT)0.f;
      if (seg >= seg_end) continue;
      const int ws = seg % nws;
      int tmp = seg / nws;
      const int ho = tmp % Ho; tmp /= Ho;
      const int t = tmp % Tn;
      const int n = tmp / Tn;
      const int wo0 = ws * 64;
      const int hi = 2 * ho + xkh - 1;
      if (xrow_ok && hi >= 0 && hi < H && 2 * wo0 + 8 * xv < W) {
        const HT* src = x + ((((long long)n * Cin + xci) * Tn + t) * H + hi) * W + 2 * wo0 + 8 * xv;
        rx[q] = *(const hx8*)src;
        if (xv == 0 && wo0 > 0) rl[q] = src[-1];
      }
    }
  };
  auto commit = [&]() {
    if constexpr (NHWC) { st3.commit(Bs, LP); return; }
#pragma unroll
    for (int q = 0; q < SEGS; q++) {
      if (xrow_ok) {
        const int tap1 = xci * 9 + xkh * 3 + 1;
        hx4 ev, od;
#pragma unroll
        for (int e = 0; e < 4; e++) { ev[e] = rx[q][2 * e]; od[e] = rx[q][2 * e + 1]; }
        *(hx4*)&Bs[tap1 * LP + q * 64 + 4 * xv] = ev;          // kw = 1: wi = 2wo
        *(hx4*)&Bs[(tap1 + 1) * LP + q * 64 + 4 * xv] = od;    // kw = 2: wi = 2wo + 1
        HT* k0 = &Bs[(tap1 - 1) * LP + q * 64 + 4 * xv + 1];    // kw = 0: wi = 2wo - 1  (one point later)
#pragma unroll
        for (int e = 0; e < 4; e++)
          if (4 * xv + 1 + e < 64) k0[e] = od[e];
        if (xv == 0) Bs[(tap1 - 1) * LP + q * 64] = rl[q];
      }
    }
  };

  // transposed-read geometry: lane -> (tap row 8*(g16>>1)+q4 (+4), 4 points)
  const int g16 = lane >> 4, q4 = (lane & 15) >> 2, pp = lane & 3;
  const int tr_row = 8 * (g16 >> 1) + q4;
  const int tr_col = wid * 64 + 16 * (g16 & 1) + 4 * pp;   // + 32 for the segment's second point tile
  float* myOs = Os + wid * 32 * OP;

  if (seg_begin < seg_end) issue(seg_begin);
  for (int s0 = seg_begin; s0 < seg_end; s0 += SEGS) {
    __syncthreads();
    commit();
    __syncthreads();
    if (s0 + SEGS < seg_end) issue(s0 + SEGS);
    const int seg = s0 + wid;
    f32x16 acc[2];
#pragma unroll
    for (int nt = 0; nt < 2; nt++) {
#pragma unroll
      for (int j = 0; j < 16; j++) acc[nt][j] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 2; ks++) {
        const s16x4_s b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(&Bs[(ks * 16 + tr_row) * LP + tr_col + nt * 32]));
        const s16x4_s b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(&Bs[(ks * 16 + tr_row + 4) * LP + tr_col + nt * 32]));
        const s16x8_s bs = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
        acc[nt] = mfma16<HT>(afrag[ks], __builtin_bit_cast(hx8, bs), acc[nt]);
      }
#pragma unroll
      for (int j = 0; j < 16; j++) myOs[((j & 3) + 8 * (j >> 2) + 4 * half) * OP + nt * 32 + r] = acc[nt][j];
    }
    // the slab is private to the wave: its own LDS writes are visible to it after the wait the compiler inserts
    if (seg < seg_end) {
      const int ws = seg % nws;
      int tmp = seg / nws;
      const int ho = tmp % Ho; tmp /= Ho;
      const int t = tmp % Tn;
      const int n = tmp / Tn;
      const int wo0 = ws * 64;
      // 32 rows x 8 vectors of 8 points = 256 vectors per wave, 4 per lane: row = i*8 + lane/8, vector = lane & 7
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int co = i * 8 + (lane >> 3), v = lane & 7;
        if (co < Cout && wo0 + 8 * v < Wo) {
          const f32x4 v0 = *(const f32x4*)&myOs[co * OP + 8 * v], v1 = *(const f32x4*)&myOs[co * OP + 8 * v + 4];
          float val[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
          HT* dst = y + ((((long long)n * Cout + co) * Tn + t) * Ho + ho) * Wo + wo0 + 8 * v;
          if (wo0 + 8 * v + 8 <= Wo) VecIO<HT, 8>::store(dst, val);
          else {   // Wo % 8 == 4 (W = 312): the row's last vector is half a vector
            const float lo[4] = {val[0], val[1], val[2], val[3]};
            VecIO<HT, 4>::store(dst, lo);
          }
        }
      }
    }
  }
}

// the matrix-core kernels take the clip batch channels-last as it is: 16-bit storage, three channels, rows of whole vectors
extern "C" int x3d_stem_s_nthwc_supported(int Cin, int W, int Cout, int dtype) {
  return (x3d_is_half(dtype) && Cin == 3 && (W % 8) == 0 && Cout <= 32) ? 1 : 0;
}

extern "C" int x3d_stem_s_fwd(const void* x, const float* w, void* y, int N, int Cin, int T, int H, int W,
                              int Cout, int dtype, int x_layout, void* stream) {
  X3D_REQUIRE(x && w && y && N > 0 && T > 0 && H > 0 && W > 0, "stem_s_fwd: bad args");
  X3D_REQUIRE(Cin == 3, "stem_s_fwd: Cin must be 3 (DATA.NUM_INPUT_CHANNELS)");
  X3D_REQUIRE(Cout == 24 || Cout == 32 || Cout == 8 || Cout == 16, "stem_s_fwd: unsupported Cout %d", Cout);
  X3D_REQUIRE(x3d_dtype_ok(dtype), "stem_s_fwd: bad dtype");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  dim3 grid(ceil_div(Ho * Wo, 256), T, N);
  hipStream_t st = (hipStream_t)stream;
  // (W % 8 == 0: input rows are whole 16-byte vectors, output rows whole 8-byte ones -- W = 312 has Wo = 156 = 19.5 vectors)
  X3D_REQUIRE(x_layout == X3D_LAYOUT_NCTHW || x_layout == X3D_LAYOUT_NTHWC, "stem_s_fwd: bad x_layout");
  const bool nhwc = x_layout == X3D_LAYOUT_NTHWC;
  X3D_REQUIRE(!nhwc || (x3d_stem_s_nthwc_supported(Cin, W, Cout, dtype) && ((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0),
              "stem_s_fwd: channels-last x needs 16-bit storage, W %% 8 == 0 and 16-byte aligned tensors (x3d_stem_s_nthwc_supported)");
  if (x3d_is_half(dtype) && (W % 8) == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0 && Cout <= 32) {
    // matrix-core path (weights rounded to bf16 like every pointwise conv): rows of x / y 16-byte aligned
    constexpr int SEGS = 4;
    const int nws = ceil_div(Wo, 64);
    const long long total_segs = (long long)N * T * Ho * nws;
    static int slots = 0;
    if (slots == 0) {
      int nb = 0, dev = 0, cus = 256;
      hipDeviceProp_t prop;
      if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, stem_s_fwd_bf16_kernel<bf16, SEGS, false>, 256, 0) != hipSuccess || nb < 1) nb = 2;
      slots = nb * cus;
    }
    long long spb2 = ceil_div_ll(total_segs, slots);
    if (spb2 < 4 * SEGS) spb2 = 4 * SEGS;
    spb2 = ceil_div_ll(spb2, SEGS) * SEGS;
    const long long gx2 = ceil_div_ll(total_segs, spb2);
    X3D_REQUIRE(total_segs + spb2 < (1ll << 31), "stem_s_fwd: too many row segments");
#define STEM_FWD(TT, NHWC_) hipLaunchKernelGGL((stem_s_fwd_bf16_kernel<TT, SEGS, NHWC_>), dim3((unsigned)gx2), dim3(256), 0, st, (const TT*)x, w, \
                                               (TT*)y, Cin, Cout, T, H, W, Ho, Wo, nws, (int)total_segs, (int)spb2)
    if (dtype == X3D_F16) { if (nhwc) STEM_FWD(f16, true); else STEM_FWD(f16, false); }
    else { if (nhwc) STEM_FWD(bf16, true); else STEM_FWD(bf16, false); }
#undef STEM_FWD
    X3D_LAUNCH_CHECK("stem_s_fwd");
    return X3D_OK;
  }
#define LAUNCH(TT, CO) \
  hipLaunchKernelGGL((stem_s_fwd_kernel<TT, 3, CO>), grid, dim3(256), 0, st, (const TT*)x, w, (TT*)y, T, H, W, Ho, Wo)
#define BY_CO(TT)                      \
  switch (Cout) {                      \
    case 8: LAUNCH(TT, 8); break;      \
    case 16: LAUNCH(TT, 16); break;    \
    case 24: LAUNCH(TT, 24); break;    \
    default: LAUNCH(TT, 32); break;    \
  }
  if (dtype == X3D_F32) { BY_CO(float) } else if (dtype == X3D_F16) { BY_CO(f16) } else { BY_CO(bf16) }
#undef BY_CO
#undef LAUNCH
  X3D_LAUNCH_CHECK("stem_s_fwd");
  return X3D_OK;
}

// ------------------------------------------------------------------------------------------------
// conv_s weight gradient on the matrix cores:  dW[co][tap] = sum_p dY[co][p] * im2col(x)[tap][p],
// tap = (ci, kh, kw) -> 27 rows padded to 32, co <= 32 rows: ONE 32x32 fp32 MFMA tile with the
// points as the K dimension.  The four waves of a workgroup split each 64-point step between them
// and the partial tiles meet in the fp32 atomics on dw.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void stem_s_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                           float* dw, int Cin, int Cout, int Tn, int H, int W,
                                                           int Ho, int Wo, int steps_per_block) {
  constexpr int BP = 64, LP = 65;
  __shared__ float As[32 * LP];  // dY  [co][p]
  __shared__ float Bs[32 * LP];  // im2col [tap][p]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, half = lane >> 5;
  const long long P = (long long)Tn * Ho * Wo;
  const int steps_per_n = (int)((P + BP - 1) / BP);
  const int chunks_per_n = (steps_per_n + steps_per_block - 1) / steps_per_block;
  const int n = blockIdx.x / chunks_per_n;
  const int chunk = blockIdx.x - n * chunks_per_n;
  const int s_begin = chunk * steps_per_block, s_end = min(s_begin + steps_per_block, steps_per_n);
  const int ntap = Cin * 9;
  f32x16 acc;
#pragma unroll
  for (int j = 0; j < 16; j++) acc[j] = 0.f;

  for (int step = s_begin; step < s_end; ++step) {
    const long long p0 = (long long)step * BP;
    __syncthreads();
    for (int v = tid; v < 32 * BP; v += 256) {
      const int row = v / BP, pp = v - row * BP;
      const long long p = p0 + pp;
      float a = 0.f, b = 0.f;
      if (p < P) {
        if (row < Cout) a = to_f<T>(dy[((long long)n * Cout + row) * P + p]);
        if (row < ntap) {
          const int ci = row / 9, k = row - ci * 9, kh = k / 3, kw = k - kh * 3;
          const long long hw = (long long)Ho * Wo;
          const long long t = p / hw;
          const int rem = (int)(p - t * hw);
          const int ho = rem / Wo, wo = rem - ho * Wo;
          const int hi = ho * 2 + kh - 1, wi = wo * 2 + kw - 1;
          if (hi >= 0 && hi < H && wi >= 0 && wi < W)
            b = to_f<T>(x[((((long long)n * Cin + ci) * Tn + t) * H + hi) * W + wi]);
        }
      }
      As[row * LP + pp] = a;
      Bs[row * LP + pp] = b;
    }
    __syncthreads();
    const float* ap = As + r * LP + wid * 16 + half;
    const float* bp = Bs + r * LP + wid * 16 + half;
#pragma unroll
    for (int kk = 0; kk < 16; kk += 2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[kk], bp[kk], acc, 0, 0, 0);
  }
  // D[row = co][col = tap]
  if (r < ntap) {
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const int co = (j & 3) + 8 * (j >> 2) + 4 * half;
      if (co < Cout) atomicAdd(&dw[co * ntap + r], acc[j]);
    }
  }
}


// ---- the same tile without the per-element index arithmetic (round 6: fp32 storage, and 16-bit rows that are not whole vectors) ----
// The generic kernel above walks the flattened points of a sample and pays two integer divisions and a bounds test per gathered
// element (548 us per X3D-S fp32 step, 3 % of BASELINE config 2).  Here a step is ONE segment of 64 consecutive output columns of
// one output row (n, t, ho): the decomposition is per step and wave-uniform, a tap row (ci, kh, kw) of the im2col tile is the
// input row 2 ho + kh - 1 at the columns 2 wo + kw - 1 -- a base pointer and a stride of two elements.  The four waves split the
// 64 points of a step between them as above (exact fp32 products on v_mfma_f32_32x32x2_f32), one atomic flush per workgroup.
template <typename T>
__global__ __launch_bounds__(256) void stem_s_wgrad_rows_kernel(const T* __restrict__ x, const T* __restrict__ dy, float* dw, int Cin,
                                                                int Cout, int Tn, int H, int W, int Ho, int Wo, int nws,
                                                                int total_segs, int segs_per_block) {
  constexpr int LP = 65;
  __shared__ float smem[2 * 32 * LP];      // one array: both tiles end as the [4][32][32] reduction buffer
  static_assert(2 * 32 * LP >= 4 * 32 * 32, "reduction buffer");
  float* const As = smem;                  // dY     [co][p]
  float* const Bs = smem + 32 * LP;        // im2col [tap][p]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, half = lane >> 5;
  const int ntap = Cin * 9;
  const int seg_begin = blockIdx.x * segs_per_block, seg_end = min(seg_begin + segs_per_block, total_segs);
  f32x16 acc;
#pragma unroll
  for (int j = 0; j < 16; j++) acc[j] = 0.f;
  // this thread's items of a step: rows row0 + 4 i (i = 0 .. 7) of both tiles, point pp
  const int pp = tid & 63, row0 = tid >> 6;
  float av[8], bv[8];
  auto load_seg = [&](int seg) {      // this thread's 8 + 8 elements of a segment (zeros outside the image / past the last segment)
    const int ws = seg % nws;
    int tmp = seg / nws;
    const int ho = tmp % Ho; tmp /= Ho;
    const int t = tmp % Tn;
    const int n = tmp / Tn;
    const int wo = ws * 64 + pp;
    const bool pin = wo < Wo && seg < seg_end;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int row = row0 + 4 * i;
      av[i] = 0.f; bv[i] = 0.f;
      if (pin && row < Cout) av[i] = to_f<T>(dy[((((long long)n * Cout + row) * Tn + t) * Ho + ho) * Wo + wo]);
      if (pin && row < ntap) {
        const int ci = row / 9, k = row - ci * 9, kh = k / 3, kw = k - kh * 3;      // (row = a loop constant + the wave index: cheap)
        const int hi = 2 * ho + kh - 1, wi = 2 * wo + kw - 1;
        if (hi >= 0 && hi < H && wi >= 0 && wi < W) bv[i] = to_f<T>(x[((((long long)n * Cin + ci) * Tn + t) * H + hi) * W + wi]);
      }
    }
  };
  if (seg_begin < seg_end) load_seg(seg_begin);
  for (int seg = seg_begin; seg < seg_end; ++seg) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; i++) {
      As[(row0 + 4 * i) * LP + pp] = av[i];
      Bs[(row0 + 4 * i) * LP + pp] = bv[i];
    }
    __syncthreads();
    load_seg(seg + 1);              // in flight behind this segment's products
    const float* ap = As + r * LP + wid * 16 + half;
    const float* bp = Bs + r * LP + wid * 16 + half;
#pragma unroll
    for (int kk = 0; kk < 16; kk += 2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[kk], bp[kk], acc, 0, 0, 0);
  }
  // D[row = co][col = tap]: the four waves' partial tiles meet in LDS, one atomic per (co, tap)
  __syncthreads();
  float* Ds = smem;              // [4][32][32]
#pragma unroll
  for (int j = 0; j < 16; j++) Ds[(wid * 32 + (j & 3) + 8 * (j >> 2) + 4 * half) * 32 + r] = acc[j];
  __syncthreads();
  for (int i = tid; i < 32 * 32; i += 256) {
    const int co = i >> 5, tap = i & 31;
    if (co < Cout && tap < ntap) atomicAdd(&dw[co * ntap + tap], Ds[i] + Ds[1024 + i] + Ds[2048 + i] + Ds[3072 + i]);
  }
}

// ---- bf16 fast path of the stem weight gradient ------------------------------------------------
// The generic kernel above gathers every im2col element with two integer divisions and a 2-byte load; it is
// instruction bound (1.39 ms on X3D-M B=64 against ~0.25 ms of HBM time).  Here a step is SEGS segments of
// 64 consecutive output columns of one output row (n, t, ho):
//   * dY rows: 16-byte loads, written as they are (MFMA A operand, k = points contiguous);
//   * x: for each (ci, kh) the input row 2ho+kh-1, columns [2wo0, 2wo0+128), as 16-byte loads; a vector of 8
//     input columns is de-interleaved on the way to LDS: even columns -> tap kw=1, odd -> kw=2 and (shifted by
//     one point) kw=0.  That IS the im2col tile [tap][point], bf16, k-contiguous -- the B operand;
//   * one v_mfma_f32_32x32x16_bf16 tile D[co][tap] per wave (wave = segment), summed across the four waves in LDS
//     and added to dW with <= Cout*Cin*9 atomics per workgroup.
template <typename HT, int SEGS, bool NHWC>
__global__ __launch_bounds__(256) void stem_s_wgrad_bf16_kernel(const HT* __restrict__ x, const HT* __restrict__ dy,
                                                                float* dw, int Cin, int Cout, int Tn, int H, int W,
                                                                int Ho, int Wo, int nws, int total_segs,
                                                                int segs_per_block) {
  typedef typename HV<HT>::x8 hx8; typedef typename HV<HT>::x4 hx4; typedef typename HV<HT>::x2 hx2;
  static_assert(SEGS == 4, "one wave per segment");
  constexpr int LP = SEGS * 64 + 8;
  __shared__ __attribute__((aligned(16))) HT As[32 * LP];  // dY     [co][point]
  __shared__ __attribute__((aligned(16))) HT Bs[32 * LP];  // im2col [tap][point]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, half = lane >> 5;
  const int ntap = Cin * 9;
  const int seg_begin = blockIdx.x * segs_per_block;      // 32-bit segment arithmetic (host check), as in the forward
  const int seg_end = min(seg_begin + segs_per_block, total_segs);
  for (int i = tid; i < 32 * LP / 8; i += 256) {
    hx8 z;
#pragma unroll
    for (int e = 0; e < 8; e++) z[e] = (HT)0.f;
    ((hx8*)As)[i] = z;
    ((hx8*)Bs)[i] = z;
  }
  f32x16 acc;
#pragma unroll
  for (int j = 0; j < 16; j++) acc[j] = 0.f;

  // staging roles: dY -- SEGS x 32 rows x 8 vectors = 4 per thread; x -- per segment 16 rows (ci,kh) x 16 vectors
  const int xrow = tid >> 4, xv = tid & 15;
  const int xci = xrow / 3, xkh = xrow - xci * 3;
  const bool xrow_ok = xrow < Cin * 3;
  hx8 rd[SEGS], rx[SEGS];
  HT rl[SEGS];
  bool okd[SEGS], okx[SEGS];
  StemNhwcStage<HT> st3;
  st3.roles(tid);
  auto issue = [&](int s0) {
    if constexpr (NHWC) st3.issue(x, s0, seg_end, nws, Ho, Tn, H, W);
#pragma unroll
    for (int q = 0; q < SEGS; q++) {
      const int seg = s0 + q;
      hx8 z;
#pragma unroll
      for (int e = 0; e < 8; e++) z[e] = (HT)0.f;
      rd[q] = z; rx[q] = z; rl[q] = (HT)0.f;
      okd[q] = false; okx[q] = false;
      if (seg >= seg_end) continue;
      const int ws = seg % nws;
      int tmp = seg / nws;
      const int ho = tmp % Ho; tmp /= Ho;
      const int t = tmp % Tn;
      const int n = tmp / Tn;
      const int wo0 = ws * 64;
      {  // dY: this thread's vector (row, v) of segment q is item tid of the segment's 256
        const int row = tid >> 3, v = tid & 7;
        if (row < Cout && wo0 + 8 * v < Wo) {
          const HT* src = dy + ((((long long)n * Cout + row) * Tn + t) * Ho + ho) * Wo + wo0 + 8 * v;
          if (wo0 + 8 * v + 8 <= Wo) rd[q] = *(const hx8*)src;
          else {   // Wo % 8 == 4: half a vector, the points past the row stay zero
            const hx4 h4 = *(const hx4*)src;
#pragma unroll
            for (int e = 0; e < 4; e++) rd[q][e] = h4[e];
          }
          okd[q] = true;
        }
      }
      const int hi = 2 * ho + xkh - 1;
      if (!NHWC && xrow_ok && hi >= 0 && hi < H && 2 * wo0 + 8 * xv < W) {
        const HT* src = x + ((((long long)n * Cin + xci) * Tn + t) * H + hi) * W + 2 * wo0 + 8 * xv;
        rx[q] = *(const hx8*)src;
        if (xv == 0 && wo0 > 0) rl[q] = src[-1];
        okx[q] = true;
      }
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int q = 0; q < SEGS; q++) {
      {
        const int row = tid >> 3, v = tid & 7;
        *(hx8*)&As[row * LP + q * 64 + 8 * v] = rd[q];   // zeros where invalid
      }
      if constexpr (NHWC) { if (q == 0) st3.commit(Bs, LP); }
      else if (xrow_ok) {
        const int tap1 = xci * 9 + xkh * 3 + 1;
        hx4 ev, od;
#pragma unroll
        for (int e = 0; e < 4; e++) { ev[e] = rx[q][2 * e]; od[e] = rx[q][2 * e + 1]; }
        *(hx4*)&Bs[tap1 * LP + q * 64 + 4 * xv] = ev;          // kw = 1: wi = 2wo
        *(hx4*)&Bs[(tap1 + 1) * LP + q * 64 + 4 * xv] = od;    // kw = 2: wi = 2wo + 1
        HT* k0 = &Bs[(tap1 - 1) * LP + q * 64 + 4 * xv + 1];    // kw = 0: wi = 2wo - 1  (one point later)
#pragma unroll
        for (int e = 0; e < 4; e++)
          if (4 * xv + 1 + e < 64) k0[e] = od[e];
        if (xv == 0) Bs[(tap1 - 1) * LP + q * 64] = rl[q];
      }
    }
  };

  if (seg_begin < seg_end) issue(seg_begin);
  for (int s0 = seg_begin; s0 < seg_end; s0 += SEGS) {
    __syncthreads();
    commit();
    __syncthreads();
    if (s0 + SEGS < seg_end) issue(s0 + SEGS);
    const HT* ap = As + r * LP + wid * 64 + 8 * half;
    const HT* bp = Bs + r * LP + wid * 64 + 8 * half;
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      const hx8 af = *(const hx8*)(ap + ks * 16);
      const hx8 bf = *(const hx8*)(bp + ks * 16);
      acc = mfma16<HT>(af, bf, acc);
    }
  }
  // sum the four waves' partial tiles in LDS, then one atomic per (co, tap)
  __syncthreads();
  float* Ds = (float*)As;   // [4][32][32] fp32 = 16 KB <= sizeof(As)
#pragma unroll
  for (int j = 0; j < 16; j++) Ds[(wid * 32 + (j & 3) + 8 * (j >> 2) + 4 * half) * 32 + r] = acc[j];
  __syncthreads();
  for (int i = tid; i < 32 * 32; i += 256) {
    const int co = i >> 5, tap = i & 31;
    if (co < Cout && tap < ntap)
      atomicAdd(&dw[co * ntap + tap], Ds[i] + Ds[1024 + i] + Ds[2048 + i] + Ds[3072 + i]);
  }
}

extern "C" int x3d_stem_s_wgrad(const void* x, const void* dy, float* dw, int N, int Cin, int T, int H, int W,
                                int Cout, int dtype, int x_layout, void* stream) {
  X3D_REQUIRE(x && dy && dw && N > 0 && T > 0 && H > 0 && W > 0, "stem_s_wgrad: bad args");
  X3D_REQUIRE(Cin * 9 <= 32 && Cout <= 32, "stem_s_wgrad: needs Cin*9 <= 32 and Cout <= 32");
  X3D_REQUIRE(x3d_dtype_ok(dtype), "stem_s_wgrad: bad dtype");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const long long P = (long long)T * Ho * Wo;
  const long long steps_per_n = ceil_div_ll(P, 64);
  int spb = (int)(steps_per_n * N / 2048);
  if (spb < 2) spb = 2;
  if (spb > 64) spb = 64;
  if (spb > steps_per_n) spb = (int)steps_per_n;
  const long long gx = ceil_div_ll(steps_per_n, spb) * N;
  hipStream_t st = (hipStream_t)stream;
  X3D_REQUIRE(x_layout == X3D_LAYOUT_NCTHW || x_layout == X3D_LAYOUT_NTHWC, "stem_s_wgrad: bad x_layout");
  const bool nhwc = x_layout == X3D_LAYOUT_NTHWC;
  X3D_REQUIRE(!nhwc || (x3d_stem_s_nthwc_supported(Cin, W, Cout, dtype) && ((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 16) == 0),
              "stem_s_wgrad: channels-last x needs 16-bit storage, W %% 8 == 0 and 16-byte aligned tensors (x3d_stem_s_nthwc_supported)");
  if (x3d_is_half(dtype) && (W % 8) == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 16) == 0) {
    // fast path: rows of x are whole 16-byte vectors, rows of dY whole 8-byte ones (W % 8 == 0 -> Wo % 4 == 0)
    constexpr int SEGS = 4;
    const int nws = ceil_div(Wo, 64);
    const long long total_segs = (long long)N * T * Ho * nws;
    static int slots = 0;
    if (slots == 0) {
      int nb = 0, dev = 0, cus = 256;
      hipDeviceProp_t prop;
      if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, stem_s_wgrad_bf16_kernel<bf16, SEGS, false>, 256, 0) != hipSuccess || nb < 1) nb = 2;
      slots = nb * cus;
    }
    long long spb2 = ceil_div_ll(total_segs, slots);
    if (spb2 < 8 * SEGS) spb2 = 8 * SEGS;
    spb2 = ceil_div_ll(spb2, SEGS) * SEGS;
    const long long gx2 = ceil_div_ll(total_segs, spb2);
    X3D_REQUIRE(gx2 < (1ll << 31), "stem_s_wgrad: grid too large");
    X3D_REQUIRE(total_segs + spb2 < (1ll << 31), "stem_s_wgrad: too many row segments");
#define STEM_WG(TT, NHWC_) hipLaunchKernelGGL((stem_s_wgrad_bf16_kernel<TT, SEGS, NHWC_>), dim3((unsigned)gx2), dim3(256), 0, st, (const TT*)x, \
                                              (const TT*)dy, dw, Cin, Cout, T, H, W, Ho, Wo, nws, (int)total_segs, (int)spb2)
    if (dtype == X3D_F16) { if (nhwc) STEM_WG(f16, true); else STEM_WG(f16, false); }
    else { if (nhwc) STEM_WG(bf16, true); else STEM_WG(bf16, false); }
#undef STEM_WG
    X3D_LAUNCH_CHECK("stem_s_wgrad");
    return X3D_OK;
  }
  {
    // row segments (round 6): 2-3 workgroups per CU, an equal share of the segments each
    const int nws = ceil_div(Wo, 64);
    const long long total_segs = (long long)N * T * Ho * nws;
    if (total_segs < (1ll << 31) && x3d_env_int("X3D_STEM_WGRAD_ROWS", 1) != 0) {      // X3D_STEM_WGRAD_ROWS=0: A/B hook (the flattened-point kernel)
      long long spb2 = ceil_div_ll(total_segs, 3ll * x3d_device_cus());
      if (spb2 < 8) spb2 = 8;
      const long long gx2 = ceil_div_ll(total_segs, spb2);
#define STEM_WGR(TT) hipLaunchKernelGGL((stem_s_wgrad_rows_kernel<TT>), dim3((unsigned)gx2), dim3(256), 0, st, (const TT*)x, (const TT*)dy, dw, \
                                        Cin, Cout, T, H, W, Ho, Wo, nws, (int)total_segs, (int)spb2)
      if (dtype == X3D_F32) STEM_WGR(float); else if (dtype == X3D_F16) STEM_WGR(f16); else STEM_WGR(bf16);
#undef STEM_WGR
      X3D_LAUNCH_CHECK("stem_s_wgrad");
      return X3D_OK;
    }
  }
  if (dtype == X3D_F32)
    hipLaunchKernelGGL((stem_s_wgrad_kernel<float>), dim3((unsigned)gx), dim3(256), 0, st, (const float*)x,
                       (const float*)dy, dw, Cin, Cout, T, H, W, Ho, Wo, spb);
  else if (dtype == X3D_F16)
    hipLaunchKernelGGL((stem_s_wgrad_kernel<f16>), dim3((unsigned)gx), dim3(256), 0, st, (const f16*)x,
                       (const f16*)dy, dw, Cin, Cout, T, H, W, Ho, Wo, spb);
  else
    hipLaunchKernelGGL((stem_s_wgrad_kernel<bf16>), dim3((unsigned)gx), dim3(256), 0, st, (const bf16*)x,
                       (const bf16*)dy, dw, Cin, Cout, T, H, W, Ho, Wo, spb);
  X3D_LAUNCH_CHECK("stem_s_wgrad");
  return X3D_OK;
}

// ------------------------------------------------------------------------------------------------
// conv_t: temporal depthwise conv, KT taps (KT odd, <= 7).  A thread owns VEC consecutive spatial
// positions of one (n, c) plane stack and walks T with a rolling register window, so each element
// is read once and written once.
// ------------------------------------------------------------------------------------------------
#define DWT_MAXK 7

template <typename T, int VEC, int KT>
__global__ __launch_bounds__(256) void dwt_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                      T* __restrict__ y, double* stats, const float* __restrict__ oss, int oact,
                                                      int C, int Tn, long long HW) {
  __shared__ float scratch[2 * 4];
  const int nc = blockIdx.y, c = nc % C;
  const long long q = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * VEC;
  constexpr int R = KT / 2;
  float wk[KT];
#pragma unroll
  for (int k = 0; k < KT; k++) wk[k] = w[c * KT + k];
  float red[2] = {0.f, 0.f};
  // inference epilogue: the stem's BatchNorm (moving statistics) + ReLU on the accumulator -- the raw conv_t output and
  // the separate BN + ReLU pass over the widest tensor of the network never touch HBM
  const float os = oss ? oss[c * 2] : 1.0f, ot = oss ? oss[c * 2 + 1] : 0.f;
  const float olo = (oss && oact == X3D_ACT_RELU) ? 0.f : -__builtin_inff();
  if (q < HW) {
    const T* xp = x + (long long)nc * Tn * HW + q;
    T* yp = y + (long long)nc * Tn * HW + q;
    float win[KT][VEC];  // win[k] = x[t + k - R]
#pragma unroll
    for (int k = 0; k < KT; k++)
#pragma unroll
      for (int e = 0; e < VEC; e++) win[k][e] = 0.f;
    // preload x[0 .. R-1] into win[R+1 .. KT-1] (they become win[R..] after the first shift)
#pragma unroll
    for (int k = 0; k < R; k++)
      if (k < Tn) VecIO<T, VEC>::load(xp + (long long)k * HW, win[R + 1 + k]);
    for (int t = 0; t < Tn; t++) {
#pragma unroll
      for (int k = 0; k < KT - 1; k++)
#pragma unroll
        for (int e = 0; e < VEC; e++) win[k][e] = win[k + 1][e];
      if (t + R < Tn) {
        VecIO<T, VEC>::load(xp + (long long)(t + R) * HW, win[KT - 1]);
      } else {
#pragma unroll
        for (int e = 0; e < VEC; e++) win[KT - 1][e] = 0.f;
      }
      float o[VEC];
#pragma unroll
      for (int e = 0; e < VEC; e++) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < KT; k++) acc += wk[k] * win[k][e];
        if (oss) acc = fmaxf(os * acc + ot, olo);
        o[e] = acc;
        const float vr = round_to<T>(acc);
        red[0] += vr;
        red[1] += vr * vr;
      }
      VecIO<T, VEC>::store(yp + (long long)t * HW, o);
    }
  }
  if (stats) {
    block_sum<2>(red, scratch);
    if (threadIdx.x == 0) {
      double* sp = stats_replica(stats, C, blockIdx.x + (unsigned)(nc / C));
      atomic_add_d(&sp[c * 2], (double)red[0]);
      atomic_add_d(&sp[c * 2 + 1], (double)red[1]);
    }
  }
}

// backward: dY[t] = A*g[t] + B*yraw[t] + C ; dx[t] = sum_k w[k]*dY[t + R - k] ; dw[k] += sum dY[t]*x[t+k-R]
// The three planes of step tau+1 are loaded while step tau is computed.  Every global access of the loop is an
// unconditional bounds-checked buffer instruction (planes past T / steps before the first complete dx: out of range,
// zero / dropped): the store then never sits behind a conservative vmcnt(0) in front of the prefetched planes
// (vmcnt retires in order and conditional memory operations cannot be counted).
typedef __attribute__((ext_vector_type(4))) unsigned int stem_u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int stem_u32x2;
template <typename T, int VEC> struct BufVec {
  static constexpr int BYTES = VEC * (int)sizeof(T);
  static_assert(BYTES == 2 || BYTES == 4 || BYTES == 8 || BYTES == 16, "buffer vector of 2 / 4 / 8 / 16 bytes");
  static constexpr int NW = BYTES >= 4 ? BYTES / 4 : 1;
  unsigned int w[NW];
  __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rs, int off) {
    if constexpr (BYTES == 16) { const stem_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0); w[0] = v[0]; w[1] = v[1]; w[2] = v[2]; w[3] = v[3]; }
    else if constexpr (BYTES == 8) { const stem_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 0); w[0] = v[0]; w[1] = v[1]; }
    else if constexpr (BYTES == 4) w[0] = __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0);
    else w[0] = __builtin_amdgcn_raw_buffer_load_b16(rs, off, 0, 0);
  }
  __device__ __forceinline__ void store(__amdgpu_buffer_rsrc_t rs, int off) const {
    if constexpr (BYTES == 16) { const stem_u32x4 v = {w[0], w[1], w[2], w[3]}; __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 0); }
    else if constexpr (BYTES == 8) { const stem_u32x2 v = {w[0], w[1]}; __builtin_amdgcn_raw_buffer_store_b64(v, rs, off, 0, 0); }
    else if constexpr (BYTES == 4) __builtin_amdgcn_raw_buffer_store_b32(w[0], rs, off, 0, 0);
    else __builtin_amdgcn_raw_buffer_store_b16((unsigned short)w[0], rs, off, 0, 0);
  }
  __device__ __forceinline__ float get(int e) const {
    if constexpr (sizeof(T) == 4) return __uint_as_float(w[e]);
    else if constexpr (__is_same(T, bf16)) return __uint_as_float(((w[e >> 1] >> (16 * (e & 1))) & 0xffffu) << 16);
    else return (float)__builtin_bit_cast(T, (unsigned short)(w[e >> 1] >> (16 * (e & 1))));
  }
  __device__ __forceinline__ void set(const float (&v)[VEC]) {
    if constexpr (sizeof(T) == 4) {
#pragma unroll
      for (int e = 0; e < VEC; e++) w[e] = __float_as_uint(v[e]);
    } else if constexpr (VEC == 1) {
      w[0] = (unsigned int)__builtin_bit_cast(unsigned short, (T)v[0]);
    } else {
#pragma unroll
      for (int e = 0; e < VEC / 2; e++)
        w[e] = (unsigned int)__builtin_bit_cast(unsigned short, (T)v[2 * e]) |
               ((unsigned int)__builtin_bit_cast(unsigned short, (T)v[2 * e + 1]) << 16);
    }
  }
};

template <typename T, int VEC, int KT>
__global__ __launch_bounds__(256) void dwt_bwd_kernel(const T* __restrict__ g, const T* __restrict__ yraw,
                                                      const float* __restrict__ rss,
                                                      const float* __restrict__ coef, const T* __restrict__ x,
                                                      const float* __restrict__ w, T* __restrict__ dx, float* dw,
                                                      int C, int Tn, long long HW) {
  __shared__ float scratch[KT * 4];
  const int nc = blockIdx.y, c = nc % C;
  const long long q = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * VEC;
  constexpr int R = KT / 2;
  constexpr int OOB = 0x40000000;
  float wk[KT], dwk[KT];
#pragma unroll
  for (int k = 0; k < KT; k++) { wk[k] = w[c * KT + k]; dwk[k] = 0.f; }
  const float A = coef[c * 4], B = coef[c * 4 + 1], Cc = coef[c * 4 + 2];
  // rss: g is the unmasked gradient, the ReLU mask [ms*yraw + mt > 0] is applied here (ms = 0, mt = 1: always on)
  const float ms = rss ? rss[c * 2] : 0.f, mt = rss ? rss[c * 2 + 1] : 1.f;
  // one (n, c) slab of T planes per resource: < 2^30 bytes (host check)
  const long long slab = (long long)nc * Tn * HW;
  const int slab_bytes = (int)(Tn * HW * (long long)sizeof(T)), plane_bytes = (int)(HW * (long long)sizeof(T));
  const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((T*)g + slab, 0, slab_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((T*)yraw + slab, 0, slab_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((T*)x + slab, 0, slab_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(dx + slab, 0, slab_bytes, 0x00020000);
  const bool live = q < HW;
  const int qoff = live ? (int)(q * (long long)sizeof(T)) : OOB;
  // step tau brings in dY[tau] and x[tau]; then dx[tau-R] = sum_k w[k]*dY[tau-k] is complete and
  // dY[tau-R] meets its whole x window x[tau-2R .. tau].
  float dwin[KT][VEC], xwin[KT][VEC];  // dwin[k] = dY[tau - (KT-1) + k], same for xwin
#pragma unroll
  for (int k = 0; k < KT; k++)
#pragma unroll
    for (int e = 0; e < VEC; e++) { dwin[k][e] = 0.f; xwin[k][e] = 0.f; }
  BufVec<T, VEC> bg, by, bx;
  bg.load(rg, qoff); by.load(ry, qoff); bx.load(rx, qoff);
  {  // a dropped store behind the first prefetch, as in every iteration of the loop
    BufVec<T, VEC> z;
#pragma unroll
    for (int i = 0; i < BufVec<T, VEC>::NW; i++) z.w[i] = 0u;
    z.store(rd, OOB);
  }
  for (int tau = 0; tau < Tn + R; tau++) {
#pragma unroll
    for (int k = 0; k < KT - 1; k++)
#pragma unroll
      for (int e = 0; e < VEC; e++) { dwin[k][e] = dwin[k + 1][e]; xwin[k][e] = xwin[k + 1][e]; }
    const bool in = live && tau < Tn;   // past T the loads returned zeros: dY must be 0 there, not C
#pragma unroll
    for (int e = 0; e < VEC; e++) {
      const float yv = by.get(e);
      const float gv = (ms * yv + mt > 0.f) ? bg.get(e) : 0.f;
      dwin[KT - 1][e] = in ? A * gv + B * yv + Cc : 0.f;
      xwin[KT - 1][e] = bx.get(e);
    }
    const int noff = qoff + (tau + 1) * plane_bytes;   // tau + 1 >= Tn: past the slab -> zeros, no traffic
    bg.load(rg, noff); by.load(ry, noff); bx.load(rx, noff);
    const int t = tau - R;
    float o[VEC];
#pragma unroll
    for (int e = 0; e < VEC; e++) {
      float acc = 0.f;
      // dx[t] = sum_k w[k]*dY[t+R-k] = sum_k w[k]*dwin[KT-1-k]
#pragma unroll
      for (int k = 0; k < KT; k++) acc += wk[k] * dwin[KT - 1 - k][e];
      o[e] = acc;
      // dY[t] = dwin[KT-1-R]; x[t+k-R] = xwin[KT-1-2R+k] = xwin[k]   (t < 0: dwin[KT-1-R] is still zero)
#pragma unroll
      for (int k = 0; k < KT; k++) dwk[k] += dwin[KT - 1 - R][e] * xwin[k][e];
    }
    BufVec<T, VEC> ob;
    ob.set(o);
    ob.store(rd, t >= 0 ? qoff + t * plane_bytes : OOB);
  }
  block_sum<KT>(dwk, scratch);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < KT; k++) atomicAdd(&dw[c * KT + k], dwk[k]);
  }
}

template <typename T, int VEC>
static int dwt_fwd_kt(const void* x, const float* w, void* y, double* stats, const float* oss, int oact, int NC, int C, int T_,
                      long long HW, int KT, hipStream_t st) {
  dim3 grid((unsigned)ceil_div_ll(HW, 256ll * VEC), (unsigned)NC);
  switch (KT) {
    case 1: hipLaunchKernelGGL((dwt_fwd_kernel<T, VEC, 1>), grid, dim3(256), 0, st, (const T*)x, w, (T*)y, stats, oss, oact, C, T_, HW); break;
    case 3: hipLaunchKernelGGL((dwt_fwd_kernel<T, VEC, 3>), grid, dim3(256), 0, st, (const T*)x, w, (T*)y, stats, oss, oact, C, T_, HW); break;
    case 5: hipLaunchKernelGGL((dwt_fwd_kernel<T, VEC, 5>), grid, dim3(256), 0, st, (const T*)x, w, (T*)y, stats, oss, oact, C, T_, HW); break;
    case 7: hipLaunchKernelGGL((dwt_fwd_kernel<T, VEC, 7>), grid, dim3(256), 0, st, (const T*)x, w, (T*)y, stats, oss, oact, C, T_, HW); break;
    default: x3d_set_error("dwt_fwd: KT must be 1,3,5 or 7"); return X3D_ERR_INVALID;
  }
  X3D_LAUNCH_CHECK("dwt_fwd");
  return X3D_OK;
}

extern "C" int x3d_dwt_fwd(const void* x, const float* w, void* y, double* stats, const float* oss, int oact, int N, int C, int T,
                           int HW, int KT, int dtype, void* stream) {
  X3D_REQUIRE(x && w && y && N > 0 && C > 0 && T > 0 && HW > 0, "dwt_fwd: bad args");
  X3D_REQUIRE(!(oss && stats), "dwt_fwd: the inference epilogue (out_scale_shift) takes no statistics");
  X3D_REQUIRE(oact == X3D_ACT_NONE || oact == X3D_ACT_RELU, "dwt_fwd: out_act must be none or ReLU");
  X3D_REQUIRE(x3d_dtype_ok(dtype), "dwt_fwd: bad dtype");
  hipStream_t st = (hipStream_t)stream;
  const int eb = dtype == X3D_F32 ? 4 : 2;
  const int vec = pick_vec(eb, HW, x, y);
  if (dtype == X3D_F32)
    return vec >= 4 ? dwt_fwd_kt<float, 4>(x, w, y, stats, oss, oact, N * C, C, T, HW, KT, st)
                    : dwt_fwd_kt<float, 1>(x, w, y, stats, oss, oact, N * C, C, T, HW, KT, st);
  if (dtype == X3D_F16)
    return vec >= 4 ? dwt_fwd_kt<f16, 4>(x, w, y, stats, oss, oact, N * C, C, T, HW, KT, st)
                    : dwt_fwd_kt<f16, 1>(x, w, y, stats, oss, oact, N * C, C, T, HW, KT, st);
  return vec >= 4 ? dwt_fwd_kt<bf16, 4>(x, w, y, stats, oss, oact, N * C, C, T, HW, KT, st)
                  : dwt_fwd_kt<bf16, 1>(x, w, y, stats, oss, oact, N * C, C, T, HW, KT, st);
}

template <typename T, int VEC>
static int dwt_bwd_kt(const void* g, const void* yraw, const float* rss, const float* coef, const void* x, const float* w, void* dx,
                      float* dw, int NC, int C, int T_, long long HW, int KT, hipStream_t st) {
  dim3 grid((unsigned)ceil_div_ll(HW, 256ll * VEC), (unsigned)NC);
#define L(K) hipLaunchKernelGGL((dwt_bwd_kernel<T, VEC, K>), grid, dim3(256), 0, st, (const T*)g, (const T*)yraw, rss, coef, (const T*)x, w, (T*)dx, dw, C, T_, HW)
  switch (KT) {
    case 1: L(1); break;
    case 3: L(3); break;
    case 5: L(5); break;
    case 7: L(7); break;
    default: x3d_set_error("dwt_bwd: KT must be 1,3,5 or 7"); return X3D_ERR_INVALID;
  }
#undef L
  X3D_LAUNCH_CHECK("dwt_bwd");
  return X3D_OK;
}

extern "C" int x3d_dwt_bwd(const void* g, const void* yraw, const float* rss, const float* coef, const void* x,
                           const float* w, void* dx, float* dw, int N, int C, int T, int HW, int KT, int dtype,
                           void* stream) {
  X3D_REQUIRE(g && yraw && coef && x && w && dx && dw && N > 0 && C > 0 && T > 0 && HW > 0, "dwt_bwd: bad args");
  X3D_REQUIRE(x3d_dtype_ok(dtype), "dwt_bwd: bad dtype");
  hipStream_t st = (hipStream_t)stream;
  const int eb = dtype == X3D_F32 ? 4 : 2;
  X3D_REQUIRE((long long)T * HW * eb < (1ll << 30), "dwt_bwd: one channel slab exceeds the 1 GB buffer window");
  const int vec = pick_vec(eb, HW, g, yraw, x, dx);
  const int want = x3d_env_int("X3D_DWT_BWD_VEC", 4);   // experiment hook: elements per thread (bf16: 2 or 4)
  if (dtype == X3D_F32)
    return vec >= 2 ? dwt_bwd_kt<float, 2>(g, yraw, rss, coef, x, w, dx, dw, N * C, C, T, HW, KT, st)
                    : dwt_bwd_kt<float, 1>(g, yraw, rss, coef, x, w, dx, dw, N * C, C, T, HW, KT, st);
  if (dtype == X3D_F16) {
    if (vec >= 8 && want >= 8) return dwt_bwd_kt<f16, 8>(g, yraw, rss, coef, x, w, dx, dw, N * C, C, T, HW, KT, st);
    if (vec >= 4 && want >= 4) return dwt_bwd_kt<f16, 4>(g, yraw, rss, coef, x, w, dx, dw, N * C, C, T, HW, KT, st);
    return vec >= 2 ? dwt_bwd_kt<f16, 2>(g, yraw, rss, coef, x, w, dx, dw, N * C, C, T, HW, KT, st)
                    : dwt_bwd_kt<f16, 1>(g, yraw, rss, coef, x, w, dx, dw, N * C, C, T, HW, KT, st);
  }
  if (vec >= 8 && want >= 8) return dwt_bwd_kt<bf16, 8>(g, yraw, rss, coef, x, w, dx, dw, N * C, C, T, HW, KT, st);
  if (vec >= 4 && want >= 4) return dwt_bwd_kt<bf16, 4>(g, yraw, rss, coef, x, w, dx, dw, N * C, C, T, HW, KT, st);
  return vec >= 2 ? dwt_bwd_kt<bf16, 2>(g, yraw, rss, coef, x, w, dx, dw, N * C, C, T, HW, KT, st)
                  : dwt_bwd_kt<bf16, 1>(g, yraw, rss, coef, x, w, dx, dw, N * C, C, T, HW, KT, st);
}
