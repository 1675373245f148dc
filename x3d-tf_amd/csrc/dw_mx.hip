// Channelwise 3x3x3 convolution on the MATRIX CORES, 14x14 stride-1 planes, 16-bit storage (X3D-M stage 4).
//
// Why.  The stride-1 depthwise kernels are bound by VALU issue, not by HBM: 27 (forward) / 54 (fused
// backward) multiply-adds per output plus the staging / window / emit instructions -- ~150 / ~350 vector instructions per
// thread and plane of four outputs, at 3-5 clocks each (tools/micro/vgpr_banks.hip).  The matrix cores take the multiply-adds
// off the vector pipe.  Along one image row the convolution is a product with a banded Toeplitz matrix: for a fixed tap row
// (kt, kh)
//     out[h][w] += sum_k  Wt[w][k] * in[t + kt - 1][h + kh - 1][k],     Wt[w][k] = w[kt][kh][k - w + 1]  (0 <= k - w + 1 <= 2)
// which is v_mfma_f32_16x16x32_{bf16,f16} with  A = Wt (M = output column w, K = input column k),  B[k][n] = the input row
// of output row n = h,  D[w][h] = out^T.  A 14-wide row has 16 window columns (k = w: the left zero pad is the missing k = -1,
// the right one the zero columns 14, 15), so the K = 32 of one MFMA holds TWO tap rows (lane groups 0-1 and 2-3 read
// different rows): the 9 tap rows (kt, kh) of an output plane are 6 MFMAs (per kt: rows kh = 0 | 1 together, kh = 2 with a
// zero half) = 96 matrix-core clocks per (n, c, t) plane against ~500 vector clocks of the tap loop they replace.
//   * A (the weights) depends on the channel only: six operands (24 VGPRs) built once per wave.
//   * B is read from an LDS image of the plane in the storage type (BN_a + ReLU applied in fp32 at staging, then rounded --
//     what the pointwise matrix-core kernels do with their operands, and what the reference's mixed-precision policy does with
//     the whole convolution): rows -1 .. 16 at a pitch of 48 bytes -- the 16 lanes of a ds_read_b128 quarter cover 64 distinct
//     banks -- three planes in a ring.  Lane (h = lane & 15, g = lane >> 4) reads 8 columns 8 (g & 1) .. of row h + kh.
//   * D: lane (h, g) holds out[h][4 g .. 4 g + 3] -- the strip of four outputs it also LOADED (one 8-byte load per lane and
//     plane, one LDS write) and stores (two 4-byte halves: the last strip of a 14-wide row is half empty).
//   * ONE wave per workgroup, one (n, c) channel per wave (dw_pk.hip: waves that share barriers take turns on the pipes; here
//     there is no barrier at all -- a wave's LDS accesses execute in order), ~60 VGPRs: eight waves per SIMD.
// Sums are fp32 in the matrix cores' own order: results agree with the vector kernels to the operand rounding (tests:
// test_dw3d_fwd compares 16-bit storage with the fp64 convolution of the ROUNDED operands).
#include "dw_common.h"

#define MX_PITCH 48                     // bytes per LDS row: 16 columns + 8 of padding (conflict-free b128 rows)
#define MX_ROWS 18                      // image rows -1 .. 16 (outputs rows 14, 15 of the 16-row tile are never stored)
#define MX_TILE (MX_PITCH * MX_ROWS)    // 864 bytes per plane

struct DwMxFwdArgs {
  DwFwdArgs f;
  unsigned bytes;    // whole-tensor size (buffer num_records)
};

template <typename T> struct MxOp;
template <> struct MxOp<bf16> {
  typedef bf16x8 x8;
  static __device__ __forceinline__ f32x4 mfma(x8 a, x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct MxOp<f16> {
  typedef f16x8 x8;
  static __device__ __forceinline__ f32x4 mfma(x8 a, x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};

// The six weight operands of one channel.  Operand i (0..5): kt = i / 2; even i: lane groups 0-1 hold tap row kh = 0, groups
// 2-3 tap row kh = 1; odd i: groups 0-1 hold kh = 2, groups 2-3 zeros.  FLIP (data gradient): the transposed / reversed
// kernel, A[w][k] = w[2 - kt][2 - kh][w - k + 1].
template <typename T, bool FLIP>
__device__ __forceinline__ void mx_weight_operands(const float* __restrict__ wc, int lane, typename MxOp<T>::x8 (&A)[6]) {
  const int w = lane & 15, g = lane >> 4;
  float wr[27];
#pragma unroll
  for (int k = 0; k < 27; k++) wr[k] = wc[k];
#pragma unroll
  for (int i = 0; i < 6; i++) {
    const int kt = i >> 1;
    // the tap row of this lane group: kh = 0 | 1 (even i), 2 | none (odd i)
    float t3[3];
#pragma unroll
    for (int kw = 0; kw < 3; kw++) {
      float lo, hi;
      if (!FLIP) {
        lo = (i & 1) ? wr[kt * 9 + 6 + kw] : wr[kt * 9 + kw];
        hi = (i & 1) ? 0.f : wr[kt * 9 + 3 + kw];
      } else {   // tap (kt, kh, kw) of the flipped kernel = w[2 - kt][2 - kh][2 - kw]
        lo = (i & 1) ? wr[(2 - kt) * 9 + 0 + (2 - kw)] : wr[(2 - kt) * 9 + 6 + (2 - kw)];
        hi = (i & 1) ? 0.f : wr[(2 - kt) * 9 + 3 + (2 - kw)];
      }
      t3[kw] = (g >> 1) ? hi : lo;
    }
    typename MxOp<T>::x8 a;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int kw = 8 * (g & 1) + j - w + 1;     // k - w + 1
      const float v = kw == 0 ? t3[0] : (kw == 1 ? t3[1] : (kw == 2 ? t3[2] : 0.f));
      a[j] = (T)v;
    }
    A[i] = a;
  }
}

// Which image strip a lane owns.  Lane (h = lane & 15, s = lane >> 4) is row h, columns 4 s .. 4 s + 3 of the 16 x 16 tile.
//   P7 = false: ONE plane of H x W <= 14 x 14 (even W) -- tile row / column = image row / column, rows >= H and columns >= W
//     hold zeros (the bottom / right TF-SAME pad).
//   P7 = true: FOUR 7 x 7 planes (samples n0 .. n0 + 3 of one channel) in one tile: rows 0-6 | zero row 7 | rows 8-14, columns
//     0-6 | zero column 7 | columns 8-14 -- the separators are the planes' zero pads, so the products are those of the single
//     plane.  Rows of 7 elements start at any 2-byte address: the second strip of a row (columns 4, 5, 6) is loaded one
//     element early (columns 3 .. 6: it never leaves its row, so never the tensor) and shifted down; its third element is
//     stored alone.
template <bool P7>
struct MxMap {
  int voffL, shift;          // load offset (bytes from the tensor base; DW_OOB: nothing), elements to shift the loaded vector down
  int vst0, vst1, vst1e;     // stores: pair 0, pair 1 (4 bytes each), element 2 alone (2 bytes)
  unsigned mk0, mk1;         // valid elements of the two pairs as AND masks
  int n;                     // sample of this lane
  int planeB;
  __device__ __forceinline__ void build(int lane, int blk, int N, int C, int T, int H, int W, int& c) {
    const int h = lane & 15, s = lane >> 4;
    if constexpr (!P7) {
      c = __builtin_amdgcn_readfirstlane(blk % C);
      n = __builtin_amdgcn_readfirstlane(blk / C);
      const bool ok0 = h < H && 4 * s < W, ok1 = h < H && 4 * s + 2 < W;
      const long long e = (((long long)n * C + c) * T * H + h) * W + 4 * s;
      voffL = ok0 ? (int)(e * 2) : DW_OOB;
      shift = 0;
      vst0 = voffL;
      vst1 = ok1 ? (int)(e * 2) + 4 : DW_OOB;
      vst1e = DW_OOB;
      mk0 = ok0 ? 0xffffffffu : 0u;
      mk1 = ok1 ? 0xffffffffu : 0u;
      planeB = H * W * 2;
    } else {
      c = __builtin_amdgcn_readfirstlane(blk % C);
      const int n0 = __builtin_amdgcn_readfirstlane(blk / C) * 4;
      n = n0 + 2 * (h >> 3) + (s >> 1);
      const int r = h & 7, second = s & 1;                 // in-plane row, second strip of the row (columns 4, 5, 6)
      const bool ok = r < 7 && n < N;
      const long long e = (((long long)n * C + c) * T * 7 + r) * 7 + 4 * second;
      voffL = ok ? (int)((e - second) * 2) : DW_OOB;
      shift = second;
      vst0 = ok ? (int)(e * 2) : DW_OOB;
      vst1 = (ok && !second) ? (int)(e * 2) + 4 : DW_OOB;
      vst1e = (ok && second) ? (int)(e * 2) + 4 : DW_OOB;
      mk0 = ok ? 0xffffffffu : 0u;
      mk1 = ok ? (second ? 0x0000ffffu : 0xffffffffu) : 0u;
      planeB = 49 * 2;
    }
  }
  // the loaded vector with the strip's first element in element 0
  __device__ __forceinline__ Raw aligned(const Raw& r) const {
    if constexpr (P7) {
      const unsigned long long v = (((unsigned long long)r.w[1] << 32) | r.w[0]) >> (16 * shift);
      Raw o; o.w[0] = (unsigned)v; o.w[1] = (unsigned)(v >> 32); o.w[2] = o.w[3] = 0u;
      return o;
    } else {
      return r;
    }
  }
  __device__ __forceinline__ void store(__amdgpu_buffer_rsrc_t rs, unsigned p0, unsigned p1, int soff) const {
    Raw o, o1;
    o.w[0] = p0; o1.w[0] = p1;
    raw_bstore<4>(o, rs, vst0, soff);
    raw_bstore<4>(o1, rs, vst1, soff);
    if constexpr (P7) raw_bstore<2>(o1, rs, vst1e, soff);
  }
};

// RB: planes of the LDS ring (3, or 4 where T % 4 == 0 lets the unrolled loop go without an exit test: UN = 4);
// EXACT: T % UN == 0 -- no `break` inside the unrolled body (with it the compiler's counts of outstanding accesses merge
// over the exit paths and some waits of the loop fall back to small vmcnt values: the stores' latency shows)
template <typename T, int UN, int RB, int PD, bool EXACT, bool P7>
__global__ __launch_bounds__(64, 7) void dw3d_fwd_mx14_kernel(const DwMxFwdArgs pa) {
  static_assert(UN % RB == 0 && UN % PD == 0, "ring period RB, slots period PD");
  typedef typename MxOp<T>::x8 x8;
  __shared__ __attribute__((aligned(16))) unsigned char lds[RB * MX_TILE];
  const DwFwdArgs& a = pa.f;
  const DwGeom& g = a.g;
  const int lane = threadIdx.x;
  const int h = lane & 15, s = lane >> 4;
  int c;
  MxMap<P7> mp;
  mp.build(lane, blockIdx.x, g.N, g.C, g.T, g.H, g.W, c);

  for (int i = lane; i < RB * MX_TILE / 16; i += 64) ((uint4*)lds)[i] = make_uint4(0u, 0u, 0u, 0u);   // pads, halo rows, plane -1

  x8 A[6];
  mx_weight_operands<T, false>(a.w + c * 27, lane, A);
  float sc = 1.f, sh = 0.f;
  if (a.ss) { sc = a.ss[c * 2]; sh = a.ss[c * 2 + 1]; }
  const float lo = a.act == X3D_ACT_RELU ? 0.f : -__builtin_inff();

  const int planeB = mp.planeB;
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((T*)a.x, 0, pa.bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc((T*)a.y, 0, pa.bytes, 0x00020000);
  // LDS: the own strip is staged at row h + 1, columns 4 s ..; operand reads start at row h (+ kh), columns 8 (s & 1)
  unsigned char* stg = lds + (h + 1) * MX_PITCH + s * 8;
  const unsigned char* rd01 = lds + (h + (s >> 1)) * MX_PITCH + (s & 1) * 16;   // kh = 0 | 1
  const unsigned char* rd2 = lds + (h + 2) * MX_PITCH + (s & 1) * 16;           // kh = 2 | (zero weights: any finite row)

  Raw slot[PD];             // planes in flight; plane p travels in slot p % PD
  auto issue = [&](int t, Raw& r) { raw_bload<8>(r, rsX, mp.voffL, t < g.T ? t * planeB : DW_OOB); };
  const unsigned mk0 = mp.mk0, mk1 = mp.mk1;   // valid elements of the strip's two pairs
  auto stage = [&](const Raw& r0, int q, bool plane_ok) {   // plane -> ring slot q (a plane past T: zeros)
    const Raw r = mp.aligned(r0);
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; e++) v[e] = fmaxf(__builtin_fmaf(sc, raw_get<T>(r, e), sh), lo);
    const unsigned pm = plane_ok ? 0xffffffffu : 0u;        // (uniform)
    *(uint2*)(stg + q * MX_TILE) = make_uint2(Dot2<T>::pk(v[0], v[1]) & (mk0 & pm), Dot2<T>::pk(v[2], v[3]) & (mk1 & pm));
  };
  float s1 = 0.f, s2 = 0.f;

  auto dummy_stores = [&]() { mp.store(rsY, 0u, 0u, DW_OOB); };   // (the prologue issues the access sequence of a steady-state
                                                                   //  iteration: see the backward kernel)
#pragma unroll
  for (int p = 0; p < PD; p++) { issue(p, slot[p]); dummy_stores(); }
  stage(slot[0], 1, true);          // plane p lives in ring slot (p + 1) % RB
  issue(PD, slot[0]);
  dummy_stores();

  for (int t0 = 0; t0 < g.T; t0 += UN) {
#pragma unroll
    for (int d = 0; d < UN; d++) {
      const int t = t0 + d;
      if (!EXACT && t >= g.T) break;
      const int qm = d % RB, q0 = (d + 1) % RB, qp = (d + 2) % RB;   // ring slots of planes t-1, t, t+1 (t0 % RB == 0)
      stage(slot[(d + 1) % PD], qp, t + 1 < g.T);
      issue(t + 1 + PD, slot[(d + 1) % PD]);
      f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};   // two chains of three dependent MFMAs
      const int qs[3] = {qm, q0, qp};
#pragma unroll
      for (int kt = 0; kt < 3; kt++) {
        const x8 b01 = *(const x8*)(rd01 + qs[kt] * MX_TILE);
        const x8 b2 = *(const x8*)(rd2 + qs[kt] * MX_TILE);
        acc = MxOp<T>::mfma(A[2 * kt], b01, acc);
        acc2 = MxOp<T>::mfma(A[2 * kt + 1], b2, acc2);
      }
      acc += acc2;
      // acc[j] = out[h][4 s + j]
      float o4[4];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const unsigned m = e < 2 ? mk0 : mk1;
        o4[e] = (m >> (16 * (e & 1)) & 1u) ? acc[e] : 0.f;
        s1 += o4[e];
        s2 += o4[e] * o4[e];
      }
      mp.store(rsY, Dot2<T>::pk(o4[0], o4[1]), Dot2<T>::pk(o4[2], o4[3]), t * planeB);
    }
  }
  if (a.stats || a.pool) {
    const float q1 = wave_sum(s1), q2 = wave_sum(s2);
    float qp[P7 ? 4 : 1];
    if constexpr (P7) {     // the squeeze-excite pool is per sample: the four planes' sums separately
#pragma unroll
      for (int k = 0; k < 4; k++) qp[k] = wave_sum((2 * (h >> 3) + (s >> 1)) == k ? s1 : 0.f);
    }
    if (lane == 0) {
      if (a.stats) {
        double* sp = stats_replica(a.stats, g.C, (unsigned)blockIdx.x);
        atomic_add_d(&sp[c * 2], (double)q1);
        atomic_add_d(&sp[c * 2 + 1], (double)q2);
      }
      if (a.pool) {
        if constexpr (P7) {
          const int n0 = (int)(blockIdx.x / g.C) * 4;
#pragma unroll
          for (int k = 0; k < 4; k++) if (n0 + k < g.N) atomic_add_d(&a.pool[(long long)(n0 + k) * g.C + c], (double)qp[k]);
        } else {
          atomic_add_d(&a.pool[(long long)mp.n * g.C + c], (double)q1);
        }
      }
    }
  }
}

// planes the tile covers: one plane of 12 .. 14 rows and 12 / 14 columns (even: strips start at 4-byte addresses), or 7 x 7
// planes four to a tile
static bool mx_plane_ok(const DwGeom& g, bool* p7) {
  *p7 = g.H == 7 && g.W == 7;
  // (a 10 x 10 plane fills 39 % of the tile: measured 85 -> 87 us forward, 185 -> 240 us backward against the vector kernels)
  return *p7 || (g.H >= 12 && g.H <= 14 && g.W >= 12 && g.W <= 14 && (g.W & 1) == 0);
}

// X3D_DW_MX=0: never (A/B hook)
bool dw_fwd_mx_launch(const DwFwdArgs& a, int dtype, int S, hipStream_t st) {
  const DwGeom& g = a.g;
  const int e = x3d_env_int("X3D_DW_MX", 1);   // (tools/ab_mx.py switches it inside one process: X3D_EXPERIMENTS build)
  bool p7;
  if (e == 0 || dtype == X3D_F32 || S != 1 || !mx_plane_ok(g, &p7) || a.bn.stats) return false;
  const long long bytes = (long long)g.N * g.C * g.T * g.H * g.W * 2;
  if (bytes >= (1ll << 30) || (long long)g.C * g.N >= (1ll << 31)) return false;
  if (((uintptr_t)a.x & 3) || ((uintptr_t)a.y & 3)) return false;
  const bool exact = g.T % 4 == 0;
  if (x3d_describe.out) {
    snprintf(x3d_describe.out, x3d_describe.cap, "dw3d_fwd_mx14_kernel<%s, %s, %d>", dtype == X3D_BF16 ? "bf16" : "f16",
             exact ? "4, 4, 4, 1" : "6, 3, 3, 0", (int)p7);
    return true;
  }
  DwMxFwdArgs pa;
  pa.f = a;
  pa.bytes = (unsigned)bytes;
  const dim3 grid((unsigned)(g.C * (p7 ? ceil_div(g.N, 4) : g.N)));
#define MX_FWD(TT, P7_)                                                                                             \
  do {                                                                                                              \
    if (exact) hipLaunchKernelGGL((dw3d_fwd_mx14_kernel<TT, 4, 4, 4, true, P7_>), grid, dim3(64), 0, st, pa);       \
    else hipLaunchKernelGGL((dw3d_fwd_mx14_kernel<TT, 6, 3, 3, false, P7_>), grid, dim3(64), 0, st, pa);            \
  } while (0)
  if (dtype == X3D_BF16) { if (p7) MX_FWD(bf16, true); else MX_FWD(bf16, false); }
  else { if (p7) MX_FWD(f16, true); else MX_FWD(f16, false); }
#undef MX_FWD
  return true;
}

// ================================================================================================
// FUSED BACKWARD on the matrix cores (same planes, same organisation: one wave per (n, c) channel).
//   dB = cA * dv + cB * braw + cC   (BN_b / SE backward folded per (n, c): coef_nc),   A = relu(s * araw + t)
//   dA[t][h][w] = sum w[kt][kh][kw] * dB[t - kt + 1][h - kh + 1][w - kw + 1]   -- the forward product with the reversed
//       kernel: six MFMAs per plane against the dB image (ring of three planes), weight operands built once;
//   dW[kt][kh][kw] = sum_{t,h,w} dB[t][h][w] * A[t + kt - 1][h + kh - 1][w + kw - 1]:  for a fixed (kt, kw)
//       C[h][h'] = sum_w dB[t][h][w] * A[t + kt - 1][h'][w + kw - 1]   is one MFMA (M = h, N = h', K = w: both operands are
//       rows of the LDS images, 8 consecutive columns per lane) and dW[kt][kh][kw] is the diagonal h' = h + kh - 1 of C summed
//       over the planes -- nine accumulators (36 VGPRs) kept over the whole channel, the diagonals taken once at the end.
//       The column shift kw - 1 of the A rows is made in registers: one aligned ds_read_b128 + the dwords either side, five
//       v_alignbit for the three shifted operands (the A image has 8 zero columns left of column 0, pitch 80 bytes).
//       The upper half of K (lane groups 2, 3) is zero: those lanes read the dB operand from a zero row.
// Per plane and lane: three 8-byte loads, two LDS writes, 6 + 9 MFMAs, ~80 vector instructions (the vector kernel: ~350).
// ================================================================================================
#define MXA_PITCH 80                       // A image: 8 zero columns | 16 columns | 8 columns of padding, 64 + 16 bytes
#define MXA_TILE (MXA_PITCH * MX_ROWS)     // 1440 bytes per plane

struct DwMxBwdArgs {
  DwBwdArgs b;
  unsigned bytes;
};

template <typename T, int UN, int RB, int PD, bool EXACT, bool P7>
__global__ __launch_bounds__(64, 3) void dw3d_bwd_mx14_kernel(const DwMxBwdArgs pa) {
  static_assert(UN % RB == 0 && UN % 2 == 0 && UN % PD == 0, "ring periods RB (dB planes), 2 (A planes) and PD (planes in flight)");
  typedef typename MxOp<T>::x8 x8;
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_;
  // dB ring (RB planes) | A ring (2 planes) | one zero row
  __shared__ __attribute__((aligned(16))) unsigned char lds[RB * MX_TILE + 2 * MXA_TILE + 64];
  unsigned char* ldsA = lds + RB * MX_TILE;
  const DwBwdArgs& a = pa.b;
  const DwGeom& g = a.g;
  const int lane = threadIdx.x;
  const int h = lane & 15, s = lane >> 4;
  int c;
  MxMap<P7> mp;
  mp.build(lane, blockIdx.x, g.N, g.C, g.T, g.H, g.W, c);

  for (int i = lane; i < (RB * MX_TILE + 2 * MXA_TILE + 64) / 16; i += 64) ((uint4*)lds)[i] = make_uint4(0u, 0u, 0u, 0u);

  x8 Wt[6];
  mx_weight_operands<T, true>(a.w + c * 27, lane, Wt);
  const float sc = a.ss_a[c * 2], sh = a.ss_a[c * 2 + 1];
  float cA = 0.f, cB = 0.f, cC = 0.f;          // per (sample, channel): uniform for one plane per wave, per lane for four
  if (mp.n < g.N) {
    const float* cf = a.coef_nc + ((long long)mp.n * g.C + c) * 4;
    cA = cf[0]; cB = cf[1]; cC = cf[2];
  }
  const int planeB = mp.planeB;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((T*)a.araw, 0, pa.bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc((T*)a.dv, 0, pa.bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((T*)a.braw, 0, pa.bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc((T*)a.ga, 0, pa.bytes, 0x00020000);
  // dB image: as the forward's plane image.  dA operand reads: row h (+ kh), columns 8 (s & 1)
  unsigned char* stgB = lds + (h + 1) * MX_PITCH + s * 8;
  const unsigned char* rd01 = lds + (h + (s >> 1)) * MX_PITCH + (s & 1) * 16;
  const unsigned char* rd2 = lds + (h + 2) * MX_PITCH + (s & 1) * 16;
  // dW: M-operand = the dB row of output row h (image row h = LDS row h + 1), lane groups 2, 3 a zero row (K = 16 .. 31 unused)
  const bool lowk = s < 2;
  const unsigned char* rdM = lowk ? lds + (h + 1) * MX_PITCH + (s & 1) * 16 : lds + RB * MX_TILE + 2 * MXA_TILE;
  const int rdM_tile = lowk ? MX_TILE : 0;
  // A image: image column c at byte 16 + 2 c of the row.  Own strip at row h + 1; N-operand = the A row h' = lane & 15 (LDS row
  // h' + 1), columns 8 (s & 1) - 1 .. + 8 (one aligned b128, the dword before and the dword after)
  unsigned char* stgA = ldsA + (h + 1) * MXA_PITCH + 16 + s * 8;
  const unsigned char* rdN = ldsA + (h + 1) * MXA_PITCH + 16 + (s & 1) * 16;

  struct Slot { Raw A, D, R; };
  Slot slot[PD];        // plane p travels in slot p % PD
  auto issue = [&](int t, Slot& q) {
    const int soff = t < g.T ? t * planeB : DW_OOB;
    raw_bload<8>(q.A, rsA, mp.voffL, soff);
    raw_bload<8>(q.D, rsD, mp.voffL, soff);
    raw_bload<8>(q.R, rsR, mp.voffL, soff);
  };
  // validity of the strip's halves as AND masks on the packed pairs (rows 14, 15 and columns 14, 15 hold zeros)
  const unsigned mk0 = mp.mk0, mk1 = mp.mk1;
  auto stage = [&](const Slot& q, int qb, int qa, bool plane_ok) {
    const Raw rA = mp.aligned(q.A), rD = mp.aligned(q.D), rR = mp.aligned(q.R);
    float av[4], bv[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
      av[e] = fmaxf(__builtin_fmaf(sc, raw_get<T>(rA, e), sh), 0.f);
      bv[e] = __builtin_fmaf(cA, raw_get<T>(rD, e), __builtin_fmaf(cB, raw_get<T>(rR, e), cC));
    }
    const unsigned pm = plane_ok ? 0xffffffffu : 0u;        // (uniform; false once: the zero plane behind the last one)
    const unsigned m0 = mk0 & pm, m1 = mk1 & pm;
    *(uint2*)(stgA + qa * MXA_TILE) = make_uint2(Dot2<T>::pk(av[0], av[1]) & m0, Dot2<T>::pk(av[2], av[3]) & m1);
    *(uint2*)(stgB + qb * MX_TILE) = make_uint2(Dot2<T>::pk(bv[0], bv[1]) & m0, Dot2<T>::pk(bv[2], bv[3]) & m1);
  };

  f32x4 Cw[9];        // Cw[kt * 3 + kw][r] = C[h = 4 s + r][h' = lane & 15]
#pragma unroll
  for (int k = 0; k < 9; k++) Cw[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float s1 = 0.f, s2 = 0.f;
  Raw own;            // araw strip of the plane whose dA is emitted next

  // the prologue issues the load / store sequence of a steady-state iteration (its stores dropped: out of range): the loop
  // header merges the prologue's and the back edge's counts of outstanding accesses, and with fewer in the prologue every
  // iteration's wait for a prefetched plane would also wait for the previous iteration's stores (vmcnt retires in order)
  auto dummy_stores = [&]() { mp.store(rsG, 0u, 0u, DW_OOB); };
#pragma unroll
  for (int p = 0; p < PD; p++) { issue(p, slot[p]); dummy_stores(); }
  stage(slot[0], 1, 0, true);        // plane p: dB ring slot (p + 1) % RB, A ring slot p & 1
  own = mp.aligned(slot[0].A);
  issue(PD, slot[0]);
  dummy_stores();

  for (int t0 = 0; t0 < g.T; t0 += UN) {
#pragma unroll
    for (int d = 0; d < UN; d++) {
      const int t = t0 + d;
      if (!EXACT && t >= g.T) break;
      const int qs[3] = {d % RB, (d + 1) % RB, (d + 2) % RB};   // dB ring slots of planes t-1, t, t+1
      const int sl = (d + 1) % PD;
      stage(slot[sl], qs[2], (d + 1) & 1, t + 1 < g.T);
      const Raw own_next = mp.aligned(slot[sl].A);
      issue(t + 1 + PD, slot[sl]);
      // ---- data gradient of plane t
      f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kt = 0; kt < 3; kt++) {
        const x8 b01 = *(const x8*)(rd01 + qs[kt] * MX_TILE);
        const x8 b2 = *(const x8*)(rd2 + qs[kt] * MX_TILE);
        acc = MxOp<T>::mfma(Wt[2 * kt], b01, acc);
        acc2 = MxOp<T>::mfma(Wt[2 * kt + 1], b2, acc2);
      }
      // ---- weight gradient: the A rows of plane t (three column shifts) against the dB rows of planes t+1, t, t-1 (kt = 0, 1, 2)
      {
        const unsigned char* pn = rdN + (d & 1) * MXA_TILE;
        const u32x4_ mid = *(const u32x4_*)pn;
        const unsigned prv = *(const unsigned*)(pn - 4), nxt = *(const unsigned*)(pn + 16);
        const unsigned m01 = __builtin_amdgcn_alignbit(mid[1], mid[0], 16), m12 = __builtin_amdgcn_alignbit(mid[2], mid[1], 16),
                       m23 = __builtin_amdgcn_alignbit(mid[3], mid[2], 16);
        const u32x4_ lft = {__builtin_amdgcn_alignbit(mid[0], prv, 16), m01, m12, m23};     // columns 8 g' - 1 .. (kw = 0)
        const u32x4_ rgt = {m01, m12, m23, __builtin_amdgcn_alignbit(nxt, mid[3], 16)};     // columns 8 g' + 1 .. (kw = 2)
        const x8 n0 = __builtin_bit_cast(x8, lft), n1 = __builtin_bit_cast(x8, mid), n2 = __builtin_bit_cast(x8, rgt);
#pragma unroll
        for (int kt = 0; kt < 3; kt++) {
          const x8 m = *(const x8*)(rdM + qs[2 - kt] * rdM_tile);
          Cw[kt * 3 + 0] = MxOp<T>::mfma(m, n0, Cw[kt * 3 + 0]);
          Cw[kt * 3 + 1] = MxOp<T>::mfma(m, n1, Cw[kt * 3 + 1]);
          Cw[kt * 3 + 2] = MxOp<T>::mfma(m, n2, Cw[kt * 3 + 2]);
          }
      }
      // ---- emit dA[t]: ReLU mask of BN_a, the BN_a backward sums of the stored gradient
      acc += acc2;
      float o4[4];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const float av = raw_get<T>(own, e);                       // (rows / columns outside the image load zeros: the
        o4[e] = (__builtin_fmaf(sc, av, sh) > 0.f && ((e < 2 ? mk0 : mk1) >> (16 * (e & 1)) & 1u)) ? acc[e] : 0.f;   //  lane masks close them)
        s1 += o4[e];
        s2 += o4[e] * av;
      }
      mp.store(rsG, Dot2<T>::pk(o4[0], o4[1]), Dot2<T>::pk(o4[2], o4[3]), t * planeB);
      own = own_next;
    }
  }

  // ---- dW[kt][kh][kw] = sum over lanes / registers with h' - h + 1 == kh of Cw[kt][kw]
  float red[29];
#pragma unroll
  for (int kt = 0; kt < 3; kt++)
#pragma unroll
    for (int kw = 0; kw < 3; kw++) {
      float p[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int kh = h - (4 * s + r) + 1;      // h' = lane & 15 = h here (the N index), row index 4 s + r
        const float v = Cw[kt * 3 + kw][r];
        p[0] += kh == 0 ? v : 0.f;
        p[1] += kh == 1 ? v : 0.f;
        p[2] += kh == 2 ? v : 0.f;
      }
#pragma unroll
      for (int kh = 0; kh < 3; kh++) red[kt * 9 + kh * 3 + kw] = wave_sum_lane63(p[kh]);
    }
  red[27] = wave_sum_lane63(s1);
  red[28] = wave_sum_lane63(s2);
  dw_flush_sums29(red, lane, a.dw + c * 27, a.a_sums + c * 2);
}

bool dw_bwd_mx_launch(const DwBwdArgs& a, int dtype, int S, hipStream_t st) {
  const DwGeom& g = a.g;
  const int e = x3d_env_int("X3D_DW_MX", 1);
  bool p7;
  // fp16 (round 5): as bf16.  dB = cA * dv + cB * braw + cC is rounded to the storage type for the matrix cores; in fp16 it can
  // leave the range where an fp32 dB would not -- by about the factor the stored ga = conv^T(dB) leaves it as well, and exactly
  // what the reference's mixed_float16 policy does (its BatchNorm gradient is an fp16 tensor, utils.py:176-192): an overflow is
  // an inf, the loss scale halves and the step is skipped (train.py:99-100 LossScaleOptimizer; Trainer.step)
  if (e == 0 || (dtype != X3D_BF16 && (dtype != X3D_F16 || x3d_env_int("X3D_DW_MX_F16", 1) == 0)) || S != 1 || !mx_plane_ok(g, &p7)) return false;
  // 7 x 7 planes: 64.3 -> 60.3 us per launch in isolation, but 66.8 -> 73.2 us inside the X3D-M step (operands in the infinity
  // cache: the packed vector kernel gains more from that): the forward only, unless X3D_DW_MX=7
  if (p7 && e != 7) return false;
  const long long bytes = (long long)g.N * g.C * g.T * g.H * g.W * 2;
  if (bytes >= (1ll << 30) || (long long)g.C * g.N >= (1ll << 31)) return false;
  if (((uintptr_t)a.araw & 3) || ((uintptr_t)a.ga & 3) || ((uintptr_t)a.dv & 3) || ((uintptr_t)a.braw & 3)) return false;
  const bool exact = g.T % 4 == 0 && !(p7 && dtype != X3D_BF16);   // (f16 + four 7 x 7 planes per tile: the exit-free variant spills)
  if (x3d_describe.out) {
    snprintf(x3d_describe.out, x3d_describe.cap, "dw3d_bwd_mx14_kernel<%s, %s, %d>", dtype == X3D_BF16 ? "bf16" : "f16", exact ? "4, 4, 2, 1" : "6, 3, 3, 0", (int)p7);
    return true;
  }
  DwMxBwdArgs pa;
  pa.b = a;
  pa.bytes = (unsigned)bytes;
  const dim3 grid((unsigned)(g.C * (p7 ? ceil_div(g.N, 4) : g.N)));
#define MX14_BWD(T_)                                                                                              \
  do {                                                                                                            \
    if (p7) {                                                                                                     \
      if (exact) hipLaunchKernelGGL((dw3d_bwd_mx14_kernel<T_, 4, 4, 2, true, true>), grid, dim3(64), 0, st, pa);    \
      else hipLaunchKernelGGL((dw3d_bwd_mx14_kernel<T_, 6, 3, 3, false, true>), grid, dim3(64), 0, st, pa);         \
    } else {                                                                                                      \
      if (exact) hipLaunchKernelGGL((dw3d_bwd_mx14_kernel<T_, 4, 4, 2, true, false>), grid, dim3(64), 0, st, pa);   \
      else hipLaunchKernelGGL((dw3d_bwd_mx14_kernel<T_, 6, 3, 3, false, false>), grid, dim3(64), 0, st, pa);        \
    }                                                                                                             \
  } while (0)
  // (four planes in flight instead of two, <4, 4, 4>: 104.9 -> 101.8 us in isolation, 165 VGPRs; the 28 x 28 kernel 263.5 -> 269.5 us
  // at 216 VGPRs: neither is short of loads in flight -- round 4, not kept)
  if (dtype == X3D_BF16) MX14_BWD(bf16); else MX14_BWD(f16);
#undef MX14_BWD
  return true;
}


// ================================================================================================
// FUSED BACKWARD, wider planes (stride 1, rows of 26 .. 30 elements: X3D-M stage 3 at 28 x 28), H-tiled.
// One wave per (n, c, H-tile of 14 rows).  A whole image row fits ONE K = 32 window, so
//   * the LDS images hold the tile's 16 window rows (image rows r0 - 1 .. r0 + 14: the rows r0 - 1 and r0 + 14 belong to the
//     neighbour tiles, or are the zero pad), image column c at column c + 8 (eight zero columns left and right), one guard row
//     above and below; lane (n = lane & 15, g = lane >> 4) loads the strips g and g + 4 of window row n and emits the same
//     strips of the SAME row (outputs of the window rows 1 .. 14);
//   * dA: column tile j (outputs 16 j .. 16 j + 15) takes the K window at image column 16 j - 8 -- 16-byte aligned in LDS --
//     with ONE tap row per MFMA: A[w][k] = w'[kt][kh][k - w - 7] (w' the reversed kernel), the same nine operands for both
//     tiles, B = the dB row n + kh - 1: 18 MFMAs per plane;
//   * dW: C[m][n] = sum_k dB[m][k] * A[n][k + kw - 1] with K = the whole row: M-operand = dB row m for the tile's OWN rows
//     (1 .. 14; the halo rows and rows below the image read a zero row: they are the neighbour tiles' terms), N-operand = A row
//     n (all 16 window rows) in three column shifts; dW[kt][kh][kw] = the diagonal n = m + kh - 1.
// ================================================================================================
#define MXW_PITCH 112                       // 8 + 32 + 8 columns = 96 bytes of row + 16: conflict-free b128 rows
#define MXW_TILE (MXW_PITCH * MX_ROWS)      // guard row | 16 window rows | guard row

struct DwMxwBwdArgs {
  DwBwdArgs b;
  unsigned bytes;
  int HT;            // H-tiles (of 14 rows) per plane
};

template <typename T, int UN, int RB, int PD, bool EXACT>
__global__ __launch_bounds__(64, 2) void dw3d_bwd_mxw_kernel(const DwMxwBwdArgs pa) {
  static_assert(UN % RB == 0 && UN % 2 == 0 && UN % PD == 0, "ring periods RB (dB planes), 2 (A planes) and PD (planes in flight)");
  typedef typename MxOp<T>::x8 x8;
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_;
  // dB ring (RB planes) | A ring (2 planes) | one zero row
  __shared__ __attribute__((aligned(16))) unsigned char lds[(RB + 2) * MXW_TILE + 128];
  unsigned char* ldsA = lds + RB * MXW_TILE;
  const DwBwdArgs& a = pa.b;
  const DwGeom& g = a.g;
  const int lane = threadIdx.x;
  const int n16 = lane & 15, s = lane >> 4;
  const int ht = __builtin_amdgcn_readfirstlane(blockIdx.x % pa.HT);
  const int nc = __builtin_amdgcn_readfirstlane(blockIdx.x / pa.HT);
  const int c = nc % g.C, n = nc / g.C;
  const int H = g.H, W = g.W;
  const int r0 = ht * 14;

  for (int i = lane; i < ((RB + 2) * MXW_TILE + 128) / 16; i += 64) ((uint4*)lds)[i] = make_uint4(0u, 0u, 0u, 0u);

  // nine operands of the reversed kernel: A[w = lane & 15][k = 8 s + j] = w[2 - kt][2 - kh][2 - kw], kw = k - w - 7
  x8 Wt[9];
  {
    const float* wc = a.w + c * 27;
#pragma unroll
    for (int i = 0; i < 9; i++) {
      const int kt = i / 3, kh = i % 3;
      const float* w3 = wc + (2 - kt) * 9 + (2 - kh) * 3;
      const float w0 = w3[2], w1 = w3[1], w2 = w3[0];
      x8 op;
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int kw = 8 * s + j - n16 - 7;
        op[j] = (T)(kw == 0 ? w0 : (kw == 1 ? w1 : (kw == 2 ? w2 : 0.f)));
      }
      Wt[i] = op;
    }
  }
  const float sc = a.ss_a[c * 2], sh = a.ss_a[c * 2 + 1];
  const float* cf = a.coef_nc + ((long long)n * g.C + c) * 4;
  const float cA = cf[0], cB = cf[1], cC = cf[2];

  const int planeB = H * W * 2;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((T*)a.araw, 0, pa.bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc((T*)a.dv, 0, pa.bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((T*)a.braw, 0, pa.bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc((T*)a.ga, 0, pa.bytes, 0x00020000);
  const int row = r0 - 1 + n16;                                     // image row of window row n16
  const bool row_in = row >= 0 && row < H;                          // loaded (else the zero pad)
  const bool row_own = n16 >= 1 && n16 <= 14 && row < H;            // emitted by this tile
  const long long base = ((long long)n * g.C + c) * g.T * H * W + (long long)row * W;
  int vld[2], vs0[2], vs1[2];
  unsigned mk0[2], mk1[2], eo0[2], eo1[2];
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int col = 16 * j + 4 * s;
    const bool c0 = col < W, c1 = col + 2 < W;
    vld[j] = (row_in && c0) ? (int)((base + col) * 2) : DW_OOB;
    mk0[j] = (row_in && c0) ? 0xffffffffu : 0u;
    mk1[j] = (row_in && c1) ? 0xffffffffu : 0u;
    eo0[j] = (row_own && c0) ? 1u : 0u;
    eo1[j] = (row_own && c1) ? 1u : 0u;
    vs0[j] = eo0[j] ? (int)((base + col) * 2) : DW_OOB;
    vs1[j] = eo1[j] ? (int)((base + col + 2) * 2) : DW_OOB;
  }
  // window row n16 lives in LDS row n16 + 1; image column c at byte 16 + 2 c
  unsigned char* stgB = lds + (n16 + 1) * MXW_PITCH + 16 + s * 8;            // (+ 32 j)
  unsigned char* stgA = ldsA + (n16 + 1) * MXW_PITCH + 16 + s * 8;
  const unsigned char* rdB = lds + n16 * MXW_PITCH + s * 16;                 // dA: row n16 + kh - 1 (+ kh * PITCH), K window (+ 32 j)
  const unsigned char* rdM = row_own ? lds + (n16 + 1) * MXW_PITCH + 16 + s * 16 : lds + (RB + 2) * MXW_TILE;
  const int rdM_tile = row_own ? MXW_TILE : 0;
  const unsigned char* rdN = ldsA + (n16 + 1) * MXW_PITCH + 16 + s * 16;

  struct Slot { Raw A[2], D[2], R[2]; };
  Slot slot[PD];
  auto issue = [&](int t, Slot& q) {
    const int soff = t < g.T ? t * planeB : DW_OOB;
#pragma unroll
    for (int j = 0; j < 2; j++) {
      raw_bload<8>(q.A[j], rsA, vld[j], soff);
      raw_bload<8>(q.D[j], rsD, vld[j], soff);
      raw_bload<8>(q.R[j], rsR, vld[j], soff);
    }
  };
  auto stage = [&](const Slot& q, int qb, int qa, bool plane_ok) {
    const unsigned pm = plane_ok ? 0xffffffffu : 0u;
#pragma unroll
    for (int j = 0; j < 2; j++) {
      float av[4], bv[4];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        av[e] = fmaxf(__builtin_fmaf(sc, raw_get<T>(q.A[j], e), sh), 0.f);
        bv[e] = __builtin_fmaf(cA, raw_get<T>(q.D[j], e), __builtin_fmaf(cB, raw_get<T>(q.R[j], e), cC));
      }
      const unsigned m0 = mk0[j] & pm, m1 = mk1[j] & pm;
      *(uint2*)(stgA + qa * MXW_TILE + 32 * j) = make_uint2(Dot2<T>::pk(av[0], av[1]) & m0, Dot2<T>::pk(av[2], av[3]) & m1);
      *(uint2*)(stgB + qb * MXW_TILE + 32 * j) = make_uint2(Dot2<T>::pk(bv[0], bv[1]) & m0, Dot2<T>::pk(bv[2], bv[3]) & m1);
    }
  };
  auto store4 = [&](int j, unsigned p0, unsigned p1, int soff) {
    Raw o, o1;
    o.w[0] = p0; o1.w[0] = p1;
    raw_bstore<4>(o, rsG, vs0[j], soff);
    raw_bstore<4>(o1, rsG, vs1[j], soff);
  };
  auto dummy_stores = [&]() { store4(0, 0u, 0u, DW_OOB); store4(1, 0u, 0u, DW_OOB); };

  f32x4 Cw[9];
#pragma unroll
  for (int k = 0; k < 9; k++) Cw[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float s1 = 0.f, s2 = 0.f;
  Raw own[2];

#pragma unroll
  for (int p = 0; p < PD; p++) { issue(p, slot[p]); dummy_stores(); }
  stage(slot[0], 1, 0, true);        // plane p: dB ring slot (p + 1) % RB, A ring slot p & 1
  own[0] = slot[0].A[0]; own[1] = slot[0].A[1];
  issue(PD, slot[0]);
  dummy_stores();

  for (int t0 = 0; t0 < g.T; t0 += UN) {
#pragma unroll
    for (int d = 0; d < UN; d++) {
      const int t = t0 + d;
      if (!EXACT && t >= g.T) break;
      const int qs[3] = {d % RB, (d + 1) % RB, (d + 2) % RB};
      const int sl = (d + 1) % PD;
      stage(slot[sl], qs[2], (d + 1) & 1, t + 1 < g.T);
      const Raw on0 = slot[sl].A[0], on1 = slot[sl].A[1];
      issue(t + 1 + PD, slot[sl]);
      // ---- weight gradient
      {
        const unsigned char* pn = rdN + (d & 1) * MXW_TILE;
        const u32x4_ mid = *(const u32x4_*)pn;
        const unsigned prv = *(const unsigned*)(pn - 4), nxt = *(const unsigned*)(pn + 16);
        const unsigned m01 = __builtin_amdgcn_alignbit(mid[1], mid[0], 16), m12 = __builtin_amdgcn_alignbit(mid[2], mid[1], 16),
                       m23 = __builtin_amdgcn_alignbit(mid[3], mid[2], 16);
        const u32x4_ lft = {__builtin_amdgcn_alignbit(mid[0], prv, 16), m01, m12, m23};
        const u32x4_ rgt = {m01, m12, m23, __builtin_amdgcn_alignbit(nxt, mid[3], 16)};
        const x8 n0 = __builtin_bit_cast(x8, lft), n1 = __builtin_bit_cast(x8, mid), n2 = __builtin_bit_cast(x8, rgt);
#pragma unroll
        for (int kt = 0; kt < 3; kt++) {
          const x8 m = *(const x8*)(rdM + qs[2 - kt] * rdM_tile);
          Cw[kt * 3 + 0] = MxOp<T>::mfma(m, n0, Cw[kt * 3 + 0]);
          Cw[kt * 3 + 1] = MxOp<T>::mfma(m, n1, Cw[kt * 3 + 1]);
          Cw[kt * 3 + 2] = MxOp<T>::mfma(m, n2, Cw[kt * 3 + 2]);
        }
      }
      // ---- data gradient of plane t, the two column tiles, and its emit
#pragma unroll
      for (int j = 0; j < 2; j++) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f}, acc3 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < 3; kt++) {
          const unsigned char* pb = rdB + qs[kt] * MXW_TILE + 32 * j;
          acc = MxOp<T>::mfma(Wt[kt * 3 + 0], *(const x8*)(pb), acc);
          acc2 = MxOp<T>::mfma(Wt[kt * 3 + 1], *(const x8*)(pb + MXW_PITCH), acc2);
          acc3 = MxOp<T>::mfma(Wt[kt * 3 + 2], *(const x8*)(pb + 2 * MXW_PITCH), acc3);
        }
        acc += acc2 + acc3;
        float o4[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const float av = raw_get<T>(own[j], e);
          o4[e] = (__builtin_fmaf(sc, av, sh) > 0.f && (e < 2 ? eo0[j] : eo1[j])) ? acc[e] : 0.f;
          s1 += o4[e];
          s2 += o4[e] * av;
        }
        store4(j, Dot2<T>::pk(o4[0], o4[1]), Dot2<T>::pk(o4[2], o4[3]), t * planeB);
      }
      own[0] = on0; own[1] = on1;
    }
  }

  // ---- dW[kt][kh][kw]: the diagonal n = m + kh - 1 of Cw[kt][kw] (n = lane & 15, m = 4 s + r)
  float red[29];
#pragma unroll
  for (int kt = 0; kt < 3; kt++)
#pragma unroll
    for (int kw = 0; kw < 3; kw++) {
      float p[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int kh = n16 - (4 * s + r) + 1;
        const float v = Cw[kt * 3 + kw][r];
        p[0] += kh == 0 ? v : 0.f;
        p[1] += kh == 1 ? v : 0.f;
        p[2] += kh == 2 ? v : 0.f;
      }
#pragma unroll
      for (int kh = 0; kh < 3; kh++) red[kt * 9 + kh * 3 + kw] = wave_sum_lane63(p[kh]);
    }
  red[27] = wave_sum_lane63(s1);
  red[28] = wave_sum_lane63(s2);
  dw_flush_sums29(red, lane, a.dw + c * 27, a.a_sums + c * 2);
}

// X3D_DW_MXW=0: never (A/B hook)
bool dw_bwd_mxw_launch(const DwBwdArgs& a, int dtype, int S, hipStream_t st) {
  const DwGeom& g = a.g;
  const int e = x3d_env_int("X3D_DW_MXW", 1);
  // measured (108 ch x 64 clips of 16 x 28 x 28): 240 -> 226 us per launch, 235 -> 222 us inside the X3D-M step, at two waves
  // per SIMD (188 VGPRs: nine weight operands + nine dW accumulators + two strips of three tensors in flight).  Rows of 20
  // elements in tiles of 14 + 6 rows (X3D-XL stage 4) lose (341 -> 564 us): only planes that fill the tiles
  if (e == 0 || (dtype != X3D_BF16 && (dtype != X3D_F16 || x3d_env_int("X3D_DW_MX_F16", 1) == 0)) || S != 1 || (g.W & 1) || g.W < 26 || g.W > 30 || g.H < 12 ||
      (g.H % 14 != 0 && g.H % 14 < 10)) return false;
  const long long bytes = (long long)g.N * g.C * g.T * g.H * g.W * 2;
  const int HT = ceil_div(g.H, 14);
  if (bytes >= (1ll << 30) || (long long)g.C * g.N * HT >= (1ll << 31)) return false;
  if (((uintptr_t)a.araw & 7) || ((uintptr_t)a.ga & 3) || ((uintptr_t)a.dv & 7) || ((uintptr_t)a.braw & 7)) return false;
  // (f16: the exit-free <4, 4, 2> variant spills 248 bytes per lane under hipcc 7.2 -- 256 VGPRs against bf16's 188 -- so fp16 takes the
  // general one, 211 VGPRs, whatever T)
  const bool exact = g.T % 4 == 0 && dtype == X3D_BF16;
  if (x3d_describe.out) {
    snprintf(x3d_describe.out, x3d_describe.cap, "dw3d_bwd_mxw_kernel<%s, %s>", dtype == X3D_BF16 ? "bf16" : "f16", exact ? "4, 4, 2, 1" : "6, 3, 3, 0");
    return true;
  }
  DwMxwBwdArgs pa;
  pa.b = a;
  pa.bytes = (unsigned)bytes;
  pa.HT = HT;
  const dim3 grid((unsigned)((long long)g.C * g.N * HT));
  if (dtype == X3D_BF16) {
    if (exact) hipLaunchKernelGGL((dw3d_bwd_mxw_kernel<bf16, 4, 4, 2, true>), grid, dim3(64), 0, st, pa);
    else hipLaunchKernelGGL((dw3d_bwd_mxw_kernel<bf16, 6, 3, 3, false>), grid, dim3(64), 0, st, pa);
  } else {
    if (exact) hipLaunchKernelGGL((dw3d_bwd_mxw_kernel<f16, 4, 4, 2, true>), grid, dim3(64), 0, st, pa);
    else hipLaunchKernelGGL((dw3d_bwd_mxw_kernel<f16, 6, 3, 3, false>), grid, dim3(64), 0, st, pa);
  }
  return true;
}
