// Channelwise 3x3x3 convolution on the matrix cores, stride-1 planes of ANY width in 16-bit storage: the H- and W-tiled
// generalisation of dw3d_bwd_mxw_kernel (dw_mx.hip, where the Toeplitz formulation is described).
//
// One wave per (n, c, H-tile of 14 rows, W-tile).  A W-tile is a WINDOW of 16 NT image columns (NT = 2 or 3 column tiles of
// 16 outputs) starting at image column w0; the wave OWNS the columns [c_lo, c_hi) of it and loads the rest as halo:
//   * rows that fit one window (26 .. 46 columns: 28 x 28, 39 x 39) are one W-tile, w0 = 0;
//   * wider rows are cut at multiples of 4 columns into tiles of `own_w` columns (56 = 28 + 28 with NT = 2, 78 = 40 + 38 with
//     NT = 3); tile i > 0 starts its window 4 columns early, so its first own strip is a whole strip (own columns are a PREFIX
//     of every strip: stores are one or two 4-byte halves, plus one 2-byte element on rows of odd length);
//   * rows of odd length (39) start at odd element addresses: the 8-byte strip loads / 4-byte stores are then 2-byte aligned,
//     which the compute queues' unaligned access mode allows; a strip cut by the row end reads into the next row (masked) or
//     past the tensor (the buffer resource returns zeros).
// LDS row = 8 zero columns | 32 KSW window columns | 8 zero columns (KSW = K-steps of 32 columns the weight gradient takes
// over a row: 1 for NT = 2, 2 for NT = 3), at a pitch of an odd number of 16-byte units (conflict-free b128 rows).
//   * dA of column tile j: K window at window column 16 j - 8 (16-byte aligned), one tap row per MFMA, the nine reversed-kernel
//     operands shared by all tiles: 9 NT MFMAs per plane;
//   * dW: C[m][n] = sum_k dB[m][k] A[n][k + kw - 1] over the row's KSW K-steps, the dB operand masked to the OWN columns (a
//     halo column is the neighbour tile's term): 9 KSW MFMAs; dW[kt][kh][kw] = the diagonal n = m + kh - 1.
// Reference: model.py:259-267 (the depthwise `b` convolution) through SURVEY appendix A.
#include "dw_common.h"

template <typename T> struct MxgOp;
template <> struct MxgOp<bf16> {
  typedef bf16x8 x8;
  static __device__ __forceinline__ f32x4 mfma(x8 a, x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct MxgOp<f16> {
  typedef f16x8 x8;
  static __device__ __forceinline__ f32x4 mfma(x8 a, x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};

#define MXG_ROWS 18    // guard row | 16 window rows | guard row

struct DwMxgBwdArgs {
  DwBwdArgs b;
  unsigned bytes;    // whole-tensor size (buffer num_records)
  int HT, WT;        // H-tiles (of 14 rows) and W-tiles per plane
  int own_w;         // own columns of a W-tile (a multiple of 4; the last tile takes what is left)
  int own0;          // ... of the first W-tile (it has no left halo strip, so it may own 4 columns more)
};

// element masks of a strip: `v` leading elements valid -> AND masks of its two dwords
__device__ __forceinline__ void mxg_prefix_masks(int v, unsigned& m0, unsigned& m1) {
  m0 = v >= 2 ? 0xffffffffu : (v == 1 ? 0x0000ffffu : 0u);
  m1 = v >= 4 ? 0xffffffffu : (v == 3 ? 0x0000ffffu : 0u);
}

template <typename T, int NT, bool WTILED, bool ODD, int UN, int RB, int PD, bool EXACT>
__global__ __launch_bounds__(64, 2) void dw3d_bwd_mxg_kernel(const DwMxgBwdArgs pa) {
  static_assert(UN % RB == 0 && UN % 2 == 0 && UN % PD == 0, "ring periods RB (dB planes), 2 (A planes) and PD (planes in flight)");
  static_assert(NT == 2 || NT == 3, "two or three column tiles");
  typedef typename MxgOp<T>::x8 x8;
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_;
  constexpr int KSW = (NT + 1) / 2;
  constexpr int PITCH = (8 + 32 * KSW + 8) * 2 + 16;     // 112 (NT = 2) / 176 (NT = 3) bytes: 7 / 11 units of 16
  constexpr int TILE = PITCH * MXG_ROWS;
  // dB ring (RB planes) | A ring (2 planes) | one zero row
  __shared__ __attribute__((aligned(16))) unsigned char lds[(RB + 2) * TILE + 128];
  unsigned char* ldsA = lds + RB * TILE;
  const DwBwdArgs& a = pa.b;
  const DwGeom& g = a.g;
  const int lane = threadIdx.x;
  const int n16 = lane & 15, s = lane >> 4;
  int b = blockIdx.x;
  const int wt = __builtin_amdgcn_readfirstlane(b % pa.WT); b /= pa.WT;
  const int ht = __builtin_amdgcn_readfirstlane(b % pa.HT);
  const int nc = __builtin_amdgcn_readfirstlane(b / pa.HT);
  const int c = nc % g.C, n = nc / g.C;
  const int H = g.H, W = g.W;
  const int r0 = ht * 14;
  const int c_lo = wt ? pa.own0 + (wt - 1) * pa.own_w : 0, c_hi = min(W, c_lo + (wt ? pa.own_w : pa.own0));   // own image columns
  const int w0 = wt ? c_lo - 4 : 0;                                  // image column of window column 0

  for (int i = lane; i < ((RB + 2) * TILE + 128) / 16; i += 64) ((uint4*)lds)[i] = make_uint4(0u, 0u, 0u, 0u);

  // nine operands of the reversed kernel: A[w = lane & 15][k = 8 s + j] = w[2 - kt][2 - kh][2 - kw], kw = k - w - 7
  x8 Wt[9];
  {
    const float* wc = a.w + c * 27;
#pragma unroll
    for (int i = 0; i < 9; i++) {
      const int kt = i / 3, kh = i % 3;
      const float* w3 = wc + (2 - kt) * 9 + (2 - kh) * 3;
      const float w0_ = w3[2], w1_ = w3[1], w2_ = w3[0];
      x8 op;
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int kw = 8 * s + j - n16 - 7;
        op[j] = (T)(kw == 0 ? w0_ : (kw == 1 ? w1_ : (kw == 2 ? w2_ : 0.f)));
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
  int vld[NT], vs0[NT], vs1[NT], vse[NT];
  unsigned mk0[NT], mk1[NT];
  int nown[NT];                                                     // own elements of the strip (a prefix: 0 .. 4)
  int lsh[NT];                                                      // ODD: bits a loaded strip moves down by
#pragma unroll
  for (int j = 0; j < NT; j++) {
    const int col = w0 + 16 * j + 4 * s;                            // image column of the strip's first element
    const int nv = row_in ? max(0, min(4, W - col)) : 0;            // valid (inside the row)
    const int no = (row_own && col >= c_lo) ? max(0, min(4, c_hi - col)) : 0;
    // rows of odd length: a strip cut by the row end is loaded 4 - nv elements EARLY and shifted down when it is consumed, so
    // that the load never leaves the row -- read as it stands, the last strip of the tensor would straddle its end and a raw
    // buffer load returns zeros for a dword that is only partly in range (the 2-byte aligned case; on rows of even length a cut
    // strip's second dword is wholly out of range, which is the zero the masks want anyway)
    const int early = (ODD && nv > 0 && nv < 4) ? 4 - nv : 0;
    lsh[j] = 16 * early;
    vld[j] = nv ? (int)((base + col - early) * 2) : DW_OOB;
    mxg_prefix_masks(nv, mk0[j], mk1[j]);
    nown[j] = no;
    vs0[j] = no >= 2 ? (int)((base + col) * 2) : DW_OOB;
    vs1[j] = no >= 4 ? (int)((base + col + 2) * 2) : DW_OOB;
    vse[j] = (ODD && (no & 1)) ? (int)((base + col + no - 1) * 2) : DW_OOB;
  }
  // own-column masks of the weight gradient's dB operand: lane s reads window columns 32 ks + 8 s .. + 7
  unsigned mm[KSW][4];
  if constexpr (WTILED) {
#pragma unroll
    for (int ks = 0; ks < KSW; ks++)
#pragma unroll
      for (int d = 0; d < 4; d++) {
        const int col = w0 + 32 * ks + 8 * s + 2 * d;
        mm[ks][d] = ((col >= c_lo && col < c_hi) ? 0x0000ffffu : 0u) | ((col + 1 >= c_lo && col + 1 < c_hi) ? 0xffff0000u : 0u);
      }
  }
  // window row n16 lives in LDS row n16 + 1; window column k at byte 16 + 2 k
  unsigned char* stgB = lds + (n16 + 1) * PITCH + 16 + s * 8;                // (+ 32 j)
  unsigned char* stgA = ldsA + (n16 + 1) * PITCH + 16 + s * 8;
  const unsigned char* rdB = lds + n16 * PITCH + s * 16;                     // dA: row n16 + kh - 1 (+ kh * PITCH), K window (+ 32 j)
  const unsigned char* rdM = row_own ? lds + (n16 + 1) * PITCH + 16 + s * 16 : lds + (RB + 2) * TILE;   // (+ 64 ks)
  const int rdM_tile = row_own ? TILE : 0;
  const unsigned char* rdN = ldsA + (n16 + 1) * PITCH + 16 + s * 16;

  struct Slot { Raw A[NT], D[NT], R[NT]; };
  Slot slot[PD];
  auto issue = [&](int t, Slot& q) {
    const int soff = t < g.T ? t * planeB : DW_OOB;
#pragma unroll
    for (int j = 0; j < NT; j++) {
      raw_bload<8>(q.A[j], rsA, vld[j], soff);
      raw_bload<8>(q.D[j], rsD, vld[j], soff);
      raw_bload<8>(q.R[j], rsR, vld[j], soff);
    }
  };
  auto placed = [&](const Raw& r, int j) {       // the loaded strip with its first element in element 0
    if constexpr (ODD) {
      const unsigned long long v = (((unsigned long long)r.w[1] << 32) | r.w[0]) >> lsh[j];
      Raw o; o.w[0] = (unsigned)v; o.w[1] = (unsigned)(v >> 32); o.w[2] = o.w[3] = 0u;
      return o;
    } else {
      return r;
    }
  };
  auto stage = [&](Slot& q, int qb, int qa, bool plane_ok) {
    const unsigned pm = plane_ok ? 0xffffffffu : 0u;
#pragma unroll
    for (int j = 0; j < NT; j++) {
      q.A[j] = placed(q.A[j], j);                 // (kept: the emit of this plane reads its own strip again)
      const Raw qd = placed(q.D[j], j), qr = placed(q.R[j], j);
      float av[4], bv[4];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        av[e] = fmaxf(__builtin_fmaf(sc, raw_get<T>(q.A[j], e), sh), 0.f);
        bv[e] = __builtin_fmaf(cA, raw_get<T>(qd, e), __builtin_fmaf(cB, raw_get<T>(qr, e), cC));
      }
      const unsigned m0 = mk0[j] & pm, m1 = mk1[j] & pm;
      *(uint2*)(stgA + qa * TILE + 32 * j) = make_uint2(Dot2<T>::pk(av[0], av[1]) & m0, Dot2<T>::pk(av[2], av[3]) & m1);
      *(uint2*)(stgB + qb * TILE + 32 * j) = make_uint2(Dot2<T>::pk(bv[0], bv[1]) & m0, Dot2<T>::pk(bv[2], bv[3]) & m1);
    }
  };
  auto store4 = [&](int j, unsigned p0, unsigned p1, int soff) {
    Raw o, o1;
    o.w[0] = p0; o1.w[0] = p1;
    raw_bstore<4>(o, rsG, vs0[j], soff);
    raw_bstore<4>(o1, rsG, vs1[j], soff);
    if constexpr (ODD) {
      Raw oe;
      oe.w[0] = nown[j] == 1 ? p0 : p1;       // element 0 or element 2: the low half of its pair
      raw_bstore<2>(oe, rsG, vse[j], soff);
    }
  };
  auto dummy_stores = [&]() {
#pragma unroll
    for (int j = 0; j < NT; j++) store4(j, 0u, 0u, DW_OOB);
  };

  f32x4 Cw[9];
#pragma unroll
  for (int k = 0; k < 9; k++) Cw[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float s1 = 0.f, s2 = 0.f;
  Raw own[NT];

#pragma unroll
  for (int p = 0; p < PD; p++) { issue(p, slot[p]); dummy_stores(); }
  stage(slot[0], 1, 0, true);        // plane p: dB ring slot (p + 1) % RB, A ring slot p & 1
#pragma unroll
  for (int j = 0; j < NT; j++) own[j] = slot[0].A[j];
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
      Raw on[NT];
#pragma unroll
      for (int j = 0; j < NT; j++) on[j] = slot[sl].A[j];
      issue(t + 1 + PD, slot[sl]);
      // ---- weight gradient
#pragma unroll
      for (int ks = 0; ks < KSW; ks++) {
        const unsigned char* pn = rdN + (d & 1) * TILE + 64 * ks;
        const u32x4_ mid = *(const u32x4_*)pn;
        const unsigned prv = *(const unsigned*)(pn - 4), nxt = *(const unsigned*)(pn + 16);
        const unsigned m01 = __builtin_amdgcn_alignbit(mid[1], mid[0], 16), m12 = __builtin_amdgcn_alignbit(mid[2], mid[1], 16),
                       m23 = __builtin_amdgcn_alignbit(mid[3], mid[2], 16);
        const u32x4_ lft = {__builtin_amdgcn_alignbit(mid[0], prv, 16), m01, m12, m23};
        const u32x4_ rgt = {m01, m12, m23, __builtin_amdgcn_alignbit(nxt, mid[3], 16)};
        const x8 n0 = __builtin_bit_cast(x8, lft), n1 = __builtin_bit_cast(x8, mid), n2 = __builtin_bit_cast(x8, rgt);
#pragma unroll
        for (int kt = 0; kt < 3; kt++) {
          u32x4_ mr = *(const u32x4_*)(rdM + qs[2 - kt] * rdM_tile + 64 * ks);
          if constexpr (WTILED) { mr[0] &= mm[ks][0]; mr[1] &= mm[ks][1]; mr[2] &= mm[ks][2]; mr[3] &= mm[ks][3]; }
          const x8 m = __builtin_bit_cast(x8, mr);
          Cw[kt * 3 + 0] = MxgOp<T>::mfma(m, n0, Cw[kt * 3 + 0]);
          Cw[kt * 3 + 1] = MxgOp<T>::mfma(m, n1, Cw[kt * 3 + 1]);
          Cw[kt * 3 + 2] = MxgOp<T>::mfma(m, n2, Cw[kt * 3 + 2]);
        }
      }
      // ---- data gradient of plane t, the column tiles, and its emit
#pragma unroll
      for (int j = 0; j < NT; j++) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f}, acc3 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < 3; kt++) {
          const unsigned char* pb = rdB + qs[kt] * TILE + 32 * j;
          acc = MxgOp<T>::mfma(Wt[kt * 3 + 0], *(const x8*)(pb), acc);
          acc2 = MxgOp<T>::mfma(Wt[kt * 3 + 1], *(const x8*)(pb + PITCH), acc2);
          acc3 = MxgOp<T>::mfma(Wt[kt * 3 + 2], *(const x8*)(pb + 2 * PITCH), acc3);
        }
        acc += acc2 + acc3;
        float o4[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const float av = raw_get<T>(own[j], e);
          o4[e] = (__builtin_fmaf(sc, av, sh) > 0.f && e < nown[j]) ? acc[e] : 0.f;
          s1 += o4[e];
          s2 += o4[e] * av;
        }
        store4(j, Dot2<T>::pk(o4[0], o4[1]), Dot2<T>::pk(o4[2], o4[3]), t * planeB);
      }
#pragma unroll
      for (int j = 0; j < NT; j++) own[j] = on[j];
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

// (A FORWARD kernel on the same tiling -- ring of input planes, 9 NT MFMAs per plane -- was built and measured in round 4: 108 ch x 16
// clips of 16 x 39 x 39 66.8 -> 83.4 us, 54 ch x 16 of 78 x 78 125.5 -> 203.6 us against the ragged vector kernel, which moves two
// tensors at 2.5 - 2.9 TB/s already; the window's unused columns and rows cost more than the 27 multiply-adds per output save.
// Removed: only the backward, 54 multiply-adds per output and VALU-bound in the vector form, gains.)

// the tiling of a plane: false when the shape is not covered
static bool mxg_tiling(const DwGeom& g, int* NTp, int* WTp, int* own_w, int* own0, int* HTp) {
  const int W = g.W, H = g.H;
  if (W < 26 || H < 12) return false;
  // the last H-tile must be worth a wave (rows 14 k + 1 .. 14 k + 7 waste more than half of it)
  if (H % 14 != 0 && H % 14 < 8) return false;
  *HTp = ceil_div(H, 14);
  // (78 columns as THREE windows of 32 -- own 28 + 24 + 26 -- instead of two of 48: 344 -> 468 us, 235 -> 323 us in isolation.  Fewer,
  // wider waves win: a wave's set-up -- operands, ring zeroing, pipeline fill -- and its 29 reductions + atomics at the end are
  // paid per wave; twice the planes per wave (T = 32 instead of 16 at the same bytes) is -9 % at 14 x 14 and -7 % at 28 x 28)
  // W-tiles of `own` columns (a multiple of 4, the last tile takes the rest).  Window columns a tile needs: the first
  // own + 1 (right halo), a middle one 4 + own + 1, the last 4 + its own columns (right of it is the zero pad)
  for (int wt = W <= 48 ? 1 : ceil_div(W, 40); wt <= 16; wt++) {
    const int own = ceil_div(ceil_div(W, wt), 4) * 4;
    const int last = W - own * (wt - 1);
    if (last <= 0) return false;
    for (int nt = 2; nt <= 3; nt++) {
      const bool fits = wt == 1 ? W <= 16 * nt : (own + 1 <= 16 * nt && (wt == 2 || own + 5 <= 16 * nt) && 4 + last <= 16 * nt);
      if (fits) { *NTp = nt; *WTp = wt; *own_w = own; *own0 = own; return true; }
    }
  }
  return false;
}

bool dw_bwd_mxg_launch(const DwBwdArgs& a, int dtype, int S, hipStream_t st) {
  const DwGeom& g = a.g;
  const int e = x3d_env_int("X3D_DW_MXG", 1);   // A/B hook (experiments builds): 0 = never
  int NT, WT, own_w, own0, HT;
  // (fp16 since round 5: dw_mx.hip, dw_bwd_mx_launch)
  if (e == 0 || (dtype != X3D_BF16 && (dtype != X3D_F16 || x3d_env_int("X3D_DW_MX_F16", 1) == 0)) || S != 1 ||
      !mxg_tiling(g, &NT, &WT, &own_w, &own0, &HT)) return false;
  // Only rows the vector kernels must take with ragged (flat, unaligned) staging.  Measured in isolation, vector -> this kernel:
  // 108 ch x 16 clips of 16 x 39 x 39 (X3D-L stage 3) 226.7 -> 160.5 us, 162 ch x 8 (XL) 157.3 -> 109.3; 54 ch x 16 of 78 x 78
  // (two W-tiles of 40 + 38 columns in windows of 48) 380.5 -> 361.7, 72 ch x 8 275.5 -> 242.4; but the ALIGNED 56 x 56 plane of
  // X3D-M (54 ch x 64 clips, two W-tiles of 28 in windows of 32) 599.1 -> 622.9: the aligned vector kernel keeps those
  // (e == 2: every covered shape -- experiments builds)
  if ((g.W & 7) == 0 && e != 2) return false;
  const long long bytes = (long long)g.N * g.C * g.T * g.H * g.W * 2;
  if (bytes >= (1ll << 30) || (long long)g.C * g.N * HT * WT >= (1ll << 31)) return false;
  if (((uintptr_t)a.araw & 1) || ((uintptr_t)a.ga & 1) || ((uintptr_t)a.dv & 1) || ((uintptr_t)a.braw & 1)) return false;
  const bool odd = (g.W & 1) != 0, wtiled = WT > 1;
  // (f16, one W-tile of two column tiles, even rows: the exit-free variant spills 188 bytes per lane -- the general one then)
  const bool exact = g.T % 4 == 0 && !(dtype != X3D_BF16 && NT == 2 && !wtiled && !odd);
  if (x3d_describe.out) {
    snprintf(x3d_describe.out, x3d_describe.cap, "dw3d_bwd_mxg_kernel<%s, %d, %d, %d, %s>", dtype == X3D_BF16 ? "bf16" : "f16", NT, (int)wtiled, (int)odd, exact ? "4, 4, 2, 1" : "6, 3, 3, 0");
    return true;
  }
  DwMxgBwdArgs pa;
  pa.b = a;
  pa.bytes = (unsigned)bytes;
  pa.HT = HT; pa.WT = WT; pa.own_w = own_w; pa.own0 = own0;
  const dim3 grid((unsigned)((long long)g.C * g.N * HT * WT));
#define MXG_GO_T(T_, NT_, WTL_, ODD_) do { \
    if (exact) hipLaunchKernelGGL((dw3d_bwd_mxg_kernel<T_, NT_, WTL_, ODD_, 4, 4, 2, true>), grid, dim3(64), 0, st, pa); \
    else hipLaunchKernelGGL((dw3d_bwd_mxg_kernel<T_, NT_, WTL_, ODD_, 6, 3, 3, false>), grid, dim3(64), 0, st, pa); \
    return true; } while (0)
#define MXG_GO(NT_, WTL_, ODD_) do { if (dtype == X3D_BF16) MXG_GO_T(bf16, NT_, WTL_, ODD_); else MXG_GO_T(f16, NT_, WTL_, ODD_); } while (0)
  if (NT == 2 && !wtiled && !odd) MXG_GO(2, false, false);
  if (NT == 2 && wtiled && !odd) MXG_GO(2, true, false);
  if (NT == 3 && !wtiled && !odd) MXG_GO(3, false, false);
  if (NT == 3 && !wtiled && odd) MXG_GO(3, false, true);
  if (NT == 3 && wtiled && !odd) MXG_GO(3, true, false);
  if (NT == 3 && wtiled && odd) MXG_GO(3, true, true);
  if (NT == 2 && !wtiled && odd) MXG_GO(2, false, true);
  if (NT == 2 && wtiled && odd) MXG_GO(2, true, true);
#undef MXG_GO
#undef MXG_GO_T
  return false;
}
