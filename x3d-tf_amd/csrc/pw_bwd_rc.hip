// x3d_pw_bwd with a RECOMPUTED conv output (rc_panel != NULL): the fused backward of an `a` conv that never reads a_raw.
//
// The BatchNorm behind the conv makes its backward an affine map of the upstream gradient and the conv's own output:
//   dY[co][p] = A[co] * g[co][p] + B[co] * y[co][p] + C[co]                      (x3d_bn_bwd_finalize: coef = {A, B, C})
// and y = W x is linear in the conv input, so BOTH gradients can be written without y:
//   dX = W^T dY              = (W^T diag(A)) g  +  (W^T diag(B) W) x  +  W^T C   = [W1 | M] [g ; x] + c0
//   dW = sum_p dY x^T        = diag(A) (sum_p g x^T)  +  diag(B) W (sum_p x x^T)  +  C (sum_p x)^T
// The kernel streams g and x only (pw_bwd_fused.hip streams g, yraw and x: a third less HBM traffic on layers whose output
// is 2.25x their input), applies no prologue at all (the tile is staged as it lies in memory), and one GEMM with
// K = Cout + Cin against a per-step panel [W1 | M] (x3d_pw_bwd_rc_prepare: rounded to the storage type like every matrix-core
// operand) gives dX; the raw moment sums  S = [g ; 1 ; x] x^T  (the ones row yields sum_p x) stay in accumulators across
// the tiles of a workgroup and are added to `rc_sums` with fp32 atomics once; x3d_pw_bwd_rc_finish turns them into dW in
// fp64.  c0 is added in fp32 in the epilogue (a bf16 constant on every point of a row would bias the BatchNorm-backward sums
// downstream).  Epilogues as in pw_bwd_fused.hip: + add | + strided add, and the folded residual-tail backward (tail_c).
//
// LDS: ONE tile image Z [KT*32 + MT*32][128 points]: rows [0, Co) = g, row Co = ones, rows up to KT*32 zero, rows
// [KT*32, KT*32 + Ci) = x, rest zero -- pitch 320 B with the 16-byte units of row k XOR-swizzled by (k >> 2) & 3 (as the dY
// tile of pw_bwd_fused.hip: the transposed read for dX and the row reads for the moment sums are both conflict-free).  The
// panel's K index IS the Z row index.  The fp32 output slab aliases Z (every row of Z is rewritten by the next commit).
#include <stdlib.h>

#include "pw_gemm.h"

typedef __attribute__((ext_vector_type(4))) short s16x4_r;
typedef __attribute__((ext_vector_type(8))) short s16x8_r;

struct PwBwdRcArgs {
  const void* g; const void* x;
  const void* wp;                       // rc panel [MT*32][ZR + 8], ZR = rows of the tile image
  const float* c0;                      // [Ci]
  void* dx;                             // [N][Ci][P]
  const void* add;
  int eH, eW;
  float* sums;                          // [Co + 1 + Ci][Ci] +=
  int N, Co, Ci, Kg, Cip;               // Kg = roundup(Co + 1, 16), Cip = roundup(Ci, 16)
  long long P;
  int tiles_per_block;
  const void* tail_c; const void* tail_r; double* tail_sums_c; double* tail_sums_r;
  // XS (strided shortcut conv, reference model.py:360-367): x is the block input [N][Ci][T][xH][xW], the conv reads its
  // pixels (2h, 2w); P counts the OUTPUT points (eH x eW per frame)
  long long Pin;
  int xH, xW;
};

#define RC_BN 128
#define RC_YP 160    // Z pitch (elements): 320 B
#define RC_OP 132    // fp32 output slab pitch

// MT: 32-row tiles of Ci; KT: 32-row tiles of Co + 1.  MT == 2 (Ci in 33..48): the image holds 48 x rows, not 64 -- with 64
// the 48 <-> 108 layer needs 87 KB of LDS and runs one workgroup per CU; the moment-sum MFMAs then read 16 rows past the image
// (into the panel that follows it: finite values, landing only in discarded rows / columns of the accumulators)
// XS: 0 = dense conv input; 4 / 2 / 1 = strided shortcut conv, outputs per aligned load of the gather (common.h).
// E4V (strided add): 1 = rows of whole 8-point vectors (eW % 8 == 0: one 8-byte load of the four even-column operands), 2 = rows of
// whole 4-point groups (eW % 4 == 0, e.g. 28: two 4-byte loads per vector), 0 = element by element
template <typename H, int MT, int KT, int EPI, int TAIL, int E4V, int XS = 0>
__global__ __launch_bounds__(256, (KT <= 2 && !TAIL && MT == 1 && !XS) ? 3 : (KT > 4 ? 1 : 2)) void pw_bwd_rc_kernel(const PwBwdRcArgs a) {   // (KT > 4: one workgroup per CU by LDS anyway)
  typedef typename HV<H>::x8 hx8; typedef typename HV<H>::x4 hx4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  typedef H T;
  constexpr int BN = RC_BN, YP = RC_YP, OP = RC_OP;
  constexpr bool TAILR = TAIL == 2;
  constexpr int XR = MT == 1 ? 32 : 48;             // x rows of the tile image
  constexpr int ZR = KT * 32 + XR;                  // rows of the tile image
  constexpr int XR0 = KT * 32;                      // first x row
  constexpr int WP = ZR + 8;                        // panel pitch (elements): an odd number of 16-byte units
  constexpr int RT = KT + MT;                       // row tiles of the moment sums
  constexpr int NT = RT * MT;                       // tiles of the moment sums
  constexpr int TPW = (NT + 3) / 4;
  constexpr int NKS = NT >= 4 ? 1 : 4 / NT;
  constexpr int NVY = KT * 2, NVX = XR / 16, ROWS_PT = MT * 2;
  constexpr size_t ZS_B = (size_t)ZR * YP * 2;
  static_assert((size_t)32 * OP * 4 <= ZS_B, "slab must fit the tile image it aliases");
  H* Zs = (H*)smem_raw;
  H* Ws = (H*)(smem_raw + ZS_B);
  float* Os = (float*)smem_raw;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, half = lane >> 5;
  const int tiles_per_n = (int)((a.P + BN - 1) / BN);
  const int total_tiles = tiles_per_n * a.N;
  const int tile_begin = blockIdx.x * a.tiles_per_block;
  const int tile_end = min(tile_begin + a.tiles_per_block, total_tiles);

  {   // the panel (rows >= roundup(Ci, 32) do not exist: MT covers Ci)
    const hx8* src = (const hx8*)a.wp;
    constexpr int nvec = MT * 32 * WP / 8;
#pragma unroll 4
    for (int i = tid; i < nvec; i += 256) ((hx8*)Ws)[i] = src[i];
  }

  const int srow = tid >> 4, sunit = tid & 15;
  hx8 rg[NVY], rx[NVX], rx2[XS ? NVX : 1];
  unsigned ymask[TAIL ? ROWS_PT : 1];
  if constexpr (TAIL) {
#pragma unroll
    for (int i = 0; i < ROWS_PT; i++) ymask[i] = 0u;
  }
  // every load of the tile loop is unconditional (clamped address, value selected afterwards): the waits stay countable
  auto issue = [&](int tile) __attribute__((always_inline)) {
    const int n = tile / tiles_per_n;
    const long long p = (long long)(tile - n * tiles_per_n) * BN + sunit * 8;
#pragma unroll
    for (int i = 0; i < NVY; i++) {
      const int k = srow + 16 * i;
      const long long o = (k < a.Co && p < a.P) ? ((long long)n * a.Co + k) * a.P + p : 0;
      rg[i] = *(const hx8*)((const T*)a.g + o);
    }
#pragma unroll
    for (int i = 0; i < NVX; i++) {
      const int m = srow + 16 * i;
      if constexpr (XS) {     // the even pixels of the even rows: 8 outputs = 16 input elements in rx | rx2 (clamped: row 0, point 0)
        const bool ok = m < a.Ci && p < a.P;
        strided_gather16<XS>((const T*)a.x + ((long long)n * a.Ci + (ok ? m : 0)) * a.Pin, ok ? p : 0, a.xH, a.xW, a.eH, a.eW, rx[i], rx2[i]);
      } else {
        const long long o = (m < a.Ci && p < a.P) ? ((long long)n * a.Ci + m) * a.P + p : 0;
        rx[i] = *(const hx8*)((const T*)a.x + o);
      }
    }
  };
  auto commit = [&](int tile) __attribute__((always_inline)) {
    const int n = tile / tiles_per_n;
    const long long p = (long long)(tile - n * tiles_per_n) * BN + sunit * 8;
    const bool pin = p < a.P;
#pragma unroll
    for (int i = 0; i < NVY; i++) {       // all KT*32 rows, every tile (the slab overwrote the first of them)
      const int k = srow + 16 * i;
      hx8 v = rg[i];
      if (!(pin && k < a.Co)) {
        const H fill = (pin && k == a.Co) ? (H)1.f : (H)0.f;     // the ones row: sum_p x out of the same MFMAs
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = fill;
      }
      *(hx8*)&Zs[k * YP + ((sunit ^ ((k >> 2) & 3)) << 3)] = v;
    }
#pragma unroll
    for (int i = 0; i < NVX; i++) {
      const int m = srow + 16 * i;
      if constexpr (XS) {     // output j of the vector = gathered element 2 j
        hx8 v;
#pragma unroll
        for (int e = 0; e < 4; e++) { v[e] = rx[i][2 * e]; v[4 + e] = rx2[i][2 * e]; }
        rx[i] = v;
      }
      if (!(pin && m < a.Ci)) {
#pragma unroll
        for (int e = 0; e < 8; e++) rx[i][e] = (H)0.f;
      }
      const int k = XR0 + m;
      *(hx8*)&Zs[k * YP + ((sunit ^ ((k >> 2) & 3)) << 3)] = rx[i];
      if constexpr (TAIL) {
        unsigned mk = 0;
#pragma unroll
        for (int e = 0; e < 8; e++) mk |= ((float)rx[i][e] > 0.f ? 1u : 0u) << e;
        ymask[i] = mk;
      }
    }
  };

  float tg[TAIL ? ROWS_PT : 1], tgc[TAIL ? ROWS_PT : 1], tgr[TAILR ? ROWS_PT : 1];
  if constexpr (TAIL) {
#pragma unroll
    for (int i = 0; i < ROWS_PT; i++) { tg[i] = 0.f; tgc[i] = 0.f; }
  }
  if constexpr (TAILR) {
#pragma unroll
    for (int i = 0; i < ROWS_PT; i++) tgr[i] = 0.f;
  }
  float c0r[ROWS_PT];
#pragma unroll
  for (int i = 0; i < ROWS_PT; i++) {
    const int m = (tid >> 4) + 16 * i;
    c0r[i] = m < a.Ci ? a.c0[m] : 0.f;
  }

  // transposed-read lane geometry (tile image as B operand): lane -> (row 8*(g16>>1)+q (+4), 4 points)
  const int g16 = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  const int tr_row = 8 * (g16 >> 1) + q;
  const int tr_unit = wid * 4 + 2 * (g16 & 1) + (pp >> 1);
  const int tr_off0 = ((tr_unit ^ ((tr_row >> 2) & 3)) << 3) + (pp & 1) * 4;
  const int tr_off1 = ((tr_unit ^ (((tr_row + 4) >> 2) & 3)) << 3) + (pp & 1) * 4;
  typedef s16x4_r __attribute__((address_space(3))) * lds_s16x4_ptr;

  f32x16 acc_s[TPW];
#pragma unroll
  for (int s = 0; s < TPW; s++)
#pragma unroll
    for (int j = 0; j < 16; j++) acc_s[s][j] = 0.f;

  // S[row][ci] += Z[row][:] . Z[XR0 + ci][:] over the 128 points of the tile
  auto sums_mfma = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < TPW; s++) {
      int id = wid + 4 * s, kpart = 0;
      if constexpr (NKS > 1) { id = wid % NT; kpart = wid / NT; }
      if (id < NT && (NKS == 1 || kpart < NKS)) {
        const int rt = id / MT, cit = id - rt * MT;
        const int rowa = rt * 32 + r, rowb = XR0 + cit * 32 + r;
        const int swa = (rowa >> 2) & 3, swb = (rowb >> 2) & 3;
        const H* arow = Zs + rowa * YP;
        const H* brow = Zs + rowb * YP;
        constexpr int KSTEPS = (BN / 16) / NKS;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ks++) {
          const int u = (kpart * KSTEPS + ks) * 2 + half;
          const hx8 af = *(const hx8*)(arow + ((u ^ swa) << 3));
          const hx8 bf = *(const hx8*)(brow + ((u ^ swb) << 3));
          acc_s[s] = mfma16<H>(af, bf, acc_s[s]);
        }
      }
    }
  };

  __syncthreads();            // panel in place
  if (tile_begin < tile_end) issue(tile_begin);
  for (int tile = tile_begin; tile < tile_end; ++tile) {
    const int n = tile / tiles_per_n;
    const long long p0 = (long long)(tile - n * tiles_per_n) * BN;
    __syncthreads();            // every reader of the previous tile's image / slab is done
    commit(tile);
    __syncthreads();

    // ---- global loads of this tile's epilogue, THEN the next tile's prefetch (countable waits: pw_bwd_fused.hip)
    const int oc = (tid & 15) * 8;
    constexpr bool EPL8 = (EPI == X3D_EPI_ADD), EPL4 = (EPI == X3D_EPI_ADD_STRIDED);
    hx8 epl8[EPL8 ? ROWS_PT : 1];
    hx4 epl4[EPL4 ? ROWS_PT : 1];
    hx8 tc8[TAIL ? ROWS_PT : 1], tr8[TAILR ? ROWS_PT : 1];
#pragma unroll
    for (int i = 0; i < ROWS_PT; i++) {
      const int m = (tid >> 4) + 16 * i;
      const long long p = p0 + oc;
      const bool ok = m < a.Ci && p < a.P;
      const long long orow = ok ? ((long long)n * a.Ci + m) * a.P + p : 0;
      if constexpr (TAIL) {
        const hx8 lc = *(const hx8*)((const T*)a.tail_c + orow);
#pragma unroll
        for (int e = 0; e < 8; e++) tc8[i][e] = ok ? lc[e] : (H)0.f;
        if constexpr (TAILR) {
          const hx8 lr = *(const hx8*)((const T*)a.tail_r + orow);
#pragma unroll
          for (int e = 0; e < 8; e++) tr8[i][e] = ok ? lr[e] : (H)0.f;
        }
      }
      if constexpr (EPL8) {
        const hx8 l8 = *(const hx8*)((const T*)a.add + orow);
#pragma unroll
        for (int e = 0; e < 8; e++) epl8[i][e] = ok ? l8[e] : (H)0.f;
      } else if constexpr (EPL4) {
        const int hw = a.eH * a.eW;      // per-sample point counts fit 32 bits (host check): 32-bit divisions
        const int Hh = (a.eH + 1) >> 1, Wh = (a.eW + 1) >> 1;
        const int T_ = (int)a.P / hw;
        const int t = (int)p / hw;
        const int rem = (int)p - t * hw;
        const int h = rem / a.eW, w = rem - h * a.eW;
        if constexpr (E4V == 2) {      // two groups of four points: each inside one image row (eW % 4 == 0), the second may be in the next
          typedef typename HV<H>::x2 hx2;
          // (rows m >= Ci of the last sample: no address may be formed from them -- the clamp is on the WHOLE offset)
          const long long abase = ((long long)n * a.Ci + m) * T_ * Hh * Wh;
#pragma unroll
          for (int gq = 0; gq < 2; gq++) {
            int wq = w + 4 * gq, hq = h, tq = t;
            if (wq >= a.eW) { wq -= a.eW; hq++; if (hq >= a.eH) { hq = 0; tq++; } }
            const bool okq = ok && tq < T_ && (hq & 1) == 0;
            const hx2 l2 = *(const hx2*)((const T*)a.add + (okq ? abase + ((long long)tq * Hh + (hq >> 1)) * Wh + (wq >> 1) : 0));
            epl4[i][2 * gq] = okq ? l2[0] : (H)0.f;
            epl4[i][2 * gq + 1] = okq ? l2[1] : (H)0.f;
          }
        } else {
          const bool okv = ok && E4V == 1 && (h & 1) == 0;
          const hx4 l4 = *(const hx4*)((const T*)a.add + (okv ? ((((long long)n * a.Ci + m) * T_ + t) * Hh + (h >> 1)) * Wh + (w >> 1) : 0));
#pragma unroll
          for (int e = 0; e < 4; e++) epl4[i][e] = okv ? l4[e] : (H)0.f;
        }
      }
    }
    issue(min(tile + 1, tile_end - 1));   // (past the end: the last tile again -- the number of loads in flight stays static)

    // ---- dX tile: acc[s] (rows s*32.., this wave's 32 points) = [W1 | M] [g ; x]
    f32x16 acc[MT];
#pragma unroll
    for (int s = 0; s < MT; s++)
#pragma unroll
      for (int j = 0; j < 16; j++) acc[s][j] = 0.f;
    auto kstep = [&](int kk) __attribute__((always_inline)) {
      const s16x4_r b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(&Zs[(kk + tr_row) * YP + tr_off0]));
      const s16x4_r b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(&Zs[(kk + tr_row + 4) * YP + tr_off1]));
      const s16x8_r bs = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
      const hx8 bfrag = __builtin_bit_cast(hx8, bs);
#pragma unroll
      for (int s = 0; s < MT; s++) {
        const hx8 afrag = *(const hx8*)&Ws[(s * 32 + r) * WP + kk + 8 * half];
        acc[s] = mfma16<H>(afrag, bfrag, acc[s]);
      }
    };
    for (int kk = 0; kk < a.Kg; kk += 16) kstep(kk);
    for (int kk = XR0; kk < XR0 + a.Cip; kk += 16) kstep(kk);
    sums_mfma();      // finish with the tile image before the slab reuses its LDS

    // ---- epilogue in 32-row slabs through LDS: thread owns rows (tid>>4) + 16*ii, 8 points at (tid&15)*8
    __amdgpu_buffer_rsrc_t dxr = __builtin_amdgcn_make_buffer_rsrc((T*)a.dx + (long long)n * a.Ci * a.P, 0,
                                                                   (int)((long long)a.Ci * a.P * 2), 0x00020000);
#pragma unroll
    for (int sl = 0; sl < MT; sl++) {
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 16; j++) Os[((j & 3) + 8 * (j >> 2) + 4 * half) * OP + wid * 32 + r] = acc[sl][j];
      __syncthreads();
#pragma unroll
      for (int ii = 0; ii < 2; ii++) {
        const int i = sl * 2 + ii;
        const int row = (tid >> 4) + 16 * ii;
        const int m = sl * 32 + row;
        const long long p = p0 + oc;
        const bool rvalid = m < a.Ci && p < a.P;
        float val[8];
        {
          const f32x4 v0 = *(const f32x4*)&Os[row * OP + oc], v1 = *(const f32x4*)&Os[row * OP + oc + 4];
#pragma unroll
          for (int e = 0; e < 4; e++) { val[e] = v0[e] + c0r[i]; val[4 + e] = v1[e] + c0r[i]; }
        }
        if constexpr (EPI == X3D_EPI_ADD) {
#pragma unroll
          for (int e = 0; e < 8; e++) val[e] += (float)epl8[i][e];
        } else if constexpr (EPL4) {
          if (E4V != 0) {   // loaded above (zeros on odd rows)
#pragma unroll
            for (int e = 0; e < 4; e++) val[2 * e] += (float)epl4[i][e];
          } else if (rvalid) {
            // element by element (rows of odd length, or of 4 k + 2 points): the coordinates of the first point by division, the
            // other seven by stepping -- two divisions per ELEMENT were most of this epilogue's time (272 us for the 48 -> 216 layer
            // on rows of 28 before they got the group form)
            const int hw = a.eH * a.eW;
            const int Hh = (a.eH + 1) >> 1, Wh = (a.eW + 1) >> 1;
            const int T_ = (int)a.P / hw;
            int t = (int)p / hw;
            const int rem = (int)p - t * hw;
            int h = rem / a.eW, w = rem - h * a.eW;
            const T* abase = (const T*)a.add + ((long long)n * a.Ci + m) * T_ * Hh * Wh;
#pragma unroll
            for (int e = 0; e < 8; e++) {
              if (((h | w) & 1) == 0 && t < T_) val[e] += to_f<T>(abase[((long long)t * Hh + (h >> 1)) * Wh + (w >> 1)]);
              if (++w == a.eW) { w = 0; if (++h == a.eH) { h = 0; ++t; } }
            }
          }
        }
        if constexpr (TAIL) {   // Add + ReLU backward of the block this gradient leaves: mask, then the BN_c / BN_r backward sums
          const unsigned mk = ymask[i];
#pragma unroll
          for (int e = 0; e < 8; e++) {
            const float gm = ((mk >> e) & 1u) ? round_to<T>(val[e]) : 0.f;   // the sums describe dx as stored
            val[e] = gm;
            tg[i] += gm;
            tgc[i] += gm * (float)tc8[i][e];
            if constexpr (TAILR) tgr[i] += gm * (float)tr8[i][e];
          }
        }
        {
          hx8 ov;
#pragma unroll
          for (int e = 0; e < 8; e++) ov[e] = (H)val[e];
          typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, ov), dxr,
                                                 rvalid ? (unsigned)(((long long)m * a.P + p) * 2) : 0x80000000u, 0, 0);
        }
      }
    }
  }

  if constexpr (TAIL) {
    if (tile_begin < tile_end) {
#pragma unroll
      for (int i = 0; i < ROWS_PT; i++) {
        const float s0 = row16_sum(tg[i]), s1 = row16_sum(tgc[i]);
        float s2 = 0.f;
        if constexpr (TAILR) s2 = row16_sum(tgr[i]);
        const int m = (tid >> 4) + 16 * i;
        if ((tid & 15) == 0 && m < a.Ci) {
          atomic_add_d(&a.tail_sums_c[m * 2], (double)s0);
          atomic_add_d(&a.tail_sums_c[m * 2 + 1], (double)s1);
          if constexpr (TAILR) {
            atomic_add_d(&a.tail_sums_r[m * 2], (double)s0);
            atomic_add_d(&a.tail_sums_r[m * 2 + 1], (double)s2);
          }
        }
      }
    }
  }

  // ---- moment sums -> global (fp32 atomics): rows [0, Co] = g x^T and the ones row, rows Co + 1 .. = x x^T
  if (tile_begin < tile_end) {
#pragma unroll
    for (int s = 0; s < TPW; s++) {
      int id = wid + 4 * s;
      bool live = true;
      if constexpr (NKS > 1) { id = wid % NT; live = (wid / NT) < NKS; }
      if (id < NT && live) {
        const int rt = id / MT, cit = id - rt * MT;
        const int ci = cit * 32 + r;
#pragma unroll
        for (int j = 0; j < 16; j++) {
          const int zr = rt * 32 + (j & 3) + 8 * (j >> 2) + 4 * half;
          const int sr = zr < XR0 ? (zr <= a.Co ? zr : -1) : (zr - XR0 < a.Ci ? a.Co + 1 + zr - XR0 : -1);
          if (sr >= 0 && ci < a.Ci) atomicAdd(&a.sums[(long long)sr * a.Ci + ci], acc_s[s][j]);
        }
      }
    }
  }
}

static inline int rc_zrows(int MT, int KT) { return KT * 32 + (MT == 1 ? 32 : 48); }
static inline size_t rc_lds_bytes(int MT, int KT) {
  const int ZR = rc_zrows(MT, KT);
  return (size_t)ZR * RC_YP * 2 + (size_t)MT * 32 * (ZR + 8) * 2;
}

template <typename H, int MT, int KT, int EPI, int TAIL, int E4V = 1, int XS = 0>
static int rc_launch(PwBwdRcArgs& a, hipStream_t st) {
  if constexpr (EPI == X3D_EPI_ADD_STRIDED && E4V == 1) {
    // rows of 28 points (stage 4's first block), of 156 (X3D-L / XL stage 2): groups of four
    if ((a.eW & 7) != 0 && (a.eW & 3) == 0) return rc_launch<H, MT, KT, EPI, TAIL, 2>(a, st);
    if ((a.eW & 7) != 0) return rc_launch<H, MT, KT, EPI, TAIL, 0>(a, st);
  }
  const size_t lds = rc_lds_bytes(MT, KT);
  X3D_REQUIRE(lds <= 160 * 1024, "pw_bwd (recomputed output): needs %zu B of LDS", lds);
  if constexpr (XS != 0) X3D_DESCRIBE("pw_bwd_rc_kernel<%s, %d, %d, %d, %d, s%d>", HV<H>::name, MT, KT, EPI, TAIL, XS);
  X3D_DESCRIBE("pw_bwd_rc_kernel<%s, %d, %d, %d, %d%s>", HV<H>::name, MT, KT, EPI, TAIL, E4V == 1 ? "" : (E4V == 2 ? ", g4" : ", e"));
  auto kern = pw_bwd_rc_kernel<H, MT, KT, EPI, TAIL, E4V, XS>;
  static bool attr_set = false;
  static int slots = 0;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
    int nb = 0, dev = 0, cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 256, lds) != hipSuccess || nb < 1) nb = 1;
    slots = nb * cus;
  }
  const long long total_tiles = ceil_div_ll(a.P, RC_BN) * a.N;
  X3D_REQUIRE(total_tiles < (1ll << 31), "pw_bwd: too many tiles");
  long long tpb = ceil_div_ll(total_tiles, slots);   // one balanced round
  if (tpb < 4) tpb = 4;                              // keeps the atomics small against the streamed tiles
  a.tiles_per_block = (int)tpb;
  const long long gx = ceil_div_ll(total_tiles, tpb);
  hipLaunchKernelGGL(kern, dim3((unsigned)gx), dim3(256), lds, st, a);
  X3D_LAUNCH_CHECK("pw_bwd_rc");
  return X3D_OK;
}

// panel geometry of a layer (shared with prepare / finish)
struct RcShape { int MT, KT, ZR, WP, Kg, Cip; };
static inline bool rc_shape(int Cin, int Cout, RcShape* s) {
  s->MT = ceil_div(Cin, 32);
  s->KT = ceil_div(Cout + 1, 32);
  if (s->MT < 1 || s->MT > 2 || Cin > 48 || s->KT < 1 || s->KT > 7) return false;   // (wider inputs: two workgroups per CU no longer fit)
  if (s->MT == 1 && s->KT > 4) return false;
  // two row tiles of x: instantiated for the X3D stage-3 shape class (KT = 3, 4) and for the first `a` conv of stage 4
  // (48 -> 216: KT = 7, 123 KB of LDS with the panel -- one workgroup per CU; its unfused dgrad + wgrad pair moved 2.8x the bytes)
  if (s->MT == 2 && !(s->KT == 3 || s->KT == 4 || s->KT == 7)) return false;
  s->ZR = rc_zrows(s->MT, s->KT);
  s->WP = s->ZR + 8;
  s->Kg = (Cout + 1 + 15) & ~15;
  s->Cip = (Cin + 15) & ~15;
  return true;
}

bool pw_bwd_rc_supported(const x3d_pw_bwd_args* b) {
  if (!x3d_is_half(b->dtype) || !b->rc_panel || !b->rc_c0 || !b->x) return false;   // (rc_sums: checked at the launch, as the tail sums are)
  if (b->epi != X3D_EPI_ADD && b->epi != X3D_EPI_ADD_STRIDED && b->epi != X3D_EPI_STORE) return false;
  RcShape s;
  if (!rc_shape(b->Cin, b->Cout, &s)) return false;
  const long long P = (long long)b->T * b->H * b->W;
  if (P % 8 || P >= (1ll << 31)) return false;
  if ((long long)b->Cin * P * 2 >= (1ll << 31)) return false;   // one sample's dx inside the 2 GB buffer-store window
  const void* ps[] = {b->g, b->x_stride == 2 ? nullptr : b->x, b->dx, b->rc_panel, b->epi == X3D_EPI_ADD ? b->add : nullptr, b->tail_c, b->tail_r};
  for (const void* p : ps) if (p && ((uintptr_t)p % 16)) return false;
  if (b->epi == X3D_EPI_STORE) {
    if (b->add || b->tail_c) return false;
  } else if (!b->add || (b->epi == X3D_EPI_ADD_STRIDED && ((uintptr_t)b->add % 8))) return false;
  if (b->x_stride == 2) {      // the strided shortcut conv: the STORE form only (its dx is the strided-add operand of the `a` backward)
    if (b->epi != X3D_EPI_STORE || b->xH <= 0 || b->xW <= 0 || (b->xH + 1) / 2 != b->H || (b->xW + 1) / 2 != b->W) return false;
    if (!((s.MT == 1 && s.KT <= 2) || (s.MT == 2 && s.KT == 4))) return false;          // (the X3D shortcut shapes: 24->24, 24->48, 48->96)
    if (strided_gather_gv(b->xW, b->W, P, b->x) == 0) return false;
    if ((long long)b->T * b->xH * b->xW >= (1ll << 31)) return false;
  } else if (b->x_stride > 1) return false;
  if (b->tail_r && !b->tail_c) return false;
  // (two row tiles of x: the tail's extra epilogue operands spill 96-176 bytes per lane at the 256-VGPR cap -- those layers keep
  // the separate x3d_tail_bwd pass, as they do with pw_bwd_fused.hip)
  // (measured, 48 <-> 108 @28x28 x 64 clips: 88 us + 54 us of x3d_tail_bwd against 186 us with the tail inside, 276 us with TAIL = 2)
  if (s.MT == 2 && b->tail_c) return false;
  return rc_lds_bytes(s.MT, s.KT) <= 160 * 1024;
}

template <typename H>
static int rc_pick_strided(PwBwdRcArgs& a, int MT, int KT, int gv, hipStream_t st) {
#define RC_XS(M_, K_, G_) if (MT == M_ && KT == K_ && gv == G_) return rc_launch<H, M_, K_, X3D_EPI_STORE, 0, true, G_>(a, st);
  RC_XS(1, 1, 4) RC_XS(1, 1, 2) RC_XS(1, 1, 1) RC_XS(1, 2, 4) RC_XS(1, 2, 2) RC_XS(1, 2, 1) RC_XS(2, 4, 4) RC_XS(2, 4, 2) RC_XS(2, 4, 1)
#undef RC_XS
  x3d_set_error("pw_bwd (recomputed output, strided input): unsupported tile shape");
  return X3D_ERR_INVALID;
}

template <typename H, int EPI>
static int rc_pick(PwBwdRcArgs& a, int MT, int KT, int tail, hipStream_t st) {
#define RC_CASE(M_, K_)                                                      \
  if (MT == M_ && KT == K_) {                                                \
    if (tail == 2) return rc_launch<H, M_, K_, EPI, 2>(a, st);               \
    if (tail == 1) return rc_launch<H, M_, K_, EPI, 1>(a, st);               \
    return rc_launch<H, M_, K_, EPI, 0>(a, st);                              \
  }
  RC_CASE(1, 1) RC_CASE(1, 2) RC_CASE(1, 3) RC_CASE(1, 4)
#undef RC_CASE
  if (MT == 2 && KT == 3 && tail == 0) return rc_launch<H, 2, 3, EPI, 0>(a, st);
  if (MT == 2 && KT == 4 && tail == 0) return rc_launch<H, 2, 4, EPI, 0>(a, st);
  if (MT == 2 && KT == 7 && tail == 0) return rc_launch<H, 2, 7, EPI, 0>(a, st);
  x3d_set_error("pw_bwd (recomputed output): unsupported tile shape");
  return X3D_ERR_INVALID;
}

int pw_bwd_rc(const x3d_pw_bwd_args* b, hipStream_t st) {
  X3D_REQUIRE(pw_bwd_rc_supported(b), "pw_bwd (recomputed output): shape / alignment / epilogue not covered");
  X3D_REQUIRE(!b->tail_c || (b->tail_sums_c && (!b->tail_r || b->tail_sums_r)) || x3d_describe.out, "pw_bwd: tail_c / tail_r need their sums");
  X3D_REQUIRE(b->rc_sums || x3d_describe.out, "pw_bwd (recomputed output): rc_sums is NULL");
  RcShape s;
  rc_shape(b->Cin, b->Cout, &s);
  PwBwdRcArgs a;
  memset(&a, 0, sizeof(a));
  a.g = b->g; a.x = b->x; a.wp = b->rc_panel; a.c0 = b->rc_c0; a.dx = b->dx; a.add = b->add;
  a.eH = b->H; a.eW = b->W; a.sums = b->rc_sums;
  a.N = b->N; a.Co = b->Cout; a.Ci = b->Cin; a.Kg = s.Kg; a.Cip = s.Cip;
  a.P = (long long)b->T * b->H * b->W;
  a.tail_c = b->tail_c; a.tail_r = b->tail_r; a.tail_sums_c = b->tail_sums_c; a.tail_sums_r = b->tail_sums_r;
  const int tail = b->tail_c ? (b->tail_r ? 2 : 1) : 0;
  if (b->x_stride == 2) {
    a.Pin = (long long)b->T * b->xH * b->xW; a.xH = b->xH; a.xW = b->xW;
    const int gv = strided_gather_gv(b->xW, b->W, a.P, b->x);
    return b->dtype == X3D_F16 ? rc_pick_strided<f16>(a, s.MT, s.KT, gv, st) : rc_pick_strided<bf16>(a, s.MT, s.KT, gv, st);
  }
  if (b->epi == X3D_EPI_STORE)
    return b->dtype == X3D_F16 ? rc_pick<f16, X3D_EPI_STORE>(a, s.MT, s.KT, 0, st) : rc_pick<bf16, X3D_EPI_STORE>(a, s.MT, s.KT, 0, st);
  if (b->dtype == X3D_F16)
    return b->epi == X3D_EPI_ADD ? rc_pick<f16, X3D_EPI_ADD>(a, s.MT, s.KT, tail, st) : rc_pick<f16, X3D_EPI_ADD_STRIDED>(a, s.MT, s.KT, tail, st);
  return b->epi == X3D_EPI_ADD ? rc_pick<bf16, X3D_EPI_ADD>(a, s.MT, s.KT, tail, st) : rc_pick<bf16, X3D_EPI_ADD_STRIDED>(a, s.MT, s.KT, tail, st);
}

// ------------------------------------------------------------------------------------------------------------------------
// per-step panel  [W1 | M]  and  c0  from the master weights and the BatchNorm-backward coefficients; dW from the moment sums
// ------------------------------------------------------------------------------------------------------------------------
extern "C" long long x3d_pw_bwd_rc_panel_elems(int Cout, int Cin) {
  RcShape s;
  if (Cout <= 0 || Cin <= 0 || !rc_shape(Cin, Cout, &s)) return 0;
  return (long long)s.MT * 32 * s.WP;
}
extern "C" long long x3d_pw_bwd_rc_sums_elems(int Cout, int Cin) { return (long long)(Cout + 1 + Cin) * Cin; }

template <typename H>
__device__ __forceinline__ void rc_prepare_body(const float* __restrict__ w, const float* __restrict__ coef, H* __restrict__ panel,
                                                float* __restrict__ c0, int Co, int Ci, int rows, int WP, int XR0, int start, int step) {
  for (int i = start; i < rows * WP; i += step) {
    const int ci = i / WP, col = i - ci * WP;
    float v = 0.f;
    if (ci < Ci) {
      if (col < Co) {
        v = round_to<H>(w[(long long)col * Ci + ci]) * coef[col * 4];
      } else if (col >= XR0 && col - XR0 < Ci) {
        const int cj = col - XR0;
        for (int co = 0; co < Co; co++)
          v = fmaf(round_to<H>(w[(long long)co * Ci + ci]) * coef[co * 4 + 1], round_to<H>(w[(long long)co * Ci + cj]), v);
      }
    }
    panel[i] = (H)v;
  }
  for (int ci = start; ci < Ci; ci += step) {
    float v = 0.f;
    for (int co = 0; co < Co; co++) v = fmaf(round_to<H>(w[(long long)co * Ci + ci]), coef[co * 4 + 2], v);
    c0[ci] = v;
  }
}

template <typename H>
__global__ __launch_bounds__(256) void rc_prepare_kernel(const float* __restrict__ w, const float* __restrict__ coef, H* __restrict__ panel,
                                                         float* __restrict__ c0, int Co, int Ci, int rows, int WP, int XR0) {
  rc_prepare_body<H>(w, coef, panel, c0, Co, Ci, rows, WP, XR0, blockIdx.x * 256 + threadIdx.x, gridDim.x * 256);
}

extern "C" int x3d_pw_bwd_rc_prepare(const float* w, const float* coef, void* rc_panel, float* rc_c0, int Cout, int Cin, int dtype,
                                     void* stream) {
  X3D_REQUIRE(w && coef && rc_panel && rc_c0, "pw_bwd_rc_prepare: null pointer");
  X3D_REQUIRE(x3d_is_half(dtype), "pw_bwd_rc_prepare: 16-bit storage types only");
  RcShape s;
  X3D_REQUIRE(Cout > 0 && Cin > 0 && rc_shape(Cin, Cout, &s), "pw_bwd_rc_prepare: layer shape not covered (x3d_pw_bwd_rc_panel_elems() == 0)");
  const int rows = s.MT * 32;
  const dim3 grid((unsigned)ceil_div(rows * s.WP, 256));
  if (dtype == X3D_F16)
    hipLaunchKernelGGL(rc_prepare_kernel<f16>, grid, dim3(256), 0, (hipStream_t)stream, w, coef, (f16*)rc_panel, rc_c0, Cout, Cin, rows, s.WP, s.KT * 32);
  else
    hipLaunchKernelGGL(rc_prepare_kernel<bf16>, grid, dim3(256), 0, (hipStream_t)stream, w, coef, (bf16*)rc_panel, rc_c0, Cout, Cin, rows, s.WP, s.KT * 32);
  X3D_LAUNCH_CHECK("pw_bwd_rc_prepare");
  return X3D_OK;
}

// ------------------------------------------------------------------------------------------------------------------------
// x3d_bn_bwd_finalize_rc: the BatchNorm-backward finalize of the conv's BN, the panel it feeds (prepare) and the pending
// dW of an EARLIER recomputed-output launch (finish) in ONE launch -- each of them is a ~5 us launch on the critical path of
// the backward pass.  Workgroups [0, nprep): every one derives the coefficients of all C channels into LDS (C <= 223: a few
// loads per thread), workgroup 0 publishes them (coef, dgamma, dbeta), then they split the panel; workgroups [nprep, ..):
// the finish job, independent of this BatchNorm.
// ------------------------------------------------------------------------------------------------------------------------
struct RcFinalizeArgs {
  const double* sums; double count; const float* mi; const float* gamma; float* coef; float* dgamma; float* dbeta; int C;
  const float* w; void* panel; float* c0; int Ci, rows, WP, XR0, nprep;          // prepare (w == NULL: none)
  const float* f_sums; const float* f_w; const float* f_coef; float* f_dw; int f_Co, f_Ci;   // finish (f_sums == NULL: none)
};

#define RCF_C_MAX 224             // channels of the BatchNorm behind the conv (rc_shape: Co <= 223)
#define RCF_W_MAX (RCF_C_MAX * 48)      // weights of one layer staged in LDS (Ci <= 48: rc_shape)
template <typename H>
__global__ __launch_bounds__(256) void rc_finalize_kernel(const RcFinalizeArgs a) {
  __shared__ float cf[RCF_C_MAX * 4];
  // the small operands (weights rounded as the matrix cores see them, the Gram block) in LDS: read from global inside the
  // dot products every term was a dependent L2 round trip (prepare 16 us, finish 14 us per layer against a ~5 us launch floor)
  __shared__ float wl[RCF_W_MAX];
  __shared__ float xl[48 * 48];
  if ((int)blockIdx.x >= a.nprep) {        // finish job
    const int Co = a.f_Co, Ci = a.f_Ci;
    const int i0 = (blockIdx.x - a.nprep) * 256, i = i0 + threadIdx.x;
    const int co0 = i0 / Ci, co1 = min((i0 + 255) / Ci, Co - 1);      // the weight rows this workgroup's outputs touch
    const float* sx = a.f_sums + (long long)Co * Ci;
    const float* xx = a.f_sums + (long long)(Co + 1) * Ci;
    for (int j = threadIdx.x; j < (co1 - co0 + 1) * Ci; j += 256) wl[j] = round_to<H>(a.f_w[(long long)co0 * Ci + j]);
    for (int j = threadIdx.x; j < Ci * Ci; j += 256) xl[j] = xx[j];
    __syncthreads();
    if (i >= Co * Ci) return;
    const int co = i / Ci, ci = i - co * Ci;
    double acc = 0.0;
    for (int k = 0; k < Ci; k++) acc += (double)wl[(co - co0) * Ci + k] * (double)xl[k * Ci + ci];
    const double v = (double)a.f_coef[co * 4] * (double)a.f_sums[i] + (double)a.f_coef[co * 4 + 1] * acc + (double)a.f_coef[co * 4 + 2] * (double)sx[ci];
    a.f_dw[i] += (float)v;
    return;
  }
  // the finalize (the arithmetic of bn_bwd_finalize_kernel, elem.hip), every prepare workgroup for itself
  for (int c = threadIdx.x; c < a.C; c += 256) {
    const double mean = a.mi[c * 2], invstd = a.mi[c * 2 + 1];
    const double dbe = a.sums[c * 2];
    const double dga = (a.sums[c * 2 + 1] - mean * dbe) * invstd;
    const double k1 = (double)a.gamma[c] * invstd;
    const double B = -k1 * invstd * dga / a.count;
    const float fA = (float)k1, fB = (float)B, fC = (float)(-k1 * dbe / a.count - B * mean);
    if (a.w) { cf[c * 4] = fA; cf[c * 4 + 1] = fB; cf[c * 4 + 2] = fC; cf[c * 4 + 3] = 0.f; }
    if (blockIdx.x == 0) {
      a.coef[c * 4] = fA; a.coef[c * 4 + 1] = fB; a.coef[c * 4 + 2] = fC; a.coef[c * 4 + 3] = 0.f;
      a.dgamma[c] += (float)dga;
      a.dbeta[c] += (float)dbe;
    }
  }
  if (!a.w) return;
  for (int j = threadIdx.x; j < a.C * a.Ci; j += 256) wl[j] = a.w[j];      // (rc_prepare_body rounds: the same bits as from global)
  __syncthreads();
  rc_prepare_body<H>(wl, cf, (H*)a.panel, a.c0, a.C, a.Ci, a.rows, a.WP, a.XR0, blockIdx.x * 256 + threadIdx.x, a.nprep * 256);
}

extern "C" int x3d_bn_bwd_finalize_rc(const double* sums, double count, const float* mean_invstd, const float* gamma, float* coef,
                                      float* dgamma, float* dbeta, int C, const float* w, void* rc_panel, float* rc_c0, int Cin,
                                      const float* fin_sums, const float* fin_w, const float* fin_coef, float* fin_dw, int fin_Cout,
                                      int fin_Cin, int dtype, void* stream) {
  X3D_REQUIRE(sums && mean_invstd && gamma && coef && dgamma && dbeta && C > 0 && count > 0, "bn_bwd_finalize_rc: bad args");
  X3D_REQUIRE(x3d_is_half(dtype), "bn_bwd_finalize_rc: 16-bit storage types only");
  RcFinalizeArgs a;
  memset(&a, 0, sizeof(a));
  a.sums = sums; a.count = count; a.mi = mean_invstd; a.gamma = gamma; a.coef = coef; a.dgamma = dgamma; a.dbeta = dbeta; a.C = C;
  a.nprep = 1;
  if (w) {
    X3D_REQUIRE(rc_panel && rc_c0 && Cin > 0 && C <= RCF_C_MAX - 1, "bn_bwd_finalize_rc: prepare needs the panel, c0 and C <= 223");
    RcShape s;
    X3D_REQUIRE(rc_shape(Cin, C, &s), "bn_bwd_finalize_rc: layer shape not covered (x3d_pw_bwd_rc_panel_elems() == 0)");
    a.w = w; a.panel = rc_panel; a.c0 = rc_c0; a.Ci = Cin; a.rows = s.MT * 32; a.WP = s.WP; a.XR0 = s.KT * 32;
    a.nprep = ceil_div(a.rows * a.WP, 256 * 4);          // four panel entries per thread: the 24 -> 54 panel in 4 workgroups
  } else {
    X3D_REQUIRE(C <= 256 * 1024, "bn_bwd_finalize_rc: too many channels");
  }
  int nfin = 0;
  if (fin_sums) {
    X3D_REQUIRE(fin_w && fin_coef && fin_dw && fin_Cout > 0 && fin_Cin > 0 && fin_Cin <= 48, "bn_bwd_finalize_rc: finish job incomplete / Cin > 48");
    a.f_sums = fin_sums; a.f_w = fin_w; a.f_coef = fin_coef; a.f_dw = fin_dw; a.f_Co = fin_Cout; a.f_Ci = fin_Cin;
    nfin = ceil_div(fin_Cout * fin_Cin, 256);
  }
  const dim3 grid((unsigned)(a.nprep + nfin));
  if (dtype == X3D_F16) hipLaunchKernelGGL(rc_finalize_kernel<f16>, grid, dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(rc_finalize_kernel<bf16>, grid, dim3(256), 0, (hipStream_t)stream, a);
  X3D_LAUNCH_CHECK("bn_bwd_finalize_rc");
  return X3D_OK;
}

// dW[co][ci] += A[co] * S[co][ci] + B[co] * sum_k Wr[co][k] * XX[k][ci] + C[co] * sx[ci]        (fp64)
template <typename H>
__global__ __launch_bounds__(256) void rc_finish_kernel(const float* __restrict__ sums, const float* __restrict__ w,
                                                        const float* __restrict__ coef, float* __restrict__ dw, int Co, int Ci) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Co * Ci) return;
  const int co = i / Ci, ci = i - co * Ci;
  const float* sx = sums + (long long)Co * Ci;
  const float* xx = sums + (long long)(Co + 1) * Ci;
  double acc = 0.0;
  for (int k = 0; k < Ci; k++) acc += (double)round_to<H>(w[(long long)co * Ci + k]) * (double)xx[(long long)k * Ci + ci];
  const double v = (double)coef[co * 4] * (double)sums[i] + (double)coef[co * 4 + 1] * acc + (double)coef[co * 4 + 2] * (double)sx[ci];
  dw[i] += (float)v;
}

extern "C" int x3d_pw_bwd_rc_finish(const float* rc_sums, const float* w, const float* coef, float* dw, int Cout, int Cin, int dtype,
                                    void* stream) {
  X3D_REQUIRE(rc_sums && w && coef && dw && Cout > 0 && Cin > 0, "pw_bwd_rc_finish: bad args");
  X3D_REQUIRE(x3d_is_half(dtype), "pw_bwd_rc_finish: 16-bit storage types only");
  const dim3 grid((unsigned)ceil_div(Cout * Cin, 256));
  if (dtype == X3D_F16) hipLaunchKernelGGL(rc_finish_kernel<f16>, grid, dim3(256), 0, (hipStream_t)stream, rc_sums, w, coef, dw, Cout, Cin);
  else hipLaunchKernelGGL(rc_finish_kernel<bf16>, grid, dim3(256), 0, (hipStream_t)stream, rc_sums, w, coef, dw, Cout, Cin);
  X3D_LAUNCH_CHECK("pw_bwd_rc_finish");
  return X3D_OK;
}
