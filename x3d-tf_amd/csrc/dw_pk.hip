// Channelwise 3x3x3 convolution, fused backward, PACKED variant for small stride-1 planes (rows of 10..18 outputs:
// the 14x14 layers of X3D-M stage 4, 10x10 of X3D-S / L / XL).
//
// Why.  dw3d_bwd_pd_s1_kernel (dw_pd.hip) gives one (n, c) plane to one workgroup of 128 threads: strips of two outputs,
// 98 working threads at 14x14, two barriers per plane.  Measured on MI355X (216 channels x 64 clips x 16 planes of 14x14)
// with every global access switched off it keeps 85 % of its run time: the kernel is bound inside the CU, by the VALU
// (54 FMAs per output + ~45 % staging / emit / addressing instructions) and by the LDS taking turns between barriers --
// the windows were read with ds_read2_b64 (128 B/clk, half the rate of b128) at a lane stride of 8 bytes with 2-way bank
// conflicts (SQ_LDS_BANK_CONFLICT = 64 % on top of the conflict-free cycles).  Here
//   * strips of FOUR outputs (the last strip of a 14-wide row is half empty): windows are one ds_read_b128 + one
//     ds_read_b64 per row at a pitch of 16 floats -- the 16 lanes of a b128 group cover 64 distinct banks; a third of the
//     LDS cycles per output, 18 % fewer VALU instructions per output;
//   * ONE workgroup takes NP planes of the SAME channel c and consecutive samples n (14x14: 8 planes x 56 threads = 7 full
//     waves): weights / BN_a coefficients stay wave-uniform (SGPRs), the 27 dW partials and the two BN_a sums are reduced
//     over the samples before the atomics (8x fewer of them);
//   * the LDS planes are DOUBLE BUFFERED: plane t+1 is staged while plane t is read, one barrier per plane instead of two;
//   * a thread stages exactly its own strip, so (a) the araw strip it needs again when its dA plane is emitted (ReLU mask,
//     BN_a sums) is the staged vector itself -- three loads per thread and plane -- and (b) its own dB strip of plane t+1
//     is in registers one plane early: the weight gradient's three temporal taps are dB[t-1], dB[t], dB[t+1] against ONE
//     window of A[t] (the previous plane's windows need not be kept: 18 VGPRs).
// Sums are taken in a different order than in dw3d_bwd_pd_s1_kernel (strips of 4, taps regrouped): ga / dw / a_sums agree
// with it to fp32 rounding, and with the fp64 oracle within the same tolerances (tests/test_kernels_gpu.py::test_dw3d_bwd).
#include "dw_common.h"

#define PK_MAX_THREADS 512
// Buffer offsets are voffset (per thread) + soffset (per plane) from the TENSOR base.  Tensors are < 1 GB (host check), an
// inactive thread has voffset = DW_OOB (2^30) and a plane past T has soffset = DW_OOB: every combination lands in
// [2^30, 2^31], past num_records -- the load returns zeros / the store is dropped without touching memory, no 32-bit wrap.

struct DwPkArgs {
  DwBwdArgs b;
  int NP;          // planes (samples) per workgroup
  int items;       // threads per plane = H * ceil(W / SW)
  int ngroups;     // ceil(N / NP)
  int LP;          // LDS row pitch (floats) of both tiles
  unsigned bytes;  // whole-tensor size (buffer num_records; stride 1: input and output tensors have the same extents)
  int noload;      // diagnostics (X3D_DW_PK_NOLOAD=1): every global access out of range -> the compute-only time
};

// LPC / HC: LDS row pitch and plane height as compile-time constants (0: run-time values from the arguments).  With both
// known every LDS address of the loop is ONE base register + an immediate offset (tiles, buffers and window rows are
// constant distances apart); with run-time values the loop carries ~20 address registers and spills at the 128-VGPR cap.
// ODD: rows of an odd number of elements (7x7 planes, strips 4 + 3): strips start at any element, loads are unaligned
// (2-byte aligned 8-byte buffer loads: the compute queues run in unaligned-access mode), a strip's last half may hold ONE
// valid element and is then stored as a single element.  The LOAD of a short last strip starts `shift` elements early
// (columns W-4 .. W-1) so that it never leaves its row: a vector load that straddles the end of the tensor is dropped
// whole by the bounds check, valid elements included.
// DOT (experiment, X3D_DW_DOT=1; bit 1: dW, bit 2: dA): the tap loops on two-element dot products (dw_common.h, Dot2): dW
// pairs adjacent outputs of the strip, dA pairs the window rows kh = 0, 1 (weights as packed pairs in SGPRs; the kh = 2 row
// stays scalar fp32): 216 FMAs -> 126 dot2 / FMA + 23 pair conversions per thread and plane of 4 outputs.  Not faster
// (dw_common.h, dw_use_dot): off by default.
template <typename T, int SW, int PD, int UN, int LPC, int HC, bool ODD, int DOT = 0>
__global__ __launch_bounds__(PK_MAX_THREADS, 4) void dw3d_bwd_pk_kernel(const DwPkArgs pa) {
  static_assert(!(DOT & 3) || (Dot2<T>::ok && SW == 4), "dot2 form: 16-bit storage, strips of 4");
  static_assert(UN % 6 == 0 && UN % PD == 0, "roles have periods 2 (LDS buffers) and 3 (planes); slots period PD");
  static_assert(SW == 2 || SW == 4, "strips of 2 or 4 outputs");
  constexpr int WIN = SW + 2;
  constexpr int EB = (int)sizeof(T);
  constexpr int NH = SW / 2;                     // a strip is stored as NH halves of two elements (the last strip of a row may be half)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const DwBwdArgs& a = pa.b;
  const DwGeom& g = a.g;
  const int LP = LPC ? LPC : pa.LP;
  const int tile = ((HC ? HC : g.H) + 2) * LP;   // one tile: rows -1 .. H, columns -1 .. (zero halo)
  const int pplane = 2 * tile;                   // A tile | dB tile
  float* scratch = lds + pa.NP * 2 * pplane + 8; // (+8: the last window of the last tile reads 2 floats past its row)

  const int c = __builtin_amdgcn_readfirstlane(blockIdx.x % g.C);
  const int grp = __builtin_amdgcn_readfirstlane(blockIdx.x / g.C);
  const int nstr = (g.W + SW - 1) / SW;
  const int p = threadIdx.x / pa.items, rem = threadIdx.x - p * pa.items;
  const int r = rem / nstr, sidx = rem - r * nstr;
  const int n = grp * pa.NP + p;
  const bool active = p < pa.NP && n < g.N;
  const int ncol = min(SW, g.W - SW * sidx);     // valid outputs of this strip (W even: 2 or 4)
  bool okc[SW];
#pragma unroll
  for (int i = 0; i < SW; i++) okc[i] = active && i < ncol;

  for (int i = threadIdx.x; i < pa.NP * 2 * pplane + 8; i += blockDim.x) lds[i] = 0.f;   // halos stay zero for good

  float wgt[27];
#pragma unroll
  for (int k = 0; k < 27; k++) wgt[k] = a.w[c * 27 + k];
  uint32_t wgtP[DOT ? 9 : 1];          // DOT: (w[kt][0][kw], w[kt][1][kw]) as a storage-type pair
  if constexpr ((DOT & 2) != 0) {
#pragma unroll
    for (int k = 0; k < 9; k++)   // (uniform: kept in SGPRs)
      wgtP[k] = __builtin_amdgcn_readfirstlane(Dot2<T>::pk(wgt[(k / 3) * 9 + (k % 3)], wgt[(k / 3) * 9 + 3 + (k % 3)]));
  }
  const float sc = a.ss_a[c * 2], sh = a.ss_a[c * 2 + 1];
  float cA = 0.f, cB = 0.f, cC = 0.f;
  if (active) {
    const float* cf = a.coef_nc + ((long long)n * g.C + c) * 4;
    cA = cf[0]; cB = cf[1]; cC = cf[2];
  }

  const int planeB = g.H * g.W * EB;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((T*)a.araw, 0, pa.bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc((T*)a.ga, 0, pa.bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc((T*)a.dv, 0, pa.bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((T*)a.braw, 0, pa.bytes, 0x00020000);
  // byte offset of this thread's strip in plane 0 of its (n, c) channel (a strip of 4 at the end of a 14-wide row reads two
  // elements of the next row: loaded, never used)
  const int voff = active ? (int)((((long long)n * g.C + c) * g.T * g.H * g.W + r * g.W + SW * sidx) * EB) : DW_OOB;
  int voffH[NH], voff1[ODD ? NH : 1];            // store offsets of the halves (a half outside the row is dropped; ODD: a half
#pragma unroll                                   // with one valid element is stored through voff1 as a single element)
  for (int h = 0; h < NH; h++) {
    voffH[h] = (active && 2 * h + 1 < ncol + (ODD ? 0 : 1)) ? voff + 2 * h * EB : DW_OOB;
    if constexpr (ODD) voff1[h] = (active && 2 * h + 1 == ncol) ? voff + 2 * h * EB : DW_OOB;
  }
  const int shift = ODD ? SW - ncol : 0;          // elements the load starts early (0, or 1 for the 3-element strip of a 7-wide row)
  const int voffL = (ODD && active) ? voff - shift * EB : voff;
  // a loaded vector with output column 0 of the strip in element 0: one 64-bit shift by 16 * shift bits (ODD, 16-bit storage)
  auto aligned = [&](const Raw& rw) -> Raw {
    if constexpr (ODD) {
      static_assert(!ODD || (EB == 2 && SW == 4), "odd rows: 16-bit storage, strips of 4");
      const unsigned long long v = (((unsigned long long)rw.w[1] << 32) | rw.w[0]) >> (16 * shift);
      Raw o; o.w[0] = (unsigned)v; o.w[1] = (unsigned)(v >> 32); o.w[2] = o.w[3] = 0u;
      return o;
    } else {
      return rw;
    }
  };
  // LDS: plane p, buffer q at lds + (2 p + q) * pplane.  Image pixel (h, w) sits at row h + 1, column w + 1.
  float* myA = lds + (2 * (active ? p : 0)) * pplane;   // (idle threads read plane 0's windows and discard them)
  const int lS = (r + 1) * LP + 1 + SW * sidx;   // where the own strip is staged
  const int lW = r * LP + SW * sidx;             // window origin: rows r .. r+2, columns SW*sidx .. +WIN-1 (16-byte aligned for SW = 4)

  struct Slot { Raw A, D, R; };
  Slot slot[PD];
  auto issue = [&](int t, Slot& s) {
    const int soff = (t < g.T && !pa.noload) ? t * planeB : DW_OOB;      // wave-uniform: planes past T move no data
    raw_bload<SW * EB>(s.A, rsA, voffL, soff);
    raw_bload<SW * EB>(s.D, rsD, voffL, soff);
    raw_bload<SW * EB>(s.R, rsR, voffL, soff);
  };
  // plane in slot s -> LDS buffer q; returns the thread's own dB strip (zeros outside the image)
  auto stage = [&](const Slot& s0, int q, float (&dBown)[SW], bool plane_valid) {   // plane_valid: the plane exists (< T)
    Slot s;
    s.A = aligned(s0.A); s.D = aligned(s0.D); s.R = aligned(s0.R);
    float* A = myA + q * pplane;
    float* B = A + tile;
#pragma unroll
    for (int e = 0; e < SW; e++) {
      const float av = fmaxf(sc * raw_get<T>(s.A, e) + sh, 0.f);
      const float bv = cA * raw_get<T>(s.D, e) + cB * raw_get<T>(s.R, e) + cC;
      dBown[e] = (okc[e] && plane_valid) ? bv : 0.f;   // (a plane past T loads zeros, but bv would be the constant C)
      if (okc[e]) { A[lS + e] = av; B[lS + e] = bv; }
    }
  };

  float dAr[3][SW];            // dAr[p % 3] = gradient plane p while it is accumulated / waits for its emit (dw_pd.hip)
#pragma unroll
  for (int k = 0; k < 3; k++)
#pragma unroll
    for (int i = 0; i < SW; i++) dAr[k][i] = 0.f;
  float dW[27];
#pragma unroll
  for (int k = 0; k < 27; k++) dW[k] = 0.f;
  float dBs[3][SW];            // own dB strips: dBs[p % 3] = plane p (planes t-1, t, t+1 during iteration t)
  uint32_t dBp[3][DOT ? 2 : 1]; // DOT: the same strips as pairs (0, 1), (2, 3) of storage-type values
#pragma unroll
  for (int k = 0; k < 3; k++) {
#pragma unroll
    for (int i = 0; i < SW; i++) dBs[k][i] = 0.f;
    dBp[k][0] = 0u;
    if constexpr (DOT) dBp[k][1] = 0u;
  }
  float s1 = 0.f, s2 = 0.f;
  Raw own0, own1, own2;         // araw strips: at the top of iteration t of planes t, t-1, t-2 (own0 = the one staged last)
  own0.w[0] = own0.w[1] = own0.w[2] = own0.w[3] = 0u;
  own1 = own0; own2 = own0;

  auto emit = [&](int t, bool live, const float (&v)[SW], const Raw& own) {   // !live: nothing stored / summed
    float gv[SW];
#pragma unroll
    for (int i = 0; i < SW; i++) {
      const float av = raw_get<T>(own, i);
      gv[i] = (live && okc[i] && sc * av + sh > 0.f) ? v[i] : 0.f;
      s1 += gv[i];
      s2 += gv[i] * av;
    }
    const int soff = (live && !pa.noload) ? t * planeB : DW_OOB;
#pragma unroll
    for (int h = 0; h < NH; h++) {
      const float two[2] = {gv[2 * h], gv[2 * h + 1]};
      Raw o;
      raw_pack<T, 2>(o, two);
      raw_bstore<2 * EB>(o, rsG, voffH[h], soff);
      if constexpr (ODD) raw_bstore<EB>(o, rsG, voff1[h], soff);   // element 2h alone (the low half of the packed pair)
    }
  };
  auto dummy_stores = [&]() {   // the store slots of an iteration, dropped: keeps the vmcnt pattern of the prologue = the loop's
    Raw z; z.w[0] = z.w[1] = z.w[2] = z.w[3] = 0u;
#pragma unroll
    for (int h = 0; h < NH; h++) {
      raw_bstore<2 * EB>(z, rsG, voffH[h], DW_OOB);
      if constexpr (ODD) raw_bstore<EB>(z, rsG, voff1[h], DW_OOB);
    }
  };

  // prologue: PD planes in flight
#pragma unroll
  for (int d = 0; d < PD; d++) {
    issue(d, slot[d]);
    dummy_stores();
  }
  __syncthreads();                       // zero fill done
  stage(slot[0], 0, dBs[0], true);
  if constexpr ((DOT & 1) != 0) { dBp[0][0] = Dot2<T>::pk(dBs[0][0], dBs[0][1]); dBp[0][1] = Dot2<T>::pk(dBs[0][2], dBs[0][3]); }
  own0 = aligned(slot[0].A);
  issue(PD, slot[0]);
  dummy_stores();
  __syncthreads();                       // plane 0 visible in buffer 0

  // iteration t: windows of plane t from buffer t & 1; plane t + 1 staged into the other buffer; ONE barrier
  for (int t0 = 0; t0 < g.T; t0 += UN) {
#pragma unroll
    for (int d = 0; d < UN; d++) {
      const int t = t0 + d;
      if (t >= g.T) break;
      const int cur = d & 1, prv = cur ^ 1;                        // compile-time after unrolling (t0 % UN == 0)
      const int sl = (d + 1) % PD;                                 // slot of plane t + 1
      const int pm1 = (d + 2) % 3, p0 = d % 3, pp1 = (d + 1) % 3;  // roles of planes t-1, t, t+1
      float winA[3][WIN], winB[3][WIN];
      {
        const float* A = myA + cur * pplane + lW;
        const float* B = A + tile;
#pragma unroll
        for (int kh = 0; kh < 3; kh++) lds_window<WIN, SW>(A + kh * LP, winA[kh]);
#pragma unroll
        for (int kh = 0; kh < 3; kh++) lds_window<WIN, SW>(B + kh * LP, winB[kh]);
      }
      // plane t-2 was completed at the end of iteration t-1; its accumulator is taken over by plane t+1 now
      emit(t - 2, t >= 2, dAr[pp1], own2);
#pragma unroll
      for (int i = 0; i < SW; i++) dAr[pp1][i] = 0.f;
      // the next plane: registers -> the other LDS buffer (its last readers passed the barrier of iteration t - 1); its own
      // dB strip replaces plane t-2's
      own2 = own1; own1 = own0;
      stage(slot[sl], prv, dBs[pp1], t + 1 < g.T);
      own0 = aligned(slot[sl].A);
      issue(t + 1 + PD, slot[sl]);
      if constexpr ((DOT & 1) != 0) {
        dBp[pp1][0] = Dot2<T>::pk(dBs[pp1][0], dBs[pp1][1]);
        dBp[pp1][1] = Dot2<T>::pk(dBs[pp1][2], dBs[pp1][3]);
        // weight gradient: pairs (i, i+1) of the strip against the window pairs (c, c+1), c = i + kw
#pragma unroll
        for (int kh = 0; kh < 3; kh++) {
          uint32_t Ap[WIN - 1];
#pragma unroll
          for (int cc = 0; cc < WIN - 1; cc++) Ap[cc] = Dot2<T>::pk(winA[kh][cc], winA[kh][cc + 1]);
#pragma unroll
          for (int kw = 0; kw < 3; kw++)
#pragma unroll
            for (int h = 0; h < 2; h++) {
              dW[kh * 3 + kw] = Dot2<T>::dot(dBp[pp1][h], Ap[2 * h + kw], dW[kh * 3 + kw]);
              dW[9 + kh * 3 + kw] = Dot2<T>::dot(dBp[p0][h], Ap[2 * h + kw], dW[9 + kh * 3 + kw]);
              dW[18 + kh * 3 + kw] = Dot2<T>::dot(dBp[pm1][h], Ap[2 * h + kw], dW[18 + kh * 3 + kw]);
            }
        }
      } else {
        // weight gradient: one window of A[t] against the own strips dB[t+1], dB[t], dB[t-1] (temporal taps 0, 1, 2)
#pragma unroll
        for (int kh = 0; kh < 3; kh++)
#pragma unroll
          for (int kw = 0; kw < 3; kw++) {
#pragma unroll
            for (int i = 0; i < SW; i++) {
              const float av = winA[kh][i + kw];
              dW[kh * 3 + kw] += dBs[pp1][i] * av;
              dW[9 + kh * 3 + kw] += dBs[p0][i] * av;
              dW[18 + kh * 3 + kw] += dBs[pm1][i] * av;
            }
          }
      }
      if constexpr ((DOT & 2) != 0) {
        // data gradient: window rows 2 and 1 (taps kh = 0, 1) as pairs against the weight pairs, row 0 (kh = 2) scalar
        uint32_t Bp[WIN];
#pragma unroll
        for (int cc = 0; cc < WIN; cc++) Bp[cc] = Dot2<T>::pk(winB[2][cc], winB[1][cc]);
#pragma unroll
        for (int kw = 0; kw < 3; kw++)
#pragma unroll
          for (int i = 0; i < SW; i++) {
            const uint32_t bp = Bp[i + 2 - kw];
            const float v = winB[0][i + 2 - kw];
            dAr[pm1][i] = Dot2<T>::dot(wgtP[kw], bp, dAr[pm1][i]);
            dAr[p0][i] = Dot2<T>::dot(wgtP[3 + kw], bp, dAr[p0][i]);
            dAr[pp1][i] = Dot2<T>::dot(wgtP[6 + kw], bp, dAr[pp1][i]);
            dAr[pm1][i] += wgt[6 + kw] * v;
            dAr[p0][i] += wgt[15 + kw] * v;
            dAr[pp1][i] += wgt[24 + kw] * v;
          }
      } else {
        // data gradient: the dB[t] windows scattered into the planes t-1, t, t+1
#pragma unroll
        for (int kh = 0; kh < 3; kh++)
#pragma unroll
          for (int kw = 0; kw < 3; kw++)
#pragma unroll
            for (int i = 0; i < SW; i++) {
              const float v = winB[2 - kh][i + 2 - kw];
              dAr[pm1][i] += wgt[kh * 3 + kw] * v;
              dAr[p0][i] += wgt[9 + kh * 3 + kw] * v;
              dAr[pp1][i] += wgt[18 + kh * 3 + kw] * v;
            }
      }
      __syncthreads();
    }
  }
  // after the loop: own1 = strip of plane T-1, own2 = plane T-2 (own0 belongs to the never-used plane T)
  {
    const int m2 = (g.T + 1) % 3, m1 = (g.T + 2) % 3;   // (T-2) % 3, (T-1) % 3 for T >= 1
    float v2[SW], v1[SW];
#pragma unroll
    for (int i = 0; i < SW; i++) {
      v2[i] = m2 == 0 ? dAr[0][i] : (m2 == 1 ? dAr[1][i] : dAr[2][i]);
      v1[i] = m1 == 0 ? dAr[0][i] : (m1 == 1 ? dAr[1][i] : dAr[2][i]);
    }
    emit(g.T - 2, g.T >= 2, v2, own2);
    emit(g.T - 1, true, v1, own1);
  }

  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  float red[29];
#pragma unroll
  for (int k = 0; k < 29; k++) red[k] = wave_sum_lane63(k < 27 ? dW[k] : (k == 27 ? s1 : s2));
  if (lane == 63) {
#pragma unroll
    for (int k = 0; k < 29; k++) scratch[k * 8 + wid] = red[k];
  }
  __syncthreads();
  if (threadIdx.x < 29) {
    float v = 0.f;
    for (int w = 0; w < nw; w++) v += scratch[threadIdx.x * 8 + w];
    if (threadIdx.x < 27) atomicAdd(&a.dw[c * 27 + threadIdx.x], v);
    else atomic_add_d(&a.a_sums[c * 2 + (threadIdx.x - 27)], (double)v);
  }
}

// A/B switches: X3D_DW_PK=0 never use the packed kernel; X3D_DW_PK_SW=2|4 force the strip width (default: 4 from 12-wide rows)
static int pk_env(const char* name, int dflt) { return x3d_env_int(name, dflt); }

template <typename T, int SW>
static bool bwd_pk_t(const DwBwdArgs& a, hipStream_t st) {
  const DwGeom& g = a.g;
  DwPkArgs pa;
  pa.b = a;
  const int nstr = ceil_div(g.W, SW);
  pa.items = g.H * nstr;
  // Planes per workgroup.  Measured at 14x14 (216 ch x 64 clips, strips of 4): 1 plane = one wave per workgroup 138 us,
  // 2 planes 141, 4: 153, 8 (7 full waves): 161 = no better than the kernel this replaces.  Waves that share barriers move
  // through their LDS phase and their FMA phase together and the two pipes take turns; sixteen independent one-wave
  // workgroups per CU spread over all phases.  So: as many planes as fit ONE wave (7x7-class planes: several), not more.
  pa.NP = 64 / pa.items > 0 ? 64 / pa.items : 1;
  const int np_env = pk_env("X3D_DW_PK_NP", 0);      // A/B hook: planes per workgroup
  if (np_env > 0 && np_env * pa.items <= PK_MAX_THREADS) pa.NP = np_env;
  if (pa.NP > g.N) pa.NP = g.N;
  pa.ngroups = ceil_div(g.N, pa.NP);
  // row pitch: columns -1 .. nstr*SW.  13..16-wide strips rows: pitch 16 = conflict-free b128 windows (the last window's two
  // extra floats are the next row's first two: they only feed outputs past the end of the row)
  pa.LP = (nstr * SW + 2 + 3) & ~3;
  if (SW == 4 && nstr * SW == 16 && g.W + 2 <= 16) pa.LP = 16;
  pa.bytes = (unsigned)((long long)g.N * g.C * g.T * g.H * g.W * (long long)sizeof(T));
  const int threads = ceil_div(pa.NP * pa.items, 64) * 64;
  const size_t lds = ((size_t)pa.NP * 2 * 2 * (g.H + 2) * pa.LP + 8 + 29 * 8 + 8) * sizeof(float);
  if (lds > 64 * 1024 || threads > PK_MAX_THREADS) return false;
  // compile-time geometry for the planes X3D has: 14x14 (M stage 4; pitch 16), 10x10 (S / L / XL; pitch 12), 7x7 (M stage 5)
  const bool fixed = (SW == 4 && pa.LP == 16 && g.H == 14) || (SW == 2 && pa.LP == 12 && g.H == 10);
  const bool odd7 = SW == 4 && g.W == 7 && g.H == 7 && pa.LP == 12;
  if ((g.W & 1) && !odd7) return false;
  // two-element dot products (16-bit storage, strips of 4, compile-time geometry)
  const bool dot = SW == 4 && (fixed || odd7) && dw_use_dot(sizeof(T) == 4 ? X3D_F32 : (TypeName<T>::v[0] == 'b' ? X3D_BF16 : X3D_F16));
  if (x3d_describe.out) {
    // (all eight template arguments, as the symbol carries them: tools/dw_gbs.py joins this string with the rocprofv3 table)
    snprintf(x3d_describe.out, x3d_describe.cap, "dw3d_bwd_pk_kernel<%s, %d, 2, 6, %d, %d, %d, %d>", TypeName<T>::v, SW,
             (fixed || odd7) ? pa.LP : 0, (fixed || odd7) ? g.H : 0, (int)odd7, dot ? (odd7 ? 3 : pk_env("X3D_DW_DOTMASK", 3)) : 0);
    return true;
  }
  pa.noload = pk_env("X3D_DW_PK_NOLOAD", 0) == 1;   // result-changing timing hook: -DX3D_EXPERIMENTS builds only
  auto kern = fixed ? dw3d_bwd_pk_kernel<T, SW, 2, 6, (SW == 4 ? 16 : 12), (SW == 4 ? 14 : 10), false>
                    : dw3d_bwd_pk_kernel<T, SW, 2, 6, 0, 0, false>;
  if constexpr (SW == 4 && sizeof(T) == 2) {
    const int mask = pk_env("X3D_DW_DOTMASK", 3);
    if (odd7) kern = dot ? dw3d_bwd_pk_kernel<T, 4, 2, 6, 12, 7, true, 3> : dw3d_bwd_pk_kernel<T, 4, 2, 6, 12, 7, true>;
    else if (fixed && dot) kern = mask == 1 ? dw3d_bwd_pk_kernel<T, 4, 2, 6, 16, 14, false, 1> : mask == 2 ? dw3d_bwd_pk_kernel<T, 4, 2, 6, 16, 14, false, 2> : dw3d_bwd_pk_kernel<T, 4, 2, 6, 16, 14, false, 3>;
  }
  if (lds > 48 * 1024) {
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
      attr_set = true;
    }
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(g.C * pa.ngroups)), dim3(threads), lds, st, pa);
  return true;
}

// Covered: stride 1, rows of 10..18 outputs (the caller's strips of 2), the whole plane in one tile, even W (a strip = aligned
// pairs of elements), at least two planes per workgroup, tensors below 1 GB (32-bit buffer offsets from the tensor base).
bool dw_bwd_pk_launch(const DwBwdArgs& a, int dtype, int S, int SW_caller, hipStream_t st) {
  const DwGeom& g = a.g;
  if (pk_env("X3D_DW_PK", 1) == 0 || S != 1 || g.ntile_h != 1 || g.pw != 1 || g.ph != 1) return false;
  const bool odd7 = dtype != X3D_F32 && g.W == 7 && g.H == 7 && pk_env("X3D_DW_PK7", 1) == 1;   // 7x7: strips 4 + 3, four planes per wave
  if ((g.W % 2) != 0 && !odd7) return false;
  // strips of 4 where a compile-time-geometry instantiation exists and fits 128 VGPRs (16-bit storage, 14x14); the
  // run-time-geometry / fp32 strips-of-4 instantiations spill (22 / 86 VGPRs): strips of 2 there (rows up to 18 wide).
  // 28x28 planes (whole plane = 196 threads = four waves per workgroup) were measured with this kernel too: 316 us against
  // 297 us of dw3d_bwd_kernel (108 ch x 64 clips) -- with all global accesses off still 244 us: the fused backward costs
  // ~2.6 ps of VALU time per output whatever the tiling (54 FMAs + ~30 other instructions per output at ~55 % issue
  // utilisation), which is what bounds every stride-1 layer, not HBM.
  const bool fixed4 = dtype != X3D_F32 && g.W == 14 && g.H == 14;
  if (SW_caller != 2 && !odd7) return false;
  const int SW = odd7 ? 4 : pk_env("X3D_DW_PK_SW", fixed4 ? 4 : 2);
  if (SW != 2 && SW != 4) return false;
  const int items = g.H * ceil_div(g.W, SW);
  if (items > PK_MAX_THREADS) return false;
  const int eb = dtype == X3D_F32 ? 4 : 2;
  const long long bytes = (long long)g.N * g.C * g.T * g.H * g.W * eb;
  if (bytes >= (1ll << 30)) return false;
  const uintptr_t al = (uintptr_t)(2 * eb) - 1;
  if (((uintptr_t)a.araw & al) || ((uintptr_t)a.ga & al) || ((uintptr_t)a.dv & al) || ((uintptr_t)a.braw & al)) return false;
  if ((long long)g.C * g.N >= (1ll << 31)) return false;
#define PK_GO(TT) (SW == 4 ? bwd_pk_t<TT, 4>(a, st) : bwd_pk_t<TT, 2>(a, st))
  return dtype == X3D_BF16 ? PK_GO(bf16) : dtype == X3D_F16 ? PK_GO(f16) : PK_GO(float);
#undef PK_GO
}


// ================================================================================================
// FORWARD, same organisation: one wave per workgroup, NP planes of one channel per wave, strips of 4 (2), a thread
// stages exactly its own strip (ONE load and one store per thread and plane), double-buffered LDS tile, windows read
// as ds_read_b128 + b64 at a conflict-free pitch.  The tap loop is dw_taps_row / dw_rotate of the other forward
// kernels (dw_common.h): same products in the same order, bit-identical outputs (tools/ab_dw.py).
// ================================================================================================
struct DwPkFwdArgs {
  DwFwdArgs f;
  int NP, items, ngroups, LP;
  unsigned in_bytes;
  int noload;
};

template <typename T, int SW, int PD, int UN, int LPC, int HC, bool ODD>
__global__ __launch_bounds__(PK_MAX_THREADS, 4) void dw3d_fwd_pk_kernel(const DwPkFwdArgs pa) {
  static_assert(UN % 2 == 0 && UN % PD == 0, "LDS buffers alternate; slots period PD");
  constexpr int WIN = SW + 2;
  constexpr int EB = (int)sizeof(T);
  constexpr int NH = SW / 2;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const DwFwdArgs& a = pa.f;
  const DwGeom& g = a.g;
  const int LP = LPC ? LPC : pa.LP;
  const int tile = ((HC ? HC : g.H) + 2) * LP;
  float* scratch = lds + pa.NP * 2 * tile + 8;   // [NP][2] per-plane sums

  const int c = __builtin_amdgcn_readfirstlane(blockIdx.x % g.C);
  const int grp = __builtin_amdgcn_readfirstlane(blockIdx.x / g.C);
  const int nstr = (g.W + SW - 1) / SW;
  const int p = threadIdx.x / pa.items, rem = threadIdx.x - p * pa.items;
  const int r = rem / nstr, sidx = rem - r * nstr;
  const int n = grp * pa.NP + p;
  const bool active = p < pa.NP && n < g.N;
  const int ncol = min(SW, g.W - SW * sidx);
  bool okc[SW];
#pragma unroll
  for (int i = 0; i < SW; i++) okc[i] = active && i < ncol;

  for (int i = threadIdx.x; i < pa.NP * 2 * tile + 8 + pa.NP * 2; i += blockDim.x) lds[i] = 0.f;

  v2f w21[3][3];
  float w0[3][3];
#pragma unroll
  for (int k = 0; k < 9; k++) {
    w21[k / 3][k % 3] = (v2f){a.w[c * 27 + 18 + k], a.w[c * 27 + 9 + k]};
    w0[k / 3][k % 3] = a.w[c * 27 + k];
  }
  float sc = 1.f, sh = 0.f;
  if (a.ss) { sc = a.ss[c * 2]; sh = a.ss[c * 2 + 1]; }
  const int act = a.act;

  const int planeB = g.H * g.W * EB;
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((T*)a.x, 0, pa.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc((T*)a.y, 0, pa.in_bytes, 0x00020000);
  const int voff = active ? (int)((((long long)n * g.C + c) * g.T * g.H * g.W + r * g.W + SW * sidx) * EB) : DW_OOB;
  int voffH[NH], voff1[ODD ? NH : 1];
#pragma unroll
  for (int h = 0; h < NH; h++) {
    voffH[h] = (active && 2 * h + 1 < ncol + (ODD ? 0 : 1)) ? voff + 2 * h * EB : DW_OOB;
    if constexpr (ODD) voff1[h] = (active && 2 * h + 1 == ncol) ? voff + 2 * h * EB : DW_OOB;
  }
  const int shift = ODD ? SW - ncol : 0;
  const int voffL = (ODD && active) ? voff - shift * EB : voff;
  auto aligned = [&](const Raw& rw) -> Raw {
    if constexpr (ODD) {
      static_assert(!ODD || (EB == 2 && SW == 4), "odd rows: 16-bit storage, strips of 4");
      const unsigned long long v = (((unsigned long long)rw.w[1] << 32) | rw.w[0]) >> (16 * shift);
      Raw o; o.w[0] = (unsigned)v; o.w[1] = (unsigned)(v >> 32); o.w[2] = o.w[3] = 0u;
      return o;
    } else {
      return rw;
    }
  };
  float* myA = lds + (2 * (active ? p : 0)) * tile;
  const int lS = (r + 1) * LP + 1 + SW * sidx;
  const int lW = r * LP + SW * sidx;

  Raw slot[PD];
  auto issue = [&](int t, Raw& s) {
    const int soff = (t < g.T && !pa.noload) ? t * planeB : DW_OOB;
    raw_bload<SW * EB>(s, rsX, voffL, soff);
  };
  auto stage = [&](const Raw& s0, int q) {
    const Raw s = aligned(s0);
    float* A = myA + q * tile;
#pragma unroll
    for (int e = 0; e < SW; e++) {
      const float u = sc * raw_get<T>(s, e) + sh;
      if (okc[e]) A[lS + e] = act == X3D_ACT_RELU ? fmaxf(u, 0.f) : u;
    }
  };

  float s1 = 0.f, s2 = 0.f;
  auto store_plane = [&](int t, bool live, const float (&v)[SW]) {   // !live: nothing stored, nothing summed
    float u[SW];
#pragma unroll
    for (int i = 0; i < SW; i++) {
      u[i] = (live && okc[i]) ? v[i] : 0.f;
      s1 += u[i];
      s2 += u[i] * u[i];
    }
    const int soff = (live && !pa.noload) ? t * planeB : DW_OOB;
#pragma unroll
    for (int h = 0; h < NH; h++) {
      const float two[2] = {u[2 * h], u[2 * h + 1]};
      Raw o;
      raw_pack<T, 2>(o, two);
      raw_bstore<2 * EB>(o, rsY, voffH[h], soff);
      if constexpr (ODD) raw_bstore<EB>(o, rsY, voff1[h], soff);
    }
  };
  auto dummy_stores = [&]() {
    Raw z; z.w[0] = z.w[1] = z.w[2] = z.w[3] = 0u;
#pragma unroll
    for (int h = 0; h < NH; h++) {
      raw_bstore<2 * EB>(z, rsY, voffH[h], DW_OOB);
      if constexpr (ODD) raw_bstore<EB>(z, rsY, voff1[h], DW_OOB);
    }
  };

  v2f acc01[SW], acc2p[(SW + 1) / 2];   // (out[t-1], out[t]) per output, out[t+1] as pairs over outputs
  float fin[SW];
#pragma unroll
  for (int i = 0; i < SW; i++) { acc01[i] = (v2f){0.f, 0.f}; fin[i] = 0.f; }
#pragma unroll
  for (int j = 0; j < (SW + 1) / 2; j++) acc2p[j] = (v2f){0.f, 0.f};

#pragma unroll
  for (int d = 0; d < PD; d++) {
    issue(d, slot[d]);
    dummy_stores();
  }
  __syncthreads();
  stage(slot[0], 0);
  issue(PD, slot[0]);
  dummy_stores();
  __syncthreads();

  for (int t0 = 0; t0 < g.T; t0 += UN) {
#pragma unroll
    for (int d = 0; d < UN; d++) {
      const int t = t0 + d;
      if (t >= g.T) break;
      const int cur = d & 1, prv = cur ^ 1;
      const int sl = (d + 1) % PD;
      float win[3][WIN];
      {
        const float* A = myA + cur * tile + lW;
#pragma unroll
        for (int kh = 0; kh < 3; kh++) lds_window<WIN, SW>(A + kh * LP, win[kh]);
      }
      store_plane(t - 2, t >= 2, fin);                 // plane t-2 (finished at the end of iteration t-1)
      stage(slot[sl], prv);
      issue(t + 1 + PD, slot[sl]);
#pragma unroll
      for (int kh = 0; kh < 3; kh++) dw_taps_row<1, SW, WIN>(win[kh], w21[kh], w0[kh], acc01, acc2p);
      dw_rotate<SW>(fin, acc01, acc2p);
      __syncthreads();
    }
  }
  store_plane(g.T - 2, g.T >= 2, fin);
  float last[SW];
#pragma unroll
  for (int i = 0; i < SW; i++) last[i] = acc01[i].x;
  store_plane(g.T - 1, true, last);

  if (a.stats || a.pool) {
    // per-plane sums: the threads of plane p add into scratch[p] (LDS atomics, once per workgroup)
    if (active) {
      atomicAdd(&scratch[p * 2], s1);
      atomicAdd(&scratch[p * 2 + 1], s2);
    }
    __syncthreads();
    if (threadIdx.x < pa.NP) {
      const int nn = grp * pa.NP + threadIdx.x;
      if (nn < g.N) {
        const float q1 = scratch[threadIdx.x * 2], q2 = scratch[threadIdx.x * 2 + 1];
        if (a.stats) {
          double* sp = stats_replica(a.stats, g.C, (unsigned)nn);
          atomic_add_d(&sp[c * 2], (double)q1);
          atomic_add_d(&sp[c * 2 + 1], (double)q2);
        }
        if (a.pool) atomic_add_d(&a.pool[(long long)nn * g.C + c], (double)q1);
      }
    }
  }
}

template <typename T, int SW>
static bool fwd_pk_t(const DwFwdArgs& a, hipStream_t st) {
  const DwGeom& g = a.g;
  DwPkFwdArgs pa;
  pa.f = a;
  const int nstr = ceil_div(g.W, SW);
  pa.items = g.H * nstr;
  pa.NP = 64 / pa.items > 0 ? 64 / pa.items : 1;
  if (pa.NP > g.N) pa.NP = g.N;
  pa.ngroups = ceil_div(g.N, pa.NP);
  pa.LP = (nstr * SW + 2 + 3) & ~3;
  if (SW == 4 && nstr * SW == 16 && g.W + 2 <= 16) pa.LP = 16;
  pa.in_bytes = (unsigned)((long long)g.N * g.C * g.T * g.H * g.W * (long long)sizeof(T));
  const int threads = ceil_div(pa.NP * pa.items, 64) * 64;
  const size_t lds = ((size_t)pa.NP * 2 * (g.H + 2) * pa.LP + 8 + pa.NP * 2 + 8) * sizeof(float);
  if (lds > 64 * 1024 || threads > PK_MAX_THREADS) return false;
  const bool fixed = (SW == 4 && pa.LP == 16 && g.H == 14) || (SW == 2 && pa.LP == 12 && g.H == 10);
  const bool odd7 = SW == 4 && g.W == 7 && g.H == 7 && pa.LP == 12;
  if ((g.W & 1) && !odd7) return false;
  if (x3d_describe.out) {
    snprintf(x3d_describe.out, x3d_describe.cap, "dw3d_fwd_pk_kernel<%s, %d, 2, 2, %d, %d, %d>", TypeName<T>::v, SW,
             (fixed || odd7) ? pa.LP : 0, (fixed || odd7) ? g.H : 0, (int)odd7);
    return true;
  }
  pa.noload = pk_env("X3D_DW_PK_NOLOAD", 0) == 1;   // result-changing timing hook: -DX3D_EXPERIMENTS builds only
  auto kern = fixed ? dw3d_fwd_pk_kernel<T, SW, 2, 2, (SW == 4 ? 16 : 12), (SW == 4 ? 14 : 10), false>
                    : dw3d_fwd_pk_kernel<T, SW, 2, 2, 0, 0, false>;
  if constexpr (SW == 4 && sizeof(T) == 2) {
    if (odd7) kern = dw3d_fwd_pk_kernel<T, 4, 2, 2, 12, 7, true>;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(g.C * pa.ngroups)), dim3(threads), lds, st, pa);
  return true;
}

// forward counterpart of dw_bwd_pk_launch: stride 1, whole plane per tile, rows of 10..18 outputs (strips of 4 at 14x14,
// else 2) and 7x7 planes (16-bit storage), no folded BatchNorm finalize
bool dw_fwd_pk_launch(const DwFwdArgs& a, int dtype, int S, int SW_caller, hipStream_t st) {
  const DwGeom& g = a.g;
  if (pk_env("X3D_DW_PK", 1) == 0 || pk_env("X3D_DW_PKF", 1) == 0 || S != 1 || g.ntile_h != 1 || g.pw != 1 || g.ph != 1 || a.bn.stats)
    return false;
  const bool odd7 = dtype != X3D_F32 && g.W == 7 && g.H == 7;
  if ((g.W % 2) != 0 && !odd7) return false;
  const bool fixed4 = dtype != X3D_F32 && g.W == 14 && g.H == 14;
  if (SW_caller != 2 && !odd7) return false;
  // measured (64 clips, bf16): 14x14 65.7 -> 63.3 us, 7x7 40.8 -> 38.8 us; strips of 2 (10x10) 54 -> 69 us: the forward
  // packed kernel is used where strips of 4 apply and nowhere else (X3D_DW_PK_SW=2 forces strips of 2 for A/B runs)
  const int SW = odd7 ? 4 : pk_env("X3D_DW_PK_SW", fixed4 ? 4 : 0);
  if (SW != 2 && SW != 4) return false;
  const int items = g.H * ceil_div(g.W, SW);
  if (items > PK_MAX_THREADS) return false;
  const int eb = dtype == X3D_F32 ? 4 : 2;
  const long long bytes = (long long)g.N * g.C * g.T * g.H * g.W * eb;
  if (bytes >= (1ll << 30)) return false;
  const uintptr_t al = (uintptr_t)(2 * eb) - 1;
  if (((uintptr_t)a.x & al) || ((uintptr_t)a.y & al)) return false;
  if ((long long)g.C * g.N >= (1ll << 31)) return false;
#define PK_GO(TT) (SW == 4 ? fwd_pk_t<TT, 4>(a, st) : fwd_pk_t<TT, 2>(a, st))
  return dtype == X3D_BF16 ? PK_GO(bf16) : dtype == X3D_F16 ? PK_GO(f16) : PK_GO(float);
#undef PK_GO
}
