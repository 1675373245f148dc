// Squeeze-excite MLP (reference model.py:274-290,311-315) forward, and its backward fused with the
// BatchNorm_b backward finalize.  These are per-sample GEMVs on [C] vectors (C <= 630): a handful of
// microseconds; what matters is that they need NO extra pass over the activations -- the squeeze
// comes out of the depthwise epilogue and the backward sums out of the pointwise dgrad epilogue.
#include "common.h"

#define SE_MAXC 1024
#define SE_MAXW 64

__global__ __launch_bounds__(256) void se_fwd_kernel(const double* __restrict__ pool_sums, double P,
                                                     const float* __restrict__ ssb, const float* __restrict__ w1,
                                                     const float* __restrict__ b1, const float* __restrict__ w2,
                                                     const float* __restrict__ b2, float* gate, float* hidden,
                                                     int C, int Wd) {
  __shared__ float pooled[SE_MAXC];
  __shared__ float hid[SE_MAXW];
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  for (int c = tid; c < C; c += 256)
    pooled[c] = ssb[c * 2] * (float)(pool_sums[(long long)n * C + c] / P) + ssb[c * 2 + 1];
  __syncthreads();
  // (loops with a run-time trip count are not unrolled: every load waited for the previous one -- 32 L2 latencies per thread in
  //  the fc2 loop were most of the 28 us this kernel took on the 432-channel layers; batches of 8 loads in flight)
  // fc1: a wave takes the rows j = wid, wid + 4, ... EIGHT at a time (one row after the other every row's loads waited for the
  // previous row's reduction: 8 x ~1.5 us on the 432-channel layers)
  for (int j0 = wid; j0 < Wd; j0 += 32) {
    float acc[8];
#pragma unroll
    for (int u = 0; u < 8; u++) acc[u] = 0.f;
    for (int c = lane; c < C; c += 64) {
      const float pv = pooled[c];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int j = j0 + 4 * u;
        acc[u] += (j < Wd ? w1[j * C + c] : 0.f) * pv;
      }
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int j = j0 + 4 * u;
      const float t = wave_sum(acc[u]);
      if (lane == 0 && j < Wd) {
        const float h = fmaxf(t + b1[j], 0.f);
        hid[j] = h;
        hidden[(long long)n * Wd + j] = h;
      }
    }
  }
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    float acc = b2[c];
    const float* wr = w2 + c * Wd;
    int j = 0;
    for (; j + 8 <= Wd; j += 8) {
      float wv[8];
#pragma unroll
      for (int u = 0; u < 8; u++) wv[u] = wr[j + u];
#pragma unroll
      for (int u = 0; u < 8; u++) acc += wv[u] * hid[j + u];
    }
    for (; j < Wd; j++) acc += wr[j] * hid[j];
    gate[(long long)n * C + c] = sigmoidf_(acc);
  }
}

extern "C" int x3d_se_fwd(const double* pool_sums, double P, const float* b_scale_shift, const float* w1,
                          const float* b1, const float* w2, const float* b2, float* gate, float* hidden, int N,
                          int C, int Wd, void* stream) {
  X3D_REQUIRE(pool_sums && b_scale_shift && w1 && b1 && w2 && b2 && gate && hidden, "se_fwd: null pointer");
  X3D_REQUIRE(N > 0 && C > 0 && C <= SE_MAXC && Wd > 0 && Wd <= SE_MAXW && P > 0, "se_fwd: bad extents");
  hipLaunchKernelGGL(se_fwd_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, pool_sums, P, b_scale_shift, w1,
                     b1, w2, b2, gate, hidden, C, Wd);
  X3D_LAUNCH_CHECK("se_fwd");
  return X3D_OK;
}

// Weight-gradient slabs (x3d_hip.h dw_slab): dw[e] += sum_p slab[p][e], p ascending -- a fixed order, so the result does not
// depend on timing.  A 256-thread workgroup owns 64 consecutive elements: thread (q = tid & 15, pg = tid >> 4) loads the float4
// q of the parts pg, pg + 16, ... (16 loads in flight for 256 parts, 256 contiguous bytes per part and 16 lanes), the sixteen
// part groups meet in LDS.  Read once from L2 / the Infinity Cache right after the producer wrote them: ~20 MB in a few us,
// hidden behind the small launch that carries it.
#define DWR_EPB 64        // elements per workgroup
__device__ __forceinline__ void dw_slab_reduce_block(const x3d_dw_reduce_job& j, int blk, float (*part)[16][4]) {
  const int tid = threadIdx.x, q = tid & 15, pg = tid >> 4;
  const long long e0 = (long long)blk * DWR_EPB + q * 4;
  f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  const bool full = e0 + 4 <= j.elems;      // elems % 4 == 0 is required: a float4 is inside or outside
  if (full) {
    const float* src = j.slab + e0;
    int p = pg;
    for (; p + 48 < j.parts; p += 64) {      // four independent chains
#pragma unroll
      for (int u = 0; u < 4; u++) acc[u] += *(const f32x4*)(src + (long long)(p + 16 * u) * j.elems);
    }
    for (; p < j.parts; p += 16) acc[0] += *(const f32x4*)(src + (long long)p * j.elems);
  }
  const f32x4 t = (acc[0] + acc[1]) + (acc[2] + acc[3]);
#pragma unroll
  for (int c = 0; c < 4; c++) part[pg][q][c] = t[c];
  __syncthreads();
  if (tid < 64) {                            // element tid of the block: the sixteen part groups in a fixed order
    const int qq = tid >> 2, c = tid & 3;
    float v = 0.f;
#pragma unroll
    for (int g = 0; g < 16; g++) v += part[g][qq][c];
    const long long e = (long long)blk * DWR_EPB + tid;
    if (e < j.elems) j.dw[e] += v;
  }
}
static inline int dw_reduce_blocks(const x3d_dw_reduce_job& j) { return j.slab ? ceil_div(j.elems, DWR_EPB) : 0; }
static bool dw_reduce_job_ok(const x3d_dw_reduce_job& j) {
  return !j.slab || (j.dw && j.parts > 0 && j.elems > 0 && (j.elems % 4) == 0 && ((uintptr_t)j.slab % 16) == 0);
}

__global__ __launch_bounds__(256) void dw_slab_reduce_kernel(x3d_dw_reduce_job j0, x3d_dw_reduce_job j1, int nb0) {
  __shared__ float part[16][16][4];
  const int b = blockIdx.x;
  if (b < nb0) dw_slab_reduce_block(j0, b, part);
  else dw_slab_reduce_block(j1, b - nb0, part);
}

extern "C" int x3d_dw_slab_reduce(const x3d_dw_reduce_job* jobs, int n_jobs, void* stream) {
  X3D_REQUIRE(jobs && n_jobs >= 1 && n_jobs <= 2, "dw_slab_reduce: one or two jobs");
  x3d_dw_reduce_job j0 = jobs[0], j1;
  memset(&j1, 0, sizeof(j1));
  if (n_jobs == 2) j1 = jobs[1];
  X3D_REQUIRE(dw_reduce_job_ok(j0) && dw_reduce_job_ok(j1), "dw_slab_reduce: bad job (elems % 4 == 0, 16-byte aligned slab, dw)");
  const int nb0 = dw_reduce_blocks(j0), nb1 = dw_reduce_blocks(j1);
  if (nb0 + nb1 == 0) return X3D_OK;
  hipLaunchKernelGGL(dw_slab_reduce_kernel, dim3(nb0 + nb1), dim3(256), 0, (hipStream_t)stream, j0, j1, nb0);
  X3D_LAUNCH_CHECK("dw_slab_reduce");
  return X3D_OK;
}

// stage 1 (one block per sample): gradient through gate -> fc2 -> ReLU -> fc1 -> pooled.  No atomics: the
// per-sample vectors dz2 [N][C] (pre-sigmoid grad of fc2) and dz1 [N][Wd] (pre-ReLU grad of fc1) go to scratch
// and stage 2 contracts them over the samples.  scratch = dpool [N][C] | dz2 [N][C] | dz1 [N][Wd].
__global__ __launch_bounds__(256) void se_bwd_kernel(const x3d_se_bnb_bwd_args a) {
  __shared__ float dz2[SE_MAXC];
  __shared__ float dz1[SE_MAXW];
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int C = a.C, Wd = a.Wd;
  float* dz2_out = a.scratch + (long long)a.N * C;
  float* dz1_out = a.scratch + 2ll * a.N * C;
  for (int c = tid; c < C; c += 256) {
    const float sb = a.b_scale_shift[c * 2], tb = a.b_scale_shift[c * 2 + 1];
    const long long i = (long long)n * C + c;
    // dgate = sum_p dv*u, u = sb*braw + tb
    const float dgate = (float)((double)sb * a.nc_sums[i * 2 + 1] + (double)tb * a.nc_sums[i * 2]);
    const float g = a.gate[i];
    const float d = dgate * g * (1.f - g);
    dz2[c] = d;
    dz2_out[i] = d;
  }
  __syncthreads();
  for (int j0 = wid; j0 < Wd; j0 += 32) {      // eight rows of fc2^T at a time (see se_fwd_kernel)
    float acc[8];
#pragma unroll
    for (int u = 0; u < 8; u++) acc[u] = 0.f;
    for (int c = lane; c < C; c += 64) {
      const float dv = dz2[c];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int j = j0 + 4 * u;
        acc[u] += (j < Wd ? a.w2[c * Wd + j] : 0.f) * dv;
      }
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int j = j0 + 4 * u;
      const float t = wave_sum(acc[u]);
      if (lane == 0 && j < Wd) {
        const float d = a.hidden[(long long)n * Wd + j] > 0.f ? t : 0.f;
        dz1[j] = d;
        dz1_out[(long long)n * Wd + j] = d;
      }
    }
  }
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    float dp = 0.f;
    int j = 0;
    for (; j + 8 <= Wd; j += 8) {     // eight loads in flight (see se_fwd_kernel)
      float wv[8];
#pragma unroll
      for (int u = 0; u < 8; u++) wv[u] = a.w1[(j + u) * C + c];
#pragma unroll
      for (int u = 0; u < 8; u++) dp += wv[u] * dz1[j + u];
    }
    for (; j < Wd; j++) dp += a.w1[j * C + c] * dz1[j];
    a.scratch[(long long)n * C + c] = dp;  // d loss / d pooled[n][c]
  }
}

// stage 2 (one 64-lane workgroup per channel, lanes split the samples): BN_b backward over
// du = dv*gate + dpool/P with fp64 sums, the per-(n,c) coefficients, and the SE weight gradients of this channel
//   dw2[c][j] += sum_n dz2[n][c]*hidden[n][j]    db2[c] += sum_n dz2[n][c]
//   dw1[j][c] += sum_n dz1[n][j]*pooled[n][c]    db1[j]  += sum_n dz1[n][j]   (workgroup 0)
// Each gradient element has exactly one writer, so the += are plain read-modify-writes.
// With SE the workgroup has FOUR waves: wave 0 does the BatchNorm part, and all four split the sequential pass over the samples
// of the SE weight gradients (64 dependent rounds of loads in one wave were ~12 of the kernel's ~16 us on the 432-channel layers)
__global__ __launch_bounds__(256) void bnb_bwd_kernel(const x3d_se_bnb_bwd_args a, int has_se, int nb0) {
  __shared__ float part[4][3][64];
  if ((int)blockIdx.x >= a.C) {        // extra workgroups: the weight-gradient slabs of earlier launches (a.reduce)
    __shared__ float rpart[16][16][4];
    const int b = (int)blockIdx.x - a.C;
    if (b < nb0) dw_slab_reduce_block(a.reduce[0], b, rpart);
    else dw_slab_reduce_block(a.reduce[1], b - nb0, rpart);
    return;
  }
  if (blockDim.x == 256 && !has_se && threadIdx.x >= 64) return;   // (launched wide only for the reduce workgroups)
  const int c = blockIdx.x, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int C = a.C, N = a.N;
  if (wid == 0) {
    double sdu = 0.0, sdub = 0.0;
    for (int n = lane; n < N; n += 64) {
      const long long i = (long long)n * C + c;
      const double g = has_se ? (double)a.gate[i] : 1.0;
      const double dp = has_se ? (double)a.scratch[i] : 0.0;
      sdu += g * a.nc_sums[i * 2] + dp;
      sdub += g * a.nc_sums[i * 2 + 1] + (has_se ? dp / a.P * a.pool_sums[i] : 0.0);
    }
    sdu = wave_sum_d(sdu);
    sdub = wave_sum_d(sdub);
    const double count = (double)N * a.P;
    const double mean = a.b_mean_invstd[c * 2], invstd = a.b_mean_invstd[c * 2 + 1];
    const double dga = (sdub - mean * sdu) * invstd;
    const double k1 = (double)a.gamma_b[c] * invstd;
    const double B = -k1 * invstd * dga / count;
    const double Cc = -k1 * sdu / count - B * mean;
    if (lane == 0) {
      a.dgamma_b[c] += (float)dga;
      a.dbeta_b[c] += (float)sdu;
    }
    for (int n = lane; n < N; n += 64) {
      const long long i = (long long)n * C + c;
      const double g = has_se ? (double)a.gate[i] : 1.0;
      const double dp = has_se ? (double)a.scratch[i] : 0.0;
      float* o = a.coef_nc + i * 4;
      o[0] = (float)(k1 * g);
      o[1] = (float)B;
      o[2] = (float)(Cc + k1 * dp / a.P);
      o[3] = 0.f;
    }
  }
  if (!has_se) return;
  const int Wd = a.Wd;
  const float* dz2 = a.scratch + (long long)N * C;
  const float* dz1 = a.scratch + 2ll * N * C;
  const float sb = a.b_scale_shift[c * 2], tb = a.b_scale_shift[c * 2 + 1];
  if (wid == 0) {
    float sb2 = 0.f;
    for (int n = lane; n < N; n += 64) sb2 += dz2[(long long)n * C + c];
    sb2 = wave_sum(sb2);
    if (lane == 0) a.db2[c] += sb2;
  }
  // lanes = hidden units; the four waves take the samples n = wid, wid + 4, ... -- no cross-lane traffic, partial sums meet in LDS
  float s2 = 0.f, s1 = 0.f, sj = 0.f;
  if (lane < Wd) {
#pragma unroll 4
    for (int n = wid; n < N; n += 4) {
      const long long i = (long long)n * C + c;
      const float d2 = dz2[i];
      const float pooled = sb * (float)(a.pool_sums[i] / a.P) + tb;
      const float d1 = dz1[(long long)n * Wd + lane], h = a.hidden[(long long)n * Wd + lane];
      s2 += d2 * h;
      s1 += d1 * pooled;
      sj += d1;
    }
  }
  part[wid][0][lane] = s2; part[wid][1][lane] = s1; part[wid][2][lane] = sj;
  __syncthreads();
  if (wid == 0 && lane < Wd) {
    // (the four partial sums are added in a fixed order: the result does not depend on the waves' timing)
    const float t2 = (part[0][0][lane] + part[1][0][lane]) + (part[2][0][lane] + part[3][0][lane]);
    const float t1 = (part[0][1][lane] + part[1][1][lane]) + (part[2][1][lane] + part[3][1][lane]);
    const float tj = (part[0][2][lane] + part[1][2][lane]) + (part[2][2][lane] + part[3][2][lane]);
    a.dw2[c * Wd + lane] += t2;
    a.dw1[lane * C + c] += t1;
    if (c == 0) a.db1[lane] += tj;
  }
}

extern "C" int x3d_se_bnb_bwd(const x3d_se_bnb_bwd_args* a, void* stream) {
  X3D_REQUIRE(a && a->nc_sums && a->b_scale_shift && a->b_mean_invstd && a->gamma_b && a->dgamma_b &&
                  a->dbeta_b && a->coef_nc, "se_bnb_bwd: null pointer");
  X3D_REQUIRE(a->N > 0 && a->C > 0 && a->C <= SE_MAXC && a->P > 0, "se_bnb_bwd: bad extents");
  const int has_se = a->w1 != nullptr;
  hipStream_t st = (hipStream_t)stream;
  if (has_se) {
    X3D_REQUIRE(a->w2 && a->b1 && a->b2 && a->gate && a->hidden && a->pool_sums && a->dw1 && a->db1 && a->dw2 &&
                    a->db2 && a->scratch, "se_bnb_bwd: SE pointers missing");
    X3D_REQUIRE(a->Wd > 0 && a->Wd <= SE_MAXW, "se_bnb_bwd: bad SE width");
    hipLaunchKernelGGL(se_bwd_kernel, dim3(a->N), dim3(256), 0, st, *a);
    X3D_LAUNCH_CHECK("se_bwd");
  }
  X3D_REQUIRE(dw_reduce_job_ok(a->reduce[0]) && dw_reduce_job_ok(a->reduce[1]), "se_bnb_bwd: bad reduce job (elems % 4 == 0, 16-byte aligned slab, dw)");
  const int nb0 = dw_reduce_blocks(a->reduce[0]), nb1 = dw_reduce_blocks(a->reduce[1]);
  hipLaunchKernelGGL(bnb_bwd_kernel, dim3(a->C + nb0 + nb1), dim3((has_se || nb0 + nb1) ? 256 : 64), 0, st, *a, has_se, nb0);
  X3D_LAUNCH_CHECK("bnb_bwd");
  return X3D_OK;
}
