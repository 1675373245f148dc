// x3d_eval_views / x3d_train_clip: eval-side view construction and train-side clip construction
// (rank 4, device half) on the GPU.
//   decoded video [F][H][W][3] uint8  ->  [crops*views][T][size][size][3] normalised clips (channels-last, as the
//   model boundary takes them), one thread per output pixel.
// Reference semantics (behaviour restated in oracle/views_oracle.py, which the tests compare against bit for bit):
//   temporal looping sampler transforms.py:48-65, short-side bilinear resize + cast back to uint8 :112-147,
//   uniform crop with ceil offsets :149-190, normalise utils.py:42-72, [crops][views] order dataloader.py:107-116.
// The interpolation arithmetic must not be contracted into FMAs (fp contract off below): the uint8 truncation
// after the resize makes a 1-ulp difference visible.
#include "common.h"

struct EvalViewsArgs {
  const unsigned char* video;
  void* out;
  int F, H, W, nh, nw;       // source extents, resized extents
  int T, views, crops, size, rate;
  int start, flip;           // first frame of the sweep; mirror left-right (training clips)
  int yoff[3], xoff[3];      // crop offsets in the resized frame per spatial index
  float sy, sx;              // H / nh, W / nw (float32 division)
  float mean[3], std[3];
};

template <typename T>
__global__ __launch_bounds__(256) void eval_views_kernel(const EvalViewsArgs a) {
#pragma clang fp contract(off)   // plain * and + below must stay separate roundings (HIP's __fmul_rn / __fadd_rn are inline
                                 // functions whose bodies are contracted after inlining, so operators are used instead)
  const long long total = (long long)a.crops * a.views * a.T * a.size * a.size;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int x = (int)(i % a.size);
  long long r = i / a.size;
  const int y = (int)(r % a.size); r /= a.size;
  const int t = (int)(r % a.T); r /= a.T;
  const int v = (int)(r % a.views);
  const int ci = (int)(r / a.views);
  const int sidx = a.crops > 1 ? ci % 3 : 1;
  const int frame = (int)((a.start + (long long)(v * a.T + t) * a.rate) % a.F);
  const int ry = y + a.yoff[sidx], rx = (a.flip ? a.size - 1 - x : x) + a.xoff[sidx];
  // half-pixel-centre bilinear weights, float32, exactly as the oracle computes them
  const float fy = ((float)ry + 0.5f) * a.sy - 0.5f;
  const float fx = ((float)rx + 0.5f) * a.sx - 0.5f;
  const float fyf = floorf(fy), fxf = floorf(fx);
  const int y0 = max((int)fyf, 0), y1 = min((int)ceilf(fy), a.H - 1);
  const int x0 = max((int)fxf, 0), x1 = min((int)ceilf(fx), a.W - 1);
  const float ly = fy - fyf, lx = fx - fxf;
  const unsigned char* fr = a.video + (long long)frame * a.H * a.W * 3;
  T* o = (T*)a.out + i * 3;
  const bool ident = (a.nh == a.H) && (a.nw == a.W);
#pragma unroll
  for (int c = 0; c < 3; c++) {
    float px;
    if (ident) {
      px = (float)fr[((long long)ry * a.W + rx) * 3 + c];
    } else {
      const float tl = (float)fr[((long long)y0 * a.W + x0) * 3 + c], tr = (float)fr[((long long)y0 * a.W + x1) * 3 + c];
      const float bl = (float)fr[((long long)y1 * a.W + x0) * 3 + c], br = (float)fr[((long long)y1 * a.W + x1) * 3 + c];
      const float top = tl + (tr - tl) * lx;
      const float bot = bl + (br - bl) * lx;
      const float val = top + (bot - top) * ly;
      px = (float)(unsigned char)(int)val;    // cast back to uint8: truncation (values are inside [0, 255])
    }
    const float nv = (px / 255.0f - a.mean[c]) / a.std[c];
    o[c] = from_f<T>(nv);
  }
}

extern "C" int x3d_eval_views(const x3d_eval_views_args* e, void* stream) {
  X3D_REQUIRE(e && e->video && e->out, "eval_views: null pointer");
  X3D_REQUIRE(e->F > 0 && e->H > 0 && e->W > 0 && e->T > 0 && e->views > 0 && e->crops > 0 && e->size > 0,
              "eval_views: bad extents");
  X3D_REQUIRE(x3d_dtype_ok(e->dtype), "eval_views: bad dtype");
  EvalViewsArgs a;
  a.video = e->video; a.out = e->out; a.F = e->F; a.H = e->H; a.W = e->W;
  a.T = e->T; a.views = e->views; a.crops = e->crops; a.size = e->size;
  a.start = 0; a.flip = 0;
  a.rate = e->F / e->T > 1 ? e->F / e->T : 1;                      // transforms.py:51
  // short side -> size, the long side floor((long/short) * size) in float32 (transforms.py:129-141)
  const float h = (float)e->H, w = (float)e->W, s = (float)e->size;
  a.nh = e->H; a.nw = e->W;
  if (!((w <= h && w == s) || (h <= w && h == s))) {
    float nh = s, nw = s;
    if (w < h) nh = floorf((h / w) * s);
    else nw = floorf((w / h) * s);
    a.nh = (int)nh; a.nw = (int)nw;
  }
  X3D_REQUIRE(a.nh >= e->size && a.nw >= e->size, "eval_views: resized frame %dx%d smaller than the crop %d", a.nh, a.nw, e->size);
  a.sy = (float)e->H / (float)a.nh; a.sx = (float)e->W / (float)a.nw;
  for (int sidx = 0; sidx < 3; sidx++) {                            // transforms.py:170-186
    int y = (a.nh - e->size + 1) / 2, x = (a.nw - e->size + 1) / 2; // ceil((n - size) / 2), n >= size
    if (a.nh > a.nw) { if (sidx == 0) y = 0; else if (sidx == 2) y = a.nh - e->size; }
    else { if (sidx == 0) x = 0; else if (sidx == 2) x = a.nw - e->size; }
    a.yoff[sidx] = y; a.xoff[sidx] = x;
  }
  for (int c = 0; c < 3; c++) { a.mean[c] = e->mean[c]; a.std[c] = e->std[c]; }
  const long long total = (long long)e->crops * e->views * e->T * e->size * e->size;
  const long long blocks = ceil_div_ll(total, 256);
  X3D_REQUIRE(blocks < (1ll << 31), "eval_views: too many pixels");
  hipStream_t st = (hipStream_t)stream;
  if (e->dtype == X3D_F32) hipLaunchKernelGGL((eval_views_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, st, a);
  else if (e->dtype == X3D_F16) hipLaunchKernelGGL((eval_views_kernel<f16>), dim3((unsigned)blocks), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((eval_views_kernel<bf16>), dim3((unsigned)blocks), dim3(256), 0, st, a);
  X3D_LAUNCH_CHECK("eval_views");
  return X3D_OK;
}

// random_short_side_resize (transforms.py:126-141), float32 as in the reference: the short side becomes int(jitter)
// (the float -> int32 cast truncates), the long side floor((long / short) * jitter)
extern "C" int x3d_train_resized_hw(int H, int W, float jitter, int* new_h, int* new_w) {
  X3D_REQUIRE(H > 0 && W > 0 && jitter >= 1.0f && new_h && new_w, "train_resized_hw: bad arguments");
  const float h = (float)H, w = (float)W, s = jitter;
  *new_h = H; *new_w = W;
  if ((w <= h && w == s) || (h <= w && h == s)) return X3D_OK;
  float nh = s, nw = s;
  if (w < h) nh = floorf((h / w) * s);
  else nw = floorf((w / h) * s);
  *new_h = (int)nh; *new_w = (int)nw;
  return X3D_OK;
}

extern "C" int x3d_train_clip(const x3d_train_clip_args* e, void* stream) {
  X3D_REQUIRE(e && e->video && e->out, "train_clip: null pointer");
  X3D_REQUIRE(e->F > 0 && e->H > 0 && e->W > 0 && e->T > 0 && e->rate > 0 && e->size > 0, "train_clip: bad extents");
  X3D_REQUIRE(e->start >= 0 && e->start < e->F, "train_clip: start %d outside the %d frames", e->start, e->F);
  X3D_REQUIRE(x3d_dtype_ok(e->dtype), "train_clip: bad dtype");
  EvalViewsArgs a;
  a.video = e->video; a.out = e->out; a.F = e->F; a.H = e->H; a.W = e->W;
  a.T = e->T; a.views = 1; a.crops = 1; a.size = e->size;
  a.rate = e->rate; a.start = e->start; a.flip = e->flip ? 1 : 0;
  if (x3d_train_resized_hw(e->H, e->W, e->jitter, &a.nh, &a.nw) != X3D_OK) return X3D_ERR_INVALID;
  X3D_REQUIRE(a.nh >= e->size && a.nw >= e->size, "train_clip: resized frame %dx%d smaller than the crop %d", a.nh, a.nw, e->size);
  X3D_REQUIRE(e->y0 >= 0 && e->y0 <= a.nh - e->size && e->x0 >= 0 && e->x0 <= a.nw - e->size,
              "train_clip: crop offset (%d, %d) outside the %dx%d frame", e->y0, e->x0, a.nh, a.nw);
  a.sy = (float)e->H / (float)a.nh; a.sx = (float)e->W / (float)a.nw;
  for (int i = 0; i < 3; i++) { a.yoff[i] = e->y0; a.xoff[i] = e->x0; a.mean[i] = e->mean[i]; a.std[i] = e->std[i]; }
  const long long total = (long long)e->T * e->size * e->size;
  const long long blocks = ceil_div_ll(total, 256);
  X3D_REQUIRE(blocks < (1ll << 31), "train_clip: too many pixels");
  hipStream_t st = (hipStream_t)stream;
  if (e->dtype == X3D_F32) hipLaunchKernelGGL((eval_views_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, st, a);
  else if (e->dtype == X3D_F16) hipLaunchKernelGGL((eval_views_kernel<f16>), dim3((unsigned)blocks), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((eval_views_kernel<bf16>), dim3((unsigned)blocks), dim3(256), 0, st, a);
  X3D_LAUNCH_CHECK("train_clip");
  return X3D_OK;
}
