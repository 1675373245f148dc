// x3d_eval_views: eval-side view construction on the GPU.
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
  const int frame = (int)(((long long)(v * a.T + t) * a.rate) % a.F);
  const int ry = y + a.yoff[sidx], rx = x + a.xoff[sidx];
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
  X3D_REQUIRE(e->dtype == X3D_F32 || e->dtype == X3D_BF16, "eval_views: bad dtype");
  EvalViewsArgs a;
  a.video = e->video; a.out = e->out; a.F = e->F; a.H = e->H; a.W = e->W;
  a.T = e->T; a.views = e->views; a.crops = e->crops; a.size = e->size;
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
  else hipLaunchKernelGGL((eval_views_kernel<bf16>), dim3((unsigned)blocks), dim3(256), 0, st, a);
  X3D_LAUNCH_CHECK("eval_views");
  return X3D_OK;
}
