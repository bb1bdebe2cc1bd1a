// Input-pipeline tail on the GPU: crop window -> (flip) -> legacy-TF bilinear resize -> central-crop offset ->
// optional channel-mean subtraction, for a ragged batch of decoded uint8 images.
// Reference: preprocessing/imagenet_preprocessing.py:57-97 (crop + flip), :189-225 (_smallest_size_at_least,
// _aspect_preserving_resize, _resize_image = tf.image.resize_images(BILINEAR, align_corners=False)), :97-120
// (central_crop), :122-155 (mean_image_subtraction).  TF-1.14's resize has NO half-pixel centres:
// in = out_index * (in_size / out_size), lower = (int)in, upper = min(lower + 1, in_size - 1), lerp = in - lower,
// value = top + (bottom - top) * y_lerp with top = tl + (tr - tl) * x_lerp, all in float32, no fused multiply-add.
// HBM-bound byte work: one thread per output pixel, 12 source bytes gathered, 12 output bytes written.
#include "common.h"

namespace {

__global__ void __launch_bounds__(256)
resize_crop_flip_kernel(const uint8_t* __restrict__ src, long long src_bytes, const asm_image_desc* __restrict__ descs,
                        int out_h, int out_w, int subtract_mean, float* __restrict__ out) {
#pragma clang fp contract(off)
  const int n = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= out_h * out_w) return;
  const asm_image_desc d = descs[n];
  const int i = p / out_w, j = p - i * out_w;
  float* o = out + ((size_t)n * out_h * out_w + p) * 3;
  // memory safety only (the host mirror validates and raises): a window outside the image or the buffer gives zeros
  const long long need = d.src_offset + (long long)d.Hs * d.Ws * 3;
  const bool ok = d.src_offset >= 0 && need <= src_bytes && d.crop_h > 0 && d.crop_w > 0 && d.crop_y >= 0 &&
                  d.crop_x >= 0 && d.crop_y + d.crop_h <= d.Hs && d.crop_x + d.crop_w <= d.Ws && d.resize_h > 0 &&
                  d.resize_w > 0 && d.out_y >= 0 && d.out_x >= 0 && d.out_y + out_h <= d.resize_h &&
                  d.out_x + out_w <= d.resize_w;
  if (!ok) { o[0] = 0.f; o[1] = 0.f; o[2] = 0.f; return; }
  const float hs = (float)d.crop_h / (float)d.resize_h;
  const float ws = (float)d.crop_w / (float)d.resize_w;
  const float in_y = (float)(d.out_y + i) * hs;
  const float in_x = (float)(d.out_x + j) * ws;
  const int ly = (int)in_y, lx = (int)in_x;
  const int uy = min(ly + 1, d.crop_h - 1), ux = min(lx + 1, d.crop_w - 1);
  const float fy = in_y - (float)ly, fx = in_x - (float)lx;
  const int cl = d.flip ? d.crop_w - 1 - lx : lx;
  const int cu = d.flip ? d.crop_w - 1 - ux : ux;
  const uint8_t* base = src + d.src_offset;
  const uint8_t* r0 = base + ((size_t)(d.crop_y + ly) * d.Ws + d.crop_x) * 3;
  const uint8_t* r1 = base + ((size_t)(d.crop_y + uy) * d.Ws + d.crop_x) * 3;
  const float means[3] = {123.68f, 116.78f, 103.94f};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float tl = (float)r0[cl * 3 + c], tr = (float)r0[cu * 3 + c];
    const float bl = (float)r1[cl * 3 + c], br = (float)r1[cu * 3 + c];
    const float top = tl + (tr - tl) * fx;
    const float bot = bl + (br - bl) * fx;
    float v = top + (bot - top) * fy;
    if (subtract_mean) v = v - means[c];
    o[c] = v;
  }
}

}  // namespace

extern "C" int asm_resize_crop_flip(const uint8_t* src, int64_t src_bytes, const asm_image_desc* descs, int N,
                                    int out_h, int out_w, int subtract_mean, float* out, void* stream) {
  ASM_REQUIRE(N >= 0 && out_h > 0 && out_w > 0 && src_bytes >= 0, "resize_crop_flip: bad sizes");
  ASM_REQUIRE((long long)out_h * out_w < (1ll << 30), "resize_crop_flip: output too large");
  if (N == 0) return ASM_OK;
  ASM_REQUIRE(src && descs && out, "resize_crop_flip: null pointer");
  dim3 grid((out_h * out_w + 255) / 256, N);
  ASM_LAUNCH(resize_crop_flip_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, src_bytes, descs, out_h, out_w,
             subtract_mean ? 1 : 0, out);
  ASM_CHECK_LAUNCH("resize_crop_flip");
  return ASM_OK;
}
