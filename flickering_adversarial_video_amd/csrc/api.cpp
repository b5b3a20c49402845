// Error plumbing, version, and host-side packing of convolution weights into MFMA fragment order.
#include <stdarg.h>
#include <string.h>
#include <new>
#include "flk_internal.h"

static thread_local char g_err[1024] = "";

void flk_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int flk_version(void) { return 100; }
extern "C" const char* flk_last_error(void) { return g_err; }

static inline uint16_t f32_to_bf16_rne(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // keep NaN a NaN
  return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

int flk_conv_weights_create_impl(const float* w, int kt, int kh, int kw, int cin, int cout,
                                 const float* row_scale, int transpose, int dtype, int nf, int cin_split,
                                 flk_conv_weights** out) {
  FLK_REQUIRE(w && out, "flk_conv_weights_create: null argument");
  FLK_REQUIRE(dtype == FLK_F32 || dtype == FLK_BF16, "flk_conv_weights_create: bad dtype %d", dtype);
  FLK_REQUIRE(nf == 2 || nf == 4 || nf == 8, "flk_conv_weights_create: nf must be 2, 4 or 8 (got %d)", nf);
  FLK_REQUIRE(kt > 0 && kh > 0 && kw > 0 && cin > 0 && cout > 0, "flk_conv_weights_create: bad shape");
  const int ocin = transpose ? cout : cin;    // operator input channels (GEMM K per tap)
  const int ocout = transpose ? cin : cout;   // operator output channels
  const int epl = dtype == FLK_BF16 ? 8 : 4, slabc = 4 * epl;
  const int ntaps = kt * kh * kw;
  FLK_REQUIRE(cin_split >= 0 && cin_split < ocin && cin_split % 8 == 0 && !(cin_split && transpose),
              "flk_conv_weights_create: bad cin_split %d", cin_split);
  const int nslab1 = cin_split ? (cin_split + slabc - 1) / slabc : 0;
  const int nslab = cin_split ? nslab1 + (ocin - cin_split + slabc - 1) / slabc : (ocin + slabc - 1) / slabc;
  const int cout_frags = (ocout + 16 * nf - 1) / (16 * nf) * nf;
  const size_t nelem = (size_t)nslab * ntaps * cout_frags * 64 * epl;
  const size_t bytes = nelem * (dtype == FLK_BF16 ? 2 : 4);
  std::vector<char> host(bytes);
  uint16_t* hb = (uint16_t*)host.data();
  float* hf = (float*)host.data();
  size_t o = 0;
  for (int s = 0; s < nslab; ++s)
    for (int tap = 0; tap < ntaps; ++tap) {
      const int src_tap = transpose ? ntaps - 1 - tap : tap;   // flipping all three axes = reversing the tap index
      for (int F = 0; F < cout_frags; ++F) {
        const int ntile = F / nf, f = F % nf;
        for (int lane = 0; lane < 64; ++lane) {
          const int q = lane >> 4, m = lane & 15;
          // channel owned by accumulator row m = 4q + r of fragment f.  16-byte store group G = f / (epl/4) of lane group q
          // covers channels [G*4*epl + q*epl, +epl): in ONE store instruction the four lane groups q write four consecutive
          // 16-byte pieces (64 contiguous bytes per position).
          const int fpg = epl / 4, G = f / fpg;
          const int co = ntile * 16 * nf + G * 4 * epl + (m >> 2) * epl + (f % fpg) * 4 + (m & 3);
          for (int j = 0; j < epl; ++j, ++o) {
            int ci = s * slabc + q * epl + j;          // position in the (segment-padded) K order -> channel
            bool civalid = ci < ocin;
            if (cin_split) {
              if (s < nslab1) civalid = ci < cin_split;
              else { ci = cin_split + (s - nslab1) * slabc + q * epl + j; civalid = ci < ocin; }
            }
            float v = 0.f;
            if (co < ocout && civalid) {
              // original array is [tap][cin][cout]
              v = transpose ? w[((size_t)src_tap * cin + co) * cout + ci] : w[((size_t)src_tap * cin + ci) * cout + co];
              if (row_scale) v *= row_scale[ci];
            }
            if (dtype == FLK_BF16) hb[o] = f32_to_bf16_rne(v); else hf[o] = v;
          }
        }
      }
    }
  flk_conv_weights* cw = new (std::nothrow) flk_conv_weights();
  if (!cw) { flk_set_error("flk_conv_weights_create: out of host memory"); return FLK_ENOMEM; }
  cw->kt = kt; cw->kh = kh; cw->kw = kw; cw->cin = ocin; cw->cout = ocout;
  cw->dtype = dtype; cw->nf = nf; cw->nslab = nslab; cw->ntaps = ntaps; cw->cout_frags = cout_frags;
  cw->bytes = bytes; cw->cin_split = cin_split; cw->nslab1 = cin_split ? nslab1 : nslab;
  hipError_t e = hipMalloc(&cw->dev, bytes);
  if (e != hipSuccess) { delete cw; flk_set_error("hipMalloc(%zu): %s", bytes, hipGetErrorString(e)); return FLK_ENOMEM; }
  e = hipMemcpy(cw->dev, host.data(), bytes, hipMemcpyHostToDevice);
  if (e != hipSuccess) { (void)hipFree(cw->dev); delete cw; flk_set_error("hipMemcpy: %s", hipGetErrorString(e)); return FLK_EHIP; }
  *out = cw;
  return FLK_OK;
}

extern "C" int flk_conv_weights_create(const float* w, int kt, int kh, int kw, int cin, int cout,
                                       const float* row_scale, int transpose, int dtype, int nf,
                                       flk_conv_weights** out) {
  return flk_conv_weights_create_impl(w, kt, kh, kw, cin, cout, row_scale, transpose, dtype, nf, 0, out);
}

extern "C" int flk_conv_weights_create_split(const float* w, int kt, int kh, int kw, int cin, int cout,
                                             const float* row_scale, int cin_split, int dtype, int nf,
                                             flk_conv_weights** out) {
  return flk_conv_weights_create_impl(w, kt, kh, kw, cin, cout, row_scale, 0, dtype, nf, cin_split, out);
}

extern "C" int flk_conv_weights_destroy(flk_conv_weights* w) {
  if (!w) return FLK_OK;
  if (w->dev) (void)hipFree(w->dev);
  delete w;
  return FLK_OK;
}
