// Error plumbing, version, and host-side packing of convolution weights into MFMA fragment order.
#include <stdarg.h>
#include <string.h>
#include <new>
#include "flk_internal.h"

static thread_local char g_err[1024] = "";
thread_local const char* flk_last_kernel_tag = "";
thread_local hipEvent_t flk_stop_event = nullptr;
thread_local unsigned flk_launch_count = 0;

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
  FLK_REQUIRE(nf == 2 || nf == 4 || nf == 8 || (nf == 6 && dtype == FLK_BF16), "flk_conv_weights_create: nf must be 2, 4, 8 (or 6 in bf16: 96-channel tiles) (got %d)", nf);
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

// Folded 7x7x7/2 stem (fold_t = 3 layout, include/flicker_hip.h): a 4x4x4 convolution over 32 channels = 4 chunks of 8,
// chunk = (qt,qh).  Tap index 3 of an axis only exists for parity 0 (kernel index 2*3+1 = 7 is outside the 7-tap
// window), so for taps with dt == 3 the chunks qt = 1 are zero, for dh == 3 the chunks qh = 1.  The K steps of mode 4
// (conv_igemm.hip) are assembled from the non-zero chunks only, 49 steps instead of 64:
//   class 0 (dt<3, dh<3): one tap, chunks 0..3              class 1 (dt=3): taps (dw, dw+1) x chunks {0,1}
//   class 2 (dh=3): taps (dw, dw+1) x chunks {0,2}           class 3 (dt=3, dh=3): taps dw..dw+3 x chunk 0
// in the order dt, dh, dw.  They are packed as a 49-"tap" convolution over 32 channels whose channel (q*8 + j) of
// step s is (tap, chunk)(s, q) x element j.
extern "C" int flk_conv_weights_create_s2d_stem(const float* w, int cout, int dtype, int nf, flk_conv_weights** out) {
  FLK_REQUIRE(w && out && cout > 0, "flk_conv_weights_create_s2d_stem: bad argument");
  if (dtype != FLK_BF16) return flk_conv_weights_create_impl(w, 4, 4, 4, 32, cout, nullptr, 0, dtype, nf, 0, out);
  auto at = [&](int dt, int dh, int dw, int ch, int co) { return w[((((size_t)dt * 4 + dh) * 4 + dw) * 32 + ch) * cout + co]; };
  for (int dt = 0; dt < 4; ++dt) for (int dh = 0; dh < 4; ++dh) for (int dw = 0; dw < 4; ++dw)
    for (int ch = 0; ch < 32; ++ch) {
      const int qt = ch >> 4, qh = (ch >> 3) & 1;
      if (!((dt == 3 && qt == 1) || (dh == 3 && qh == 1))) continue;
      for (int co = 0; co < cout; ++co)
        FLK_REQUIRE(at(dt, dh, dw, ch, co) == 0.f, "flk_conv_weights_create_s2d_stem: weight (tap %d,%d,%d, channel %d) must be zero", dt, dh, dw, ch);
    }
  std::vector<float> v;
  v.reserve((size_t)49 * 32 * cout);
  for (int dt = 0; dt < 4; ++dt)
    for (int dh = 0; dh < 4; ++dh) {
      const int cls = (dt == 3) + 2 * (dh == 3), wstride = cls == 0 ? 1 : cls == 3 ? 4 : 2;
      for (int dw = 0; dw < 4; dw += wstride)
        for (int q = 0; q < 4; ++q) {
          const int chunk = cls == 0 ? q : cls == 1 ? (q & 1) : cls == 2 ? (q & 1) * 2 : 0;
          const int woff = cls == 0 ? 0 : cls == 3 ? q : (q >> 1);
          for (int j = 0; j < 8; ++j)
            for (int co = 0; co < cout; ++co) v.push_back(at(dt, dh, dw + woff, chunk * 8 + j, co));
        }
    }
  FLK_REQUIRE(v.size() == (size_t)49 * 32 * cout, "flk_conv_weights_create_s2d_stem: internal step count");
  int rc = flk_conv_weights_create_impl(v.data(), 49, 1, 1, 32, cout, nullptr, 0, dtype, nf, 0, out);
  if (rc) return rc;
  (*out)->stem4 = 1;
  (*out)->kt = (*out)->kh = (*out)->kw = 4;      // the operator it implements; ntaps stays 49 (K steps per slab)
  return FLK_OK;
}

extern "C" int flk_conv_weights_destroy(flk_conv_weights* w) {
  if (!w) return FLK_OK;
  if (w->dev) (void)hipFree(w->dev);
  delete w;
  return FLK_OK;
}
