// AddressSanitizer driver for the HOST side of libflicker_hip (linked against tests/asan/hip_stub.cpp instead of the
// HIP runtime): calls the C ABI the way the Python wrappers do -- packing, validation, whole-network plan construction
// for all four architectures, and one forward / backward launch sequence each (launches are dropped by the stub) --
// so that every host-side index computation, upload size and free() runs under ASan + LeakSanitizer.
//   abi_driver <weights file>...   (file: records of  int32 name_len | name | int64 numel | float[numel] , written by the test)
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <map>
#include <random>
#include <string>
#include <vector>
#include "../../include/flicker_hip.h"

extern "C" long flk_stub_live_allocs();
extern "C" long flk_stub_launches();

static int g_fail = 0;
#define EXPECT(cond, ...) do { if (!(cond)) { ++g_fail; fprintf(stderr, "FAIL %s:%d: %s -- ", __FILE__, __LINE__, #cond); fprintf(stderr, __VA_ARGS__); fprintf(stderr, " [%s]\n", flk_last_error()); } } while (0)

static std::vector<float> rnd(size_t n, unsigned seed) {
  std::mt19937 g(seed);
  std::normal_distribution<float> d(0.f, 1.f);
  std::vector<float> v(n);
  for (auto& x : v) x = d(g);
  return v;
}

static void packing() {
  struct Case { int kt, kh, kw, cin, cout, transpose, dtype, nf, split; };
  const Case cases[] = {
      {3, 3, 3, 96, 208, 0, FLK_BF16, 8, 0}, {3, 3, 3, 96, 208, 1, FLK_BF16, 4, 0}, {1, 1, 1, 480, 304, 0, FLK_BF16, 8, 0},
      {1, 1, 1, 304, 480, 0, FLK_BF16, 8, 192}, {1, 3, 3, 48, 232, 0, FLK_F32, 2, 0}, {3, 1, 1, 232, 128, 1, FLK_F32, 4, 0},
      {1, 1, 1, 8, 8, 0, FLK_BF16, 2, 0}, {4, 4, 4, 32, 64, 1, FLK_BF16, 2, 0}, {1, 1, 1, 1024, 400, 0, FLK_F32, 8, 0}};
  for (const Case& c : cases) {
    auto w = rnd((size_t)c.kt * c.kh * c.kw * c.cin * c.cout, 7);
    auto sc = rnd(c.transpose ? c.cout : c.cin, 8);
    flk_conv_weights* cw = nullptr;
    int rc = c.split ? flk_conv_weights_create_split(w.data(), c.kt, c.kh, c.kw, c.cin, c.cout, nullptr, c.split, c.dtype, c.nf, &cw)
                     : flk_conv_weights_create(w.data(), c.kt, c.kh, c.kw, c.cin, c.cout, c.transpose ? sc.data() : nullptr, c.transpose, c.dtype, c.nf, &cw);
    EXPECT(rc == FLK_OK && cw, "pack %dx%dx%d %d->%d", c.kt, c.kh, c.kw, c.cin, c.cout);
    EXPECT(flk_conv_weights_destroy(cw) == FLK_OK, "destroy");
  }
  // folded stem: structurally-zero chunks must be zero (accepted), a non-zero one must be refused
  std::vector<float> ws((size_t)64 * 32 * 64, 0.f);
  for (int dt = 0; dt < 4; ++dt) for (int dh = 0; dh < 4; ++dh) for (int dw = 0; dw < 4; ++dw) for (int ch = 0; ch < 32; ++ch) {
    const int qt = ch >> 4, qh = (ch >> 3) & 1;
    if ((dt == 3 && qt == 1) || (dh == 3 && qh == 1) || (ch & 7) >= 6) continue;
    for (int co = 0; co < 64; ++co) ws[((((size_t)dt * 4 + dh) * 4 + dw) * 32 + ch) * 64 + co] = 0.01f * (float)((dt + dh + dw + ch + co) % 13 - 6);
  }
  flk_conv_weights* cw = nullptr;
  EXPECT(flk_conv_weights_create_s2d_stem(ws.data(), 64, FLK_BF16, 4, &cw) == FLK_OK, "s2d stem bf16");
  flk_conv_weights_destroy(cw); cw = nullptr;
  EXPECT(flk_conv_weights_create_s2d_stem(ws.data(), 64, FLK_F32, 4, &cw) == FLK_OK, "s2d stem f32");
  flk_conv_weights_destroy(cw); cw = nullptr;
  ws[((((size_t)3 * 4 + 0) * 4 + 0) * 32 + 16) * 64] = 1.f;
  EXPECT(flk_conv_weights_create_s2d_stem(ws.data(), 64, FLK_BF16, 4, &cw) == FLK_EINVAL && !cw, "s2d stem must refuse a non-zero structural chunk");
  auto w7 = rnd((size_t)343 * 3 * 64, 3), s7 = rnd(64, 4);
  float* dev = nullptr;
  EXPECT(flk_stem_delta_grad_weights_create(w7.data(), s7.data(), &dev) == FLK_OK && dev, "stem delta-grad weights");
  flk_stem_delta_grad_weights_destroy(dev); dev = nullptr;
  EXPECT(flk_stem_delta_bias_weights_create(w7.data(), s7.data(), &dev) == FLK_OK && dev, "stem delta-bias weights");
  flk_stem_delta_grad_weights_destroy(dev);
  // the stem straight from the uint8 clip: 74 packed "taps" (37 K steps x 2 column parities); argument checks of the launch
  EXPECT(flk_stem_fwd_u8_weights_create(w7.data(), &cw) == FLK_OK && cw, "stem-from-uint8 weights");
  EXPECT(flk_stem_fwd_u8_weights_create(nullptr, &cw) == FLK_EINVAL, "stem-from-uint8 weights: null");
  {
    const int B = 1, T = 4;
    std::vector<uint8_t> clip((size_t)B * T * 224 * 224 * 3, 7);
    std::vector<float> delta((size_t)T * 3, 0.01f), sc(64, 1.f), bi(64, 0.f);
    std::vector<uint16_t> out((size_t)B * (T / 2) * 112 * 112 * 64);
    flk_apply_args a{};
    a.x = clip.data(); a.x_is_u8 = 1; a.x_scale = 1.f / 128; a.x_bias = -1.f; a.delta = delta.data(); a.dclip = 0.4f;
    a.inv_std[0] = a.inv_std[1] = a.inv_std[2] = 1.f; a.lo = -1.f; a.hi = 1.f; a.adv_flag = 1.f; a.B = B; a.T = T; a.H = 224; a.W = 224; a.fold_t = 3; a.center = 1;
    EXPECT(flk_stem_fwd_u8(&a, cw, sc.data(), bi.data(), nullptr, 0, out.data(), 64, nullptr) == FLK_OK, "stem-from-uint8 launch");
    flk_apply_args b = a; b.center = 0;
    EXPECT(flk_stem_fwd_u8(&b, cw, sc.data(), bi.data(), nullptr, 0, out.data(), 64, nullptr) == FLK_EINVAL, "stem-from-uint8 needs center = 1");
    b = a; b.H = 112;
    EXPECT(flk_stem_fwd_u8(&b, cw, sc.data(), bi.data(), nullptr, 0, out.data(), 64, nullptr) == FLK_EINVAL, "stem-from-uint8 needs 224 x 224");
    flk_conv_weights* other = nullptr;
    EXPECT(flk_conv_weights_create(w7.data(), 1, 1, 1, 32, 64, nullptr, 0, FLK_BF16, 4, &other) == FLK_OK, "a 1x1x1 weight object");
    EXPECT(flk_stem_fwd_u8(&a, other, sc.data(), bi.data(), nullptr, 0, out.data(), 64, nullptr) == FLK_EINVAL, "stem-from-uint8 refuses foreign weights");
    flk_conv_weights_destroy(other);
  }
  flk_conv_weights_destroy(cw);
}

static void validation() {
  flk_conv_weights* cw = nullptr;
  float one = 1.f;
  EXPECT(flk_conv_weights_create(nullptr, 1, 1, 1, 8, 8, nullptr, 0, FLK_BF16, 8, &cw) == FLK_EINVAL, "null weights");
  EXPECT(flk_conv_weights_create(&one, 1, 1, 1, 8, 8, nullptr, 0, 99, 8, &cw) == FLK_EINVAL, "bad dtype");
  EXPECT(flk_conv_weights_create(&one, 1, 1, 1, 8, 8, nullptr, 0, FLK_BF16, 3, &cw) == FLK_EINVAL, "bad nf");
  EXPECT(flk_conv_weights_create(&one, 0, 1, 1, 8, 8, nullptr, 0, FLK_BF16, 8, &cw) == FLK_EINVAL, "bad shape");
  EXPECT(strlen(flk_last_error()) > 0, "error text");
  EXPECT(flk_conv_weights_destroy(nullptr) == FLK_OK, "destroy(null)");
  EXPECT(flk_conv3d(nullptr, nullptr, FLK_BF16, nullptr) < 0, "conv3d(null)");
  EXPECT(flk_maxpool3d_fwd(nullptr, FLK_BF16, nullptr) < 0, "maxpool(null)");
  EXPECT(flk_perturb_apply_s2d(nullptr, nullptr, FLK_BF16, nullptr) < 0, "apply(null)");
  EXPECT(flk_softmax_adv_loss(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) < 0, "loss(null)");
  EXPECT(flk_perturb_reg_adam(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) < 0, "adam(null)");
  flk_net* n = nullptr;
  EXPECT(flk_net_create(77, FLK_BF16, 1, 16, 224, 224, 0, &n) == FLK_EINVAL && !n, "unknown arch");
  EXPECT(flk_net_create(FLK_NET_I3D, FLK_BF16, 0, 16, 224, 224, 0, &n) == FLK_EINVAL, "bad batch");
  EXPECT(flk_net_destroy(nullptr) == FLK_OK, "net destroy(null)");
  EXPECT(flk_net_workspace_bytes(nullptr) == 0 && flk_net_num_classes(nullptr) == 0, "null queries");
  EXPECT(flk_net_create(FLK_NET_I3D, FLK_BF16, 1, 16, 224, 224, 0, &n) == FLK_OK && n, "create");
  EXPECT(flk_net_finalize(n) == FLK_EINVAL, "finalize without weights must name the missing one");
  EXPECT(strstr(flk_last_error(), "missing weight") != nullptr, "error text: %s", flk_last_error());
  std::vector<float> lg(400);
  EXPECT(flk_net_forward(n, lg.data(), lg.data(), 1, nullptr) < 0, "forward before finalize");
  EXPECT(flk_net_destroy(n) == FLK_OK, "destroy unfinalized");
  EXPECT(flk_perturb_grad_scratch_bytes(8, 64, 224, 224) > 0 && flk_stem_delta_grad_scratch_bytes(8, 64, 224) > 0 && flk_dense_adam_scratch_bytes(64, 224, 224) > 0, "scratch sizes");
}

static std::map<std::string, std::vector<float>> load_weights(const char* path) {
  std::map<std::string, std::vector<float>> W;
  FILE* f = fopen(path, "rb");
  if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
  for (;;) {
    int32_t nl;
    if (fread(&nl, 4, 1, f) != 1) break;
    std::string name(nl, '\0');
    int64_t numel;
    if (fread(&name[0], 1, nl, f) != (size_t)nl || fread(&numel, 8, 1, f) != 1) { fprintf(stderr, "truncated %s\n", path); exit(2); }
    std::vector<float> v(numel);
    if (fread(v.data(), 4, numel, f) != (size_t)numel) { fprintf(stderr, "truncated %s\n", path); exit(2); }
    W[name] = std::move(v);
  }
  fclose(f);
  return W;
}

static void plan(int arch, int dtype, int B, int T, int HW, const std::map<std::string, std::vector<float>>& W, const char* tag) {
  flk_net* n = nullptr;
  EXPECT(flk_net_create(arch, dtype, B, T, HW, HW, 0, &n) == FLK_OK && n, "%s create", tag);
  for (auto& kv : W) EXPECT(flk_net_set_weight(n, kv.first.c_str(), kv.second.data(), (int64_t)kv.second.size()) == FLK_OK, "%s set %s", tag, kv.first.c_str());
  const long a0 = flk_stub_live_allocs();
  int rc = flk_net_finalize(n);
  EXPECT(rc == FLK_OK, "%s finalize", tag);
  if (rc) { flk_net_destroy(n); return; }
  EXPECT(flk_net_set_weight(n, "x", W.begin()->second.data(), 1) == FLK_EINVAL, "%s set_weight after finalize", tag);
  EXPECT(flk_net_workspace_bytes(n) > 0 && flk_net_num_classes(n) == 400, "%s queries", tag);
  const int64_t nin = flk_net_input_numel(n);
  // (VideoResNet plans: the (h,w)-folded clip, 16 channels -- 32 in bf16, where every value arrives as two bf16 numbers, fold_t = 4)
  EXPECT(nin == (arch == FLK_NET_I3D ? (int64_t)B * (T / 2) * (HW / 2) * (HW / 2) * 32
                                     : (int64_t)B * T * (HW / 2) * (HW / 2) * (dtype == FLK_BF16 ? 32 : 16)), "%s input numel", tag);
  std::vector<char> x((size_t)nin * 4), gx((size_t)nin * 4);
  std::vector<float> lg((size_t)B * 400), dl((size_t)B * 400, 0.01f);
  const long l0 = flk_stub_launches();
  EXPECT(flk_net_backward(n, dl.data(), gx.data(), nullptr) == FLK_ESTATE, "%s backward before forward", tag);
  EXPECT(flk_net_forward(n, x.data(), lg.data(), 1, nullptr) == FLK_OK, "%s forward (launch sequence)", tag);
  EXPECT(flk_net_backward(n, dl.data(), gx.data(), nullptr) == FLK_OK, "%s backward (launch sequence)", tag);
  EXPECT(flk_stub_launches() - l0 > 40, "%s: only %ld launches", tag, flk_stub_launches() - l0);
  EXPECT(flk_net_profile(n, 1) == FLK_OK && flk_net_forward(n, x.data(), lg.data(), 1, nullptr) == FLK_OK &&
         flk_net_backward(n, dl.data(), gx.data(), nullptr) == FLK_OK, "%s profiled pass", tag);
  std::vector<char> js(1 << 20);
  EXPECT(flk_net_profile_read(n, js.data(), (int64_t)js.size()) == FLK_OK && js[0] == '[', "%s profile_read", tag);
  EXPECT(flk_net_profile_read(n, js.data(), 16) == FLK_EINVAL, "%s profile_read small buffer", tag);
  flk_net_profile(n, 0);
  int64_t dims[5];
  const char* ep = arch == FLK_NET_I3D ? "Mixed_5c" : "layer4.1";
  EXPECT(flk_net_get_activation(n, ep, nullptr, 0, dims) == FLK_OK && dims[0] == B, "%s activation dims", tag);
  std::vector<float> act((size_t)dims[0] * dims[1] * dims[2] * dims[3] * dims[4]);
  EXPECT(flk_net_get_activation(n, ep, act.data(), (int64_t)act.size(), dims) == FLK_OK, "%s activation read-back", tag);
  EXPECT(flk_net_get_activation(n, ep, act.data(), (int64_t)act.size() - 1, dims) == FLK_EINVAL, "%s activation small buffer", tag);
  EXPECT(flk_net_get_activation(n, "no such endpoint", nullptr, 0, dims) == FLK_EINVAL, "%s unknown endpoint", tag);
  if (arch == FLK_NET_I3D && dtype == FLK_BF16) {
    // the fused flicker paths (stem position-bias forward, fused delta-gradient backward) with a real apply struct
    std::vector<uint8_t> clip((size_t)B * T * HW * HW * 3, 128);
    std::vector<float> delta((size_t)T * 3, 0.01f), gd((size_t)T * 3);
    std::vector<float> scratch((size_t)flk_stem_delta_grad_scratch_bytes(B, T, HW) / 4 + 16);
    flk_apply_args a{};
    a.x = clip.data(); a.x_is_u8 = 1; a.x_scale = 1.f / 128; a.x_bias = -1.f; a.delta = delta.data(); a.dclip = 0.4f;
    a.inv_std[0] = a.inv_std[1] = a.inv_std[2] = 1.f; a.lo = -1.f; a.hi = 1.f; a.adv_flag = 1.f; a.B = B; a.T = T; a.H = HW; a.W = HW; a.fold_t = 3; a.center = 1;
    EXPECT(flk_net_has_forward_flicker(n) == 1 && flk_net_has_backward_delta(n) == 1, "%s fused flicker paths", tag);
    EXPECT(flk_perturb_apply_s2d(&a, x.data(), FLK_BF16, nullptr) == FLK_OK, "%s apply", tag);
    EXPECT(flk_net_forward_flicker(n, x.data(), &a, lg.data(), nullptr) == FLK_OK, "%s forward_flicker", tag);
    EXPECT(flk_net_backward_delta(n, dl.data(), &a, gd.data(), scratch.data(), nullptr) == FLK_OK, "%s backward_delta", tag);
    EXPECT(flk_net_prepare_backward_delta(n, &a, scratch.data(), nullptr) == FLK_OK, "%s prepare_backward_delta (clip mask beside the forward pass)", tag);
    EXPECT(flk_net_forward_apply(n, &a, x.data(), lg.data(), nullptr) == FLK_OK, "%s forward_apply (apply inside the plan, per batch slice)", tag);
    EXPECT(flk_net_backward_delta(n, dl.data(), &a, gd.data(), scratch.data(), nullptr) == FLK_OK, "%s backward_delta behind a prepared mask", tag);
    EXPECT(flk_net_prepare_backward_delta(n, nullptr, scratch.data(), nullptr) == FLK_EINVAL, "%s prepare_backward_delta: null arguments", tag);
    {   // per-clip perturbations: one delta, one position-bias table and one clamp bound per clip
      std::vector<float> dpc((size_t)B * T * 3, 0.02f), gpc((size_t)B * T * 3), bounds(B, 0.3f);
      flk_apply_args pc = a; pc.delta = dpc.data(); pc.delta_per_clip = 1; pc.dclip_dev = bounds.data();
      EXPECT(flk_net_forward_apply(n, &pc, x.data(), lg.data(), nullptr) == FLK_OK, "%s forward_apply per clip", tag);
      EXPECT(flk_net_backward_delta(n, dl.data(), &pc, gpc.data(), scratch.data(), nullptr) == FLK_OK, "%s backward_delta per clip", tag);
      std::vector<int> steps(B, 0), active(B, 1);
      std::vector<float> m((size_t)B * T * 3, 0.f), v((size_t)B * T * 3, 0.f), sc((size_t)B * 8);
      flk_adam_args ad{}; ad.T = T; ad.beta0 = 1.f; ad.beta1 = ad.beta2 = ad.beta3 = 0.5f; ad.lr = 1e-3f; ad.adam_b1 = 0.9f; ad.adam_b2 = 0.999f; ad.adam_eps = 1e-8f; ad.g_scale = 1.f;
      EXPECT(flk_perturb_reg_adam_batched(&ad, B, gpc.data(), dpc.data(), m.data(), v.data(), steps.data(), active.data(), nullptr, sc.data(), nullptr) == FLK_OK,
             "%s batched reg + Adam launch", tag);
      EXPECT(flk_perturb_reg_adam_batched(&ad, 0, gpc.data(), dpc.data(), m.data(), v.data(), steps.data(), active.data(), nullptr, sc.data(), nullptr) == FLK_EINVAL,
             "%s batched reg + Adam: bad clip count", tag);
    }
    flk_apply_args wrong = a; wrong.fold_t = 2;
    EXPECT(flk_net_forward_apply(n, &wrong, x.data(), lg.data(), nullptr) == FLK_EINVAL, "%s forward_apply layout check", tag);
    flk_apply_args bad = a; bad.T = T + 2;
    EXPECT(flk_net_backward_delta(n, dl.data(), &bad, gd.data(), scratch.data(), nullptr) == FLK_EINVAL, "%s backward_delta geometry check", tag);
  }
  if (arch != FLK_NET_I3D) {
    // the VideoResNet input path with a real apply struct: fp32 clip, torch-dialect perturbation; bf16 plans take the clip as two bf16
    // numbers per value (fold_t = 4, 32 channels), fp32 plans the 16-channel fold; the gradient comes back in the 16-channel layout
    std::vector<float> clip((size_t)B * T * HW * HW * 3, 0.25f), delta((size_t)T * 3, 0.01f), gd((size_t)T * 3);
    std::vector<float> scratch((size_t)flk_perturb_grad_scratch_bytes(B, T, HW, HW) / 4 + 16);
    flk_apply_args a{};
    a.x = clip.data(); a.x_scale = 1.f; a.delta = delta.data(); a.dclip = 0.2f;
    a.inv_std[0] = a.inv_std[1] = a.inv_std[2] = 4.5f; a.lo = -1.7f; a.hi = 2.4f; a.adv_flag = 1.f; a.B = B; a.T = T; a.H = HW; a.W = HW;
    a.fold_t = dtype == FLK_BF16 ? 4 : 1;
    EXPECT(flk_net_forward_apply(n, &a, x.data(), lg.data(), nullptr) == FLK_OK, "%s forward_apply (fold_t %d)", tag, a.fold_t);
    EXPECT(flk_net_backward(n, dl.data(), gx.data(), nullptr) == FLK_OK, "%s backward behind forward_apply", tag);
    EXPECT(flk_perturb_grad_reduce(&a, gx.data(), dtype, gd.data(), scratch.data(), nullptr) == FLK_OK, "%s grad_reduce (16-channel gradient layout)", tag);
    flk_apply_args wrong = a; wrong.fold_t = dtype == FLK_BF16 ? 1 : 4;
    EXPECT(flk_net_forward_apply(n, &wrong, x.data(), lg.data(), nullptr) == FLK_EINVAL, "%s forward_apply layout check", tag);
    if (dtype == FLK_F32) { flk_apply_args hl = a; hl.fold_t = 4; EXPECT(flk_perturb_apply_s2d(&hl, x.data(), FLK_F32, nullptr) == FLK_EINVAL, "%s fold_t 4 is bf16 only", tag); }
  }
  EXPECT(flk_net_destroy(n) == FLK_OK, "%s destroy", tag);
  EXPECT(flk_stub_live_allocs() == a0, "%s: %ld device allocations leaked", tag, flk_stub_live_allocs() - a0);
}

int main(int argc, char** argv) {
  EXPECT(flk_version() >= 100, "version");
  packing();
  validation();
  for (int i = 1; i < argc; ++i) {
    auto W = load_weights(argv[i]);
    const bool i3d = W.count("RGB/inception_i3d/Conv3d_1a_7x7/conv_3d/w") > 0;
    if (i3d) {
      plan(FLK_NET_I3D, FLK_BF16, 2, 16, 224, W, "i3d bf16 bs2");
      plan(FLK_NET_I3D, FLK_BF16, 4, 16, 224, W, "i3d bf16 bs4 (stem split)");
      plan(FLK_NET_I3D, FLK_F32, 1, 18, 224, W, "i3d f32 T18");
    } else {
      const bool r21 = W.count("stem.3.weight") > 0, mc3 = !r21 && W.at("layer2.0.conv1.0.weight").size() == (size_t)128 * 64 * 9;
      const int arch = r21 ? FLK_NET_R2PLUS1D_18 : mc3 ? FLK_NET_MC3_18 : FLK_NET_R3D_18;
      plan(arch, FLK_BF16, 1, 16, 112, W, r21 ? "r2plus1d_18 bf16" : mc3 ? "mc3_18 bf16" : "r3d_18 bf16");
      plan(arch, FLK_F32, 2, 8, 112, W, r21 ? "r2plus1d_18 f32" : mc3 ? "mc3_18 f32" : "r3d_18 f32");
    }
  }
  EXPECT(flk_stub_live_allocs() == 0, "%ld device allocations alive at exit", flk_stub_live_allocs());
  if (g_fail) { fprintf(stderr, "%d expectation(s) failed\n", g_fail); return 1; }
  printf("asan driver ok: %ld launches accepted\n", flk_stub_launches());
  return 0;
}
