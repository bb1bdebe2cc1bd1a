// asm_model_plan: the Assemble-ResNet topology as a flat layer list, through the C ABI (host code only, no kernels).
//
// The reference builds its graph by walking Python (functions/model_fns.py:138-198 -> nets/resnet_model.py:305-599);
// variables get TensorFlow's auto-numbered names inside nested variable scopes.  This planner walks the same rules in
// C++ from an asm_model_cfg (the flag surface of nets/hparams_config.py that reaches the network) and emits one entry
// per layer in creation order: variable-owning layers (convolution kernel, batch norm, dense) with their TF scope name,
// shapes and offsets in tf.trainable_variables() order, and the parameter-free ops in between.  Topology parity with the
// reference is therefore checkable without a GPU: tests/test_plan_abi.py expands the entries to variable names / shapes
// and compares them with the variables the reference's own code created (tests/golden/reference_taps.json).
#include <map>
#include <string>
#include <vector>

#include "common.h"

namespace {

struct Planner {
  const asm_model_cfg& c;
  std::vector<asm_plan_entry> out;
  std::vector<std::string> scope;
  std::map<std::string, int> scope_counts;   // tf variable_scope uniquifier: opened scopes per full name
  int64_t param_off = 0;
  int64_t macs = 0;
  int N;

  Planner(const asm_model_cfg& cfg, int n) : c(cfg), N(n) {}

  std::string join() const {
    std::string s = "resnet_model";
    for (auto& p : scope) s += "/" + p;
    return s;
  }
  // tf.variable_scope(None, default_name): first of name, name_1, ... not yet opened under the current scope
  std::string unique(const std::string& base) {
    const std::string parent = join();
    std::string name = base;
    for (int idx = 0;; ++idx) {
      name = idx ? base + "_" + std::to_string(idx) : base;
      if (scope_counts[parent + "/" + name] == 0) break;
    }
    scope_counts[parent + "/" + name] += 1;
    return name;
  }
  void push(const std::string& default_name) { scope.push_back(unique(default_name)); }
  void pop() {
    const std::string full = join();   // close_variable_subscopes: forget the sub-scopes of the scope being left
    for (auto& kv : scope_counts)
      if (kv.first.size() > full.size() && kv.first.compare(0, full.size() + 1, full + "/") == 0) kv.second = 0;
    scope.pop_back();
  }

  asm_plan_entry& add(int kind, const std::string& layer, int H, int W, int C, int K, int R, int S, int stride, int Ho,
                      int Wo, int flags, int n_train, int64_t elems) {
    asm_plan_entry e{};
    e.kind = kind;
    e.N = N; e.H = H; e.W = W; e.C = C; e.K = K; e.R = R; e.S = S; e.stride = stride; e.Ho = Ho; e.Wo = Wo;
    e.flags = flags;
    e.trainable = n_train;
    e.param_offset = n_train ? param_off : -1;
    e.param_elems = elems;
    param_off += elems;
    const std::string full = layer.empty() ? join() : join() + "/" + layer;
    snprintf(e.name, sizeof(e.name), "%s", full.c_str());
    out.push_back(e);
    return out.back();
  }

  struct T {
    int H, W, C;
  };
  static int out_size(int in, int stride) { return stride == 1 ? in : (in - 1) / stride + 1; }

  // conv2d_fixed_padding (nets/model_helper.py:67-78)
  T conv(T x, int K, int k, int stride, const char* layer_name = nullptr) {
    const std::string layer = layer_name ? std::string(layer_name) : unique("conv2d");
    const int Ho = out_size(x.H, stride), Wo = out_size(x.W, stride);
    add(ASM_PLAN_CONV, layer, x.H, x.W, x.C, K, k, k, stride, Ho, Wo, 0, 1, (int64_t)k * k * x.C * K);
    macs += (int64_t)Ho * Wo * K * x.C * k * k;
    return T{Ho, Wo, K};
  }
  // batch_norm (+ the relu / residual add fused behind it by the kernels)
  T bn(T x, int flags, const char* layer_name = nullptr) {
    const std::string layer = layer_name ? std::string(layer_name) : unique("batch_normalization");
    add(ASM_PLAN_BN, layer, x.H, x.W, x.C, x.C, 1, 1, 1, x.H, x.W, flags, 2, 2 * (int64_t)x.C);
    return x;
  }
  T op(int kind, T x, int k, int stride, int Ho, int Wo, int Cout, int flags = 0) {
    add(kind, "", x.H, x.W, x.C, Cout, k, k, stride, Ho, Wo, flags, 0, 0);
    return T{Ho, Wo, Cout};
  }

  // blocks.sk_conv2d (nets/blocks.py:110-154)
  T sk(T x, int filters, int stride) {
    T f = conv(x, 2 * filters, 3, stride);
    bn(f, ASM_PLAN_RELU);
    const int d = filters / 2 > 32 ? filters / 2 : 32;
    push("sk_block");
    T s{1, 1, filters};
    op(ASM_PLAN_SK_GAP, f, 1, 1, 1, 1, filters);
    T z = conv(s, d, 1, 1, "sk_fc_1");
    bn(z, ASM_PLAN_RELU);
    conv(z, 2 * filters, 1, 1, "sk_fc_2");
    pop();
    return op(ASM_PLAN_SK_SELECT, f, 1, 1, f.H, f.W, filters);
  }
  // blocks.se_block (nets/blocks.py:156-184)
  T se(T x) {
    push("se_block");
    op(ASM_PLAN_GAP, x, 1, 1, 1, 1, x.C);
    T sq{1, 1, x.C};
    T e1 = conv(sq, x.C / 16, 1, 1, "seblock_dense_1");
    conv(e1, x.C, 1, 1, "seblock_dense_2");
    pop();
    return op(ASM_PLAN_SE_SCALE, x, 1, 1, x.H, x.W, x.C);
  }
  T blur(T x, int k, int stride) {
    const int pad = (k - 1) / 2;
    return op(ASM_PLAN_BLURPOOL, x, k, stride, (x.H + 2 * pad - k) / stride + 1, (x.W + 2 * pad - k) / stride + 1, x.C);
  }

  enum Shortcut { PLAIN, RESNET_D, BL };

  // _bottleneck_block_v1 (nets/resnet_model.py:35-97)
  T bottleneck(T x, int filters, bool project, Shortcut sc, int strides, int aa_size, int aa_type, bool last_relu) {
    const int cout = 4 * filters;
    if (project) {
      T s = x;
      if (sc == RESNET_D) {               // :123-131 (avg-pool 2x2 even at stride 1)
        s = op(ASM_PLAN_AVGPOOL, s, 2, strides, strides > 1 ? (s.H + 1 - 2) / strides + 1 : s.H,
               strides > 1 ? (s.W + 1 - 2) / strides + 1 : s.W, s.C, strides == 1 ? ASM_PLAN_COUNT_VALID : 0);
        s = conv(s, cout, 1, 1);
      } else if (sc == BL) {              // :133-141
        if (strides > 1) s = op(ASM_PLAN_AVGPOOL, s, 3, strides, (s.H + 2 - 3) / strides + 1, (s.W + 2 - 3) / strides + 1, s.C);
        s = conv(s, cout, 1, 1);
      } else if ((aa_type & ASM_AA_PROJ) && strides != 1) {   // :107-115
        s = blur(s, aa_size, strides);
        s = conv(s, cout, 1, 1);
      } else {
        s = conv(s, cout, 1, strides);
      }
      bn(s, 0);
    }
    T h = conv(x, filters, 1, 1);
    bn(h, ASM_PLAN_RELU);
    const int s3 = (aa_type & ASM_AA_SCONV) ? 1 : strides;
    if (c.use_sk_block) {
      h = sk(h, filters, s3);
    } else {
      h = conv(h, filters, 3, s3);
      bn(h, ASM_PLAN_RELU);
    }
    if ((aa_type & ASM_AA_SCONV) && strides != 1) h = blur(h, aa_size, strides);
    h = conv(h, cout, 1, 1);
    int fl = c.zero_gamma ? ASM_PLAN_ZERO_GAMMA : 0;
    if (c.use_se_block) {
      bn(h, fl);
      h = se(h);
      op(ASM_PLAN_ADD, h, 1, 1, h.H, h.W, h.C, last_relu ? ASM_PLAN_RELU : 0);
    } else {
      bn(h, fl | ASM_PLAN_RESIDUAL | (last_relu ? ASM_PLAN_RELU : 0));
    }
    return h;
  }

  // block_layer (nets/resnet_model.py:99-163)
  T block_layer(T x, int filters, int num_blocks, int strides, Shortcut sc, bool last_relu) {
    x = bottleneck(x, filters, true, sc, strides, c.anti_alias_filter_size, c.anti_alias_type, true);
    for (int i = 1; i < num_blocks; ++i)
      x = bottleneck(x, filters, false, sc, 1, 0, 0, i == num_blocks - 1 ? last_relu : true);
    return x;
  }

  int walk(int H, int W) {
    static const int v1[4][4] = {{3, 4, 6, 3}, {3, 4, 23, 3}, {3, 8, 36, 3}, {3, 24, 36, 3}};
    static const int v2[3][4] = {{3, 4, 6, 3}, {4, 8, 18, 3}, {5, 12, 30, 3}};
    const int* bs = nullptr;
    const bool bl = c.resnet_version == 2;
    switch (c.resnet_size) {   // functions/model_fns.py:98-135
      case 50: bs = bl ? v2[0] : v1[0]; break;
      case 101: bs = bl ? v2[1] : v1[1]; break;
      case 152: bs = bl ? v2[2] : v1[2]; break;
      case 200: bs = bl ? nullptr : v1[3]; break;
      default: break;
    }
    if (!bs) ASM_FAIL(ASM_EINVAL, "Could not find layers for selected Resnet size. Size received: %d", c.resnet_size);
    int strides[4] = {1, 2, 2, 2};
    if (bl) { strides[0] = 2; strides[1] = 2; strides[2] = 1; strides[3] = 2; }
    if (c.no_downsample) strides[3] = 1;
    const int nf = 64;
    T x{H, W, 3};
    const bool d = c.use_resnet_d != 0;
    // ---- stem :328-381 ----
    if (d) {
      if (bl) push("stage0");
      x = conv(x, nf / 2, 3, 2); bn(x, ASM_PLAN_RELU);
      x = conv(x, nf / 2, 3, 1); bn(x, ASM_PLAN_RELU);
      x = conv(x, nf, 3, 1);
      if (bl) pop();
    } else {
      if (bl) push("stage0");
      x = conv(x, nf, 7, 2);
      if (bl) pop();
    }
    if (bl) push("stage0");
    bn(x, ASM_PLAN_RELU);
    if (bl) pop();
    // ---- first pool :383-425 ----
    if (bl) {
      push("stage0/pool");
      T big0 = conv(x, nf, 3, 2); bn(big0, 0);
      T l0 = conv(x, nf / c.bl_alpha, 3, 1); bn(l0, ASM_PLAN_RELU);
      l0 = conv(l0, nf / c.bl_alpha, 3, 2); bn(l0, ASM_PLAN_RELU);
      l0 = conv(l0, nf, 1, 1); bn(l0, ASM_PLAN_RELU | ASM_PLAN_RESIDUAL);
      x = conv(l0, nf, 1, 1); bn(x, ASM_PLAN_RELU);
      pop();
    } else {
      x = op(ASM_PLAN_MAXPOOL, x, 3, 2, (x.H + 1) / 2, (x.W + 1) / 2, x.C);
    }
    // ---- stages :445-549 ----
    for (int i = 0; i < 4; ++i) {
      const int f = nf << i, nb = bs[i];
      if (bl && i < 3) {
        push("stage" + std::to_string(i + 1));
        push("big" + std::to_string(i + 1));
        T big = block_layer(x, f, nb - 1, 2, BL, false);
        pop();
        push("little" + std::to_string(i + 1));
        const int nl = nb / c.bl_beta - 1 > 1 ? nb / c.bl_beta - 1 : 1;
        T little = block_layer(x, f / c.bl_alpha, nl, 1, BL, true);
        T le = conv(little, 4 * f, 1, 1);
        bn(le, ASM_PLAN_RELU | ASM_PLAN_RESIDUAL | ASM_PLAN_UPSAMPLED_RESIDUAL);   // relu(little_e + UpSampling2D(big)) :493-501
        (void)big;
        pop();
        push("merge" + std::to_string(i + 1));
        x = block_layer(le, f, 1, strides[i], BL, true);
        pop();
        pop();
      } else if (bl) {
        push("stage" + std::to_string(i + 1));
        x = block_layer(x, f, nb, strides[i], d ? RESNET_D : BL, true);
        pop();
      } else {
        x = block_layer(x, f, nb, strides[i], d ? RESNET_D : PLAIN, true);
      }
    }
    // ---- head :555-599 ----
    if (c.pool_type == ASM_POOL_FLATTEN) x = op(ASM_PLAN_FLATTEN, x, 1, 1, 1, 1, x.H * x.W * x.C);
    else x = op(c.pool_type == ASM_POOL_GEM ? ASM_PLAN_GEM : ASM_PLAN_GAP, x, 1, 1, 1, 1, x.C);
    if (c.embedding_size > 0) {
      x = conv(x, c.embedding_size, 1, 1, "embedding_dense");
      bn(x, ASM_PLAN_RELU, "embedding_dense_batch_normalization");
    }
    const std::string dn = unique("dense");
    add(ASM_PLAN_DENSE, dn, 1, 1, x.C, c.num_classes, 1, 1, 1, 1, 1, 0, 2, (int64_t)x.C * c.num_classes + c.num_classes);
    macs += (int64_t)x.C * c.num_classes;
    return ASM_OK;
  }
};

}  // namespace

extern "C" int asm_model_plan(const asm_model_cfg* cfg, int N, int H, int W, asm_plan_entry* entries, int capacity,
                              asm_plan_summary* summary) {
  ASM_REQUIRE(cfg && summary, "model_plan: null pointer");
  ASM_REQUIRE(N > 0 && H >= 32 && W >= 32, "model_plan: bad input size");
  if (cfg->resnet_version != 1 && cfg->resnet_version != 2)     // nets/resnet_model.py:200-203
    ASM_FAIL(ASM_EINVAL, "Resnet version should be 1 or 2. See README for citations.");
  if (cfg->resnet_size < 50) ASM_FAIL(ASM_ENOTSUP, "only bottleneck ResNets (resnet_size >= 50)");   // :205-212
  if (cfg->dtype != ASM_BF16)                                   // fp16 / fp32 are the reference's; this path computes in bf16
    ASM_FAIL(ASM_ENOTSUP, "the MI355X path computes in bf16 with fp32 master weights (dtype %d)", cfg->dtype);
  ASM_REQUIRE(cfg->num_classes > 0 && cfg->bl_alpha > 0 && cfg->bl_beta > 0, "model_plan: bad num_classes / bl_alpha / bl_beta");
  if (cfg->pool_type < ASM_POOL_GAP || cfg->pool_type > ASM_POOL_FLATTEN) ASM_FAIL(ASM_ENOTSUP, "unknown pool_type");
  if ((cfg->anti_alias_type & (ASM_AA_SCONV | ASM_AA_PROJ)) && cfg->anti_alias_filter_size < 2)
    ASM_FAIL(ASM_ENOTSUP, "anti_alias_filter_size=1 hard-codes NCHW slicing in the reference (blocks.py:79-84)");
  Planner p(*cfg, N);
  if (int e = p.walk(H, W)) return e;
  summary->n_entries = (int32_t)p.out.size();
  summary->trainable_elems = p.param_off;
  summary->forward_macs_per_image = p.macs;
  summary->trainable_tensors = 0;
  size_t ws = 0;
  for (auto& e : p.out) {
    summary->trainable_tensors += e.trainable;
    if (e.kind == ASM_PLAN_CONV && e.C % 8 == 0) {
      asm_conv_desc d{};
      d.N = N; d.H = e.H; d.W = e.W; d.C = e.C; d.K = e.K; d.R = e.R; d.S = e.S; d.stride = e.stride;
      d.pad = (e.R - 1) / 2; d.Ho = e.Ho; d.Wo = e.Wo;
      const size_t need = asm_conv2d_wgrad_workspace_bytes(&d);
      if (need > ws) ws = need;
    }
  }
  summary->wgrad_workspace_bytes = (int64_t)ws;
  if (entries) {
    ASM_REQUIRE(capacity >= summary->n_entries, "model_plan: %d entries do not fit capacity %d", summary->n_entries, capacity);
    for (size_t i = 0; i < p.out.size(); ++i) entries[i] = p.out[i];
  }
  return ASM_OK;
}


// ---- kernel launches made by this library, process-wide (bench.py: kernels per training step) ----------------------------
#include <atomic>
namespace {
std::atomic<unsigned long long> g_launches{0};
}
void asm_count_launch() { g_launches.fetch_add(1, std::memory_order_relaxed); }
extern "C" unsigned long long asm_launch_count(void) { return g_launches.load(std::memory_order_relaxed); }

// ---- kernel-selection overrides (include/asm_hip.h: asm_tuning) ----------------------------------------------------------
namespace {
asm_tuning make_default_tuning() {
  asm_tuning t = {};
  t.igemm_mode = 0; t.igemm_tile = 0; t.igemm_pfa = -1; t.dgrad_parity = 2; t.wgrad_halo = 1; t.wgrad_big = -1; t.wgrad_splits = 0;
  t.bn_rows = 1024; t.igemm3 = 3; t.gemm1 = -1; t.wgrad_ring = -1; t.igemm8 = 1;
  return t;
}
asm_tuning g_tuning = make_default_tuning();
}  // namespace

extern "C" const asm_tuning* asm_tuning_current(void) { return &g_tuning; }
extern "C" void asm_tuning_defaults(asm_tuning* t) {
  if (t) *t = make_default_tuning();
}
extern "C" int asm_set_tuning(const asm_tuning* t) {
  if (t && (t->bn_rows <= 0 || t->igemm_mode < 0 || t->igemm_mode > 1 || (t->igemm_tile != 0 && t->igemm_tile != 1 && t->igemm_tile != 3) ||
            t->igemm_pfa < -1 || t->igemm_pfa > 1 || t->dgrad_parity < 0 || t->dgrad_parity > 2 || t->wgrad_halo < 0 || t->wgrad_halo > 2 ||
            t->wgrad_big < -1 || t->wgrad_big > 1 || t->wgrad_splits < 0 || t->igemm3 < 0 || t->igemm3 > 3 || t->gemm1 < -2 ||
            t->wgrad_ring < -1 || t->igemm8 < 0 || t->igemm8 > 2))
    ASM_FAIL(ASM_EINVAL, "asm_set_tuning: field out of range");
  g_tuning = t ? *t : make_default_tuning();
  return ASM_OK;
}
extern "C" void asm_get_tuning(asm_tuning* t) {
  if (t) *t = g_tuning;
}
