// api_weights.hip -- the state dict of lib/sttran.py:316-372 (+ lib/transformer.py:116-127, lib/dsg_detr.py:466-512) as the
// handle holds it: declaration, sttran_create / sttran_load_tensor / sttran_finalize_weights (derived parameters, the bf16
// planes of the bf16x3 engine are made on demand in api_forward.hip).
#include "api_internal.h"

using namespace sttran;
using namespace sttran_host;

namespace sttran_host {

int fail(SttranHandle* h, int code, const std::string& msg) {
  if (h) h->err = msg;
  return code;
}

// weight tensors: hipMalloc, or a guarded allocation under STTRAN_GUARD_WORKSPACE
hipError_t weight_alloc(Tensor& t, size_t bytes) {
  if (guard_workspace()) {
    void* p = nullptr;
    if (sttran_debug_guarded_alloc(bytes, &p, &t.d_guard) != STTRAN_OK) return hipErrorOutOfMemory;
    t.d = static_cast<float*>(p);
    return hipSuccess;
  }
  return hipMalloc(reinterpret_cast<void**>(&t.d), bytes);
}
void weight_free(Tensor& t) {
  if (t.d_guard) sttran_debug_guarded_free(t.d_guard);
  else if (t.d) hipFree(t.d);
  t.d = nullptr; t.d_guard = nullptr;
}

void add(SttranHandle* h, const std::string& k, std::vector<int64_t> shape, bool required, bool gemm_weight = false) {
  Tensor t;
  t.shape = std::move(shape);
  t.n = 1;
  for (int64_t d : t.shape) t.n *= (size_t)d;
  t.required = required;
  if (gemm_weight) t.ld = pad32(t.shape.back());
  h->w[k] = t;
}

// gemm = the weight is the B operand of run_linear: stored with zero-padded rows (B_KMAJOR_PAD contract)
void add_linear(SttranHandle* h, const std::string& p, int64_t out, int64_t in, bool req, bool gemm = true) {
  add(h, p + ".weight", {out, in}, req, gemm);
  add(h, p + ".bias", {out}, req);
}
void add_bn(SttranHandle* h, const std::string& p, int64_t n, bool req) {
  for (const char* s : {".weight", ".bias", ".running_mean", ".running_var"}) add(h, p + s, {n}, req);
}
void add_mha(SttranHandle* h, const std::string& p, int64_t d) {
  add(h, p + ".in_proj_weight", {3 * d, d}, true, true);
  add(h, p + ".in_proj_bias", {3 * d}, true);
  add_linear(h, p + ".out_proj", d, d, true);
}

// state-dict of lib/sttran.py:316-372 + lib/transformer.py:116-127
void declare_weights(SttranHandle* h) {
  const SttranConfig& c = h->cfg;
  const int64_t D = c.embed_dim, F = c.ffn_dim, FD = c.feat_dim, NC = c.num_obj_classes;
  const bool oc = c.mode != STTRAN_MODE_PREDCLS;
  add(h, "object_classifier.obj_embed.weight", {NC - 1, 200}, oc);
  add_bn(h, "object_classifier.pos_embed.0", 4, oc);
  add_linear(h, "object_classifier.pos_embed.1", 128, 4, oc, false);            // read by objcls_prep_kernel, dense
  add_linear(h, "object_classifier.decoder_lin.0", 1024, FD + 200 + 128, oc);
  add_bn(h, "object_classifier.decoder_lin.1", 1024, oc);
  add_linear(h, "object_classifier.decoder_lin.3", NC, 1024, oc);
  add(h, "union_func1.weight", {256, FD, 1, 1}, true);
  add(h, "union_func1.bias", {256}, true);
  add(h, "conv.0.weight", {128, 2, 7, 7}, true);
  add(h, "conv.0.bias", {128}, true);
  add_bn(h, "conv.2", 128, true);
  add(h, "conv.4.weight", {256, 128, 3, 3}, true);
  add(h, "conv.4.bias", {256}, true);
  add_bn(h, "conv.6", 256, true);
  add_linear(h, "subj_fc", 512, FD, true);
  add_linear(h, "obj_fc", 512, FD, true);
  add_linear(h, "vr_fc", 512, 256 * 49, true);
  add(h, "obj_embed.weight", {NC, 200}, true);
  add(h, "obj_embed2.weight", {NC, 200}, true);
  if (c.model == STTRAN_MODEL_DSG_DETR) {
    // lib/dsg_detr.py:497-506: sinusoid table + stock encoder layers (1 spatial, 3 temporal)
    add(h, "positional_encoder.pe", {1, 400, D}, true);
    for (int i = 0; i < 4; ++i) {
      const std::string p = i == 0 ? std::string("local_transformer.layers.0")
                                   : "global_transformer.layers." + std::to_string(i - 1);
      add_mha(h, p + ".self_attn", D);
      add_linear(h, p + ".linear1", F, D, true);
      add_linear(h, p + ".linear2", D, F, true);
      add(h, p + ".norm1.weight", {D}, true); add(h, p + ".norm1.bias", {D}, true);
      add(h, p + ".norm2.weight", {D}, true); add(h, p + ".norm2.bias", {D}, true);
    }
  }
  for (int i = 0; i < (c.model == STTRAN_MODEL_DSG_DETR ? 0 : c.enc_layers); ++i) {
    const std::string p = "glocal_transformer.local_attention.layers." + std::to_string(i);
    add_mha(h, p + ".self_attn", D);
    add_linear(h, p + ".linear1", F, D, true);
    add_linear(h, p + ".linear2", D, F, true);
    add(h, p + ".norm1.weight", {D}, true); add(h, p + ".norm1.bias", {D}, true);
    add(h, p + ".norm2.weight", {D}, true); add(h, p + ".norm2.bias", {D}, true);
  }
  for (int i = 0; i < (c.model == STTRAN_MODEL_DSG_DETR ? 0 : c.dec_layers); ++i) {
    const std::string p = "glocal_transformer.global_attention.layers." + std::to_string(i);
    add_mha(h, p + ".multihead2", D);
    add_linear(h, p + ".linear1", F, D, true);
    add_linear(h, p + ".linear2", D, F, true);
    add(h, p + ".norm3.weight", {D}, true); add(h, p + ".norm3.bias", {D}, true);
  }
  if (c.model != STTRAN_MODEL_DSG_DETR) add(h, "glocal_transformer.position_embedding.weight", {2, D}, true);
  add_linear(h, "a_rel_compress", c.attention_classes, D, true, false);         // packed (and padded) into heads_w
  add_linear(h, "s_rel_compress", c.spatial_classes, D, true, false);
  add_linear(h, "c_rel_compress", c.contact_classes, D, true, false);
}


}  // namespace sttran_host

extern "C" {

const char* sttran_version(void) { return "sttran-hip 0.1.0 (gfx950, fp32 MFMA)"; }

const char* sttran_last_error(SttranHandle* h) { return h ? h->err.c_str() : "null handle"; }

int sttran_create(const SttranConfig* cfg, SttranHandle** out) {
  if (!cfg || !out || cfg->struct_size != sizeof(SttranConfig)) return STTRAN_ERR_INVALID;
  if (cfg->embed_dim % 4 || cfg->nhead <= 0 || cfg->embed_dim % cfg->nhead || cfg->feat_dim % 32 ||
      cfg->ffn_dim % 4 || cfg->enc_layers < 0 || cfg->dec_layers < 0 || cfg->num_obj_classes < 2 ||
      cfg->num_obj_classes > 64 || cfg->embed_dim != 1536 + 400 ||
      cfg->attention_classes + cfg->spatial_classes + cfg->contact_classes > 64)
    return STTRAN_ERR_INVALID;
  if (cfg->model != STTRAN_MODEL_STTRAN && cfg->model != STTRAN_MODEL_DSG_DETR) return STTRAN_ERR_INVALID;
  // DSG-DETR: only the sgdet branch of the reference runs (lib/dsg_detr.py predcls feeds 2376-d features
  // into Linear(2048, 512))
  if (cfg->model == STTRAN_MODEL_DSG_DETR && cfg->mode != STTRAN_MODE_SGDET) return STTRAN_ERR_INVALID;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev)
    return STTRAN_ERR_HIP;
  if (hipSetDevice(cfg->device) != hipSuccess) return STTRAN_ERR_HIP;
  SttranHandle* h = new SttranHandle();
  h->cfg = *cfg;
  declare_weights(h);
  Lane* l0 = nullptr;
  if (lane_create(h, &l0) != STTRAN_OK) {
    delete h;
    return STTRAN_ERR_HIP;
  }
  h->lanes.push_back(l0);
  h->L = l0;
  *out = h;
  return STTRAN_OK;
}

void sttran_destroy(SttranHandle* h) {
  if (!h) return;
  hipSetDevice(h->cfg.device);
  hipDeviceSynchronize();
  for (auto& kv : h->w) {
    weight_free(kv.second);
    if (kv.second.planes) hipFree(kv.second.planes);
    if (kv.second.planes_fm) hipFree(kv.second.planes_fm);
  }
  if (h->w4_planes) hipFree(h->w4_planes);
  if (h->w4_planes_fm) hipFree(h->w4_planes_fm);
  if (h->fc_planes) hipFree(h->fc_planes);
  h->derived.release();
  for (Lane* L : h->lanes) lane_destroy(L);
  for (auto& e : h->prof_ev) { hipEventDestroy(e.a); hipEventDestroy(e.b); }
  delete h;
}

int sttran_load_tensor(SttranHandle* h, const char* key, const void* data, const int64_t* shape, int32_t ndim,
                       int32_t dtype, int32_t on_device) {
  if (!h || !key || !data || (ndim > 0 && !shape) || ndim < 0) return fail(h, STTRAN_ERR_INVALID, "load_tensor: bad argument");
  auto it = h->w.find(key);
  if (it == h->w.end()) return STTRAN_OK;   // strict=False: unknown keys (num_batches_tracked, ...) are ignored
  Tensor& t = it->second;
  if (dtype != STTRAN_DTYPE_F32) return fail(h, STTRAN_ERR_INVALID, std::string(key) + ": expected float32");
  size_t n = 1;
  for (int i = 0; i < ndim; ++i) n *= (size_t)shape[i];
  bool same = (size_t)ndim == t.shape.size();
  for (int i = 0; same && i < ndim; ++i) same = shape[i] == t.shape[i];
  if (!same || n != t.n) return fail(h, STTRAN_ERR_INVALID, std::string(key) + ": shape mismatch");
  HIPCK(hipSetDevice(h->cfg.device));
  const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  if (t.ld) {
    // GEMM weight: rows zero-padded to pad32(cols) (B_KMAJOR_PAD contract, csrc/gemm_f32_mfma.h)
    const size_t rows = (size_t)t.shape[0], cols = (size_t)t.shape[1], bytes = rows * (size_t)t.ld * 4 + 256;
    if (!t.d) {
      HIPCK(weight_alloc(t, bytes));
      HIPCK(hipMemset(t.d, 0, bytes));
    }
    HIPCK(hipMemcpy2D(t.d, (size_t)t.ld * 4, data, cols * 4, cols * 4, rows, kind));
  } else {
    // 256 zeroed bytes of slack: position_embedding.weight is a GEMM A operand (read up to pad32(K) per row)
    if (!t.d) {
      HIPCK(weight_alloc(t, t.n * 4 + 256));
      HIPCK(hipMemset(t.d, 0, t.n * 4 + 256));
    }
    HIPCK(hipMemcpy(t.d, data, t.n * 4, kind));
  }
  t.loaded = true;
  h->finalized = false;
  h->planes_ready = false;
  return STTRAN_OK;
}

int sttran_missing_keys(SttranHandle* h, char* buf, int64_t buflen) {
  if (!h) return -1;
  int n = 0;
  std::string s;
  for (auto& kv : h->w)
    if (kv.second.required && !kv.second.loaded) { ++n; s += kv.first; s += '\n'; }
  if (buf && buflen > 0) {
    strncpy(buf, s.c_str(), (size_t)buflen - 1);
    buf[buflen - 1] = 0;
  }
  return n;
}

int sttran_finalize_weights(SttranHandle* h) {
  if (!h) return STTRAN_ERR_INVALID;
  if (h->finalized) return STTRAN_OK;
  for (auto& kv : h->w)
    if (kv.second.required && !kv.second.loaded) return fail(h, STTRAN_ERR_WEIGHTS, "missing weight: " + kv.first);
  HIPCK(hipSetDevice(h->cfg.device));
  const SttranConfig& c = h->cfg;
  const int64_t D = c.embed_dim;
  const int nh = c.attention_classes + c.spatial_classes + c.contact_classes;
  const bool oc = c.mode != STTRAN_MODE_PREDCLS;
  // arena layout (floats)
  size_t total = 2 * 128 + 2 * 256 + 128 * 104 + 256 * 1152 + (size_t)nh * pad32(D) + 128 + (size_t)c.dec_layers * 4 * D + 2 * 4 + 2 * 1024 + 64 +
                 (size_t)1024 * pad32(c.feat_dim) + 1024 + 64;
  HIPCK(h->derived.ensure(total * 4));
  float* p = h->derived.as<float>();
  auto take = [&](size_t n) { float* r = p; p += (n + 3) & ~size_t(3); return r; };
  h->bn1_scale = take(128); h->bn1_shift = take(128);
  h->bn2_scale = take(256); h->bn2_shift = take(256);
  h->heads_w = take((size_t)nh * pad32(D) + 64); h->heads_b = take(64);
  h->w0_perm = take(128 * 104);
  h->w4_perm = take(256 * 1152);
  h->dec.resize(c.dec_layers);
  for (int i = 0; i < c.dec_layers; ++i) h->dec[i].posbias = take(4 * D);
  h->oc_pos_scale = take(4); h->oc_pos_shift = take(4);
  h->oc_bn_scale = take(1024); h->oc_bn_shift = take(1024);
  h->fc_w = take((size_t)1024 * pad32(c.feat_dim) + 64); h->fc_b = take(1024);
  // subj_fc and obj_fc (lib/sttran.py:346-347, 390-391) as ONE grouped GEMM: stacked weight rows / biases; the two column
  // groups gather their A rows through two tables (GemmOperand::aux)
  {
    const size_t wb = (size_t)512 * pad32(c.feat_dim) * 4;
    HIPCK(hipMemcpy(h->fc_w, W(h, "subj_fc.weight"), wb, hipMemcpyDeviceToDevice));
    HIPCK(hipMemcpy(h->fc_w + (size_t)512 * pad32(c.feat_dim), W(h, "obj_fc.weight"), wb, hipMemcpyDeviceToDevice));
    HIPCK(hipMemcpy(h->fc_b, W(h, "subj_fc.bias"), 512 * 4, hipMemcpyDeviceToDevice));
    HIPCK(hipMemcpy(h->fc_b + 512, W(h, "obj_fc.bias"), 512 * 4, hipMemcpyDeviceToDevice));
  }

  // eval-mode BatchNorm -> per-channel scale/shift: y = x*s + t, s = g/sqrt(var+eps), t = b - mean*s
  auto bn = [&](const std::string& pre, int n, float* ds, float* dt) -> int {
    std::vector<float> g(n), b(n), m(n), v(n), s(n), t(n);
    HIPCK(hipMemcpy(g.data(), W(h, pre + ".weight"), n * 4, hipMemcpyDeviceToHost));
    HIPCK(hipMemcpy(b.data(), W(h, pre + ".bias"), n * 4, hipMemcpyDeviceToHost));
    HIPCK(hipMemcpy(m.data(), W(h, pre + ".running_mean"), n * 4, hipMemcpyDeviceToHost));
    HIPCK(hipMemcpy(v.data(), W(h, pre + ".running_var"), n * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) {
      const double sd = (double)g[i] / std::sqrt((double)v[i] + 1e-5);
      s[i] = (float)sd;
      t[i] = (float)((double)b[i] - (double)m[i] * sd);
    }
    HIPCK(hipMemcpy(ds, s.data(), n * 4, hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(dt, t.data(), n * 4, hipMemcpyHostToDevice));
    return STTRAN_OK;
  };
  int rc;
  if ((rc = bn("conv.2", 128, h->bn1_scale, h->bn1_shift))) return rc;
  if ((rc = bn("conv.6", 256, h->bn2_scale, h->bn2_shift))) return rc;
  if (oc) {
    if ((rc = bn("object_classifier.pos_embed.0", 4, h->oc_pos_scale, h->oc_pos_shift))) return rc;
    if ((rc = bn("object_classifier.decoder_lin.1", 1024, h->oc_bn_scale, h->oc_bn_shift))) return rc;
  }
  // conv.0.weight [128][ci 2][tap 49] -> [128][group 13][ci 2][tap-in-group 4]: the K order of mask_conv1_pool_kernel,
  // whose lane halves carry the two input channels.  Tap 49 is the BIAS tap (the kernel feeds it the operand 1.0): weight
  // (conv.0.bias[co], 0) for the two halves; taps 50, 51 stay zero and are not executed.
  {
    std::vector<float> w(128 * 98), wp(128 * 104, 0.f), b0(128);
    HIPCK(hipMemcpy(w.data(), W(h, "conv.0.weight"), w.size() * 4, hipMemcpyDeviceToHost));
    HIPCK(hipMemcpy(b0.data(), W(h, "conv.0.bias"), b0.size() * 4, hipMemcpyDeviceToHost));
    for (int co = 0; co < 128; ++co) {
      for (int ci = 0; ci < 2; ++ci)
        for (int t = 0; t < 49; ++t) wp[(size_t)co * 104 + (t / 4) * 8 + ci * 4 + (t % 4)] = w[(size_t)co * 98 + ci * 49 + t];
      wp[(size_t)co * 104 + (49 / 4) * 8 + 0 * 4 + (49 % 4)] = b0[co];
    }
    HIPCK(hipMemcpy(h->w0_perm, wp.data(), wp.size() * 4, hipMemcpyHostToDevice));
  }
  // conv.4.weight [256][ci 128][ky 3][kx 3] -> [256][(ky, kx)][ci]: the K order of the B_CONV2 loader
  {
    std::vector<float> w(256 * 1152), wp(256 * 1152);
    HIPCK(hipMemcpy(w.data(), W(h, "conv.4.weight"), w.size() * 4, hipMemcpyDeviceToHost));
    for (int co = 0; co < 256; ++co)
      for (int ci = 0; ci < 128; ++ci)
        for (int t = 0; t < 9; ++t) wp[(size_t)co * 1152 + t * 128 + ci] = w[(size_t)co * 1152 + ci * 9 + t];
    HIPCK(hipMemcpy(h->w4_perm, wp.data(), wp.size() * 4, hipMemcpyHostToDevice));
  }
  // packed relation heads [a | s | c] (lib/sttran.py:370-372)
  {
    size_t ro = 0;
    const std::pair<const char*, int> hs[3] = {{"a_rel_compress", c.attention_classes},
                                               {"s_rel_compress", c.spatial_classes},
                                               {"c_rel_compress", c.contact_classes}};
    for (auto& kv : hs) {
      HIPCK(hipMemcpy2D(h->heads_w + ro * pad32(D), (size_t)pad32(D) * 4, W(h, std::string(kv.first) + ".weight"),
                        (size_t)D * 4, (size_t)D * 4, (size_t)kv.second, hipMemcpyDeviceToDevice));
      HIPCK(hipMemcpy(h->heads_b + ro, W(h, std::string(kv.first) + ".bias"), (size_t)kv.second * 4,
                      hipMemcpyDeviceToDevice));
      ro += kv.second;
    }
  }
  // position embedding folded into a per-slot bias of the q/k projections:
  //   (g + pos) Wqk^T + b = g Wqk^T + (pos Wqk^T) + b      (lib/transformer.py:51, pos is one of 2 rows)
  for (int i = 0; i < (c.model == STTRAN_MODEL_DSG_DETR ? 0 : c.dec_layers); ++i) {
    const std::string pre = "glocal_transformer.global_attention.layers." + std::to_string(i) + ".multihead2";
    GemmOperand A{W(h, "glocal_transformer.position_embedding.weight"), D, nullptr, 0};
    EpiLinear e = epi_plain(h->dec[i].posbias, 2 * D, nullptr);
    if ((rc = run_linear(h, nullptr, A, W(h, pre + ".in_proj_weight"), 2, (int)(2 * D), (int)D, e))) return rc;
  }
  HIPCK(hipDeviceSynchronize());
  h->finalized = true;
  return STTRAN_OK;
}

int sttran_set_gemm_engine(SttranHandle* h, int32_t engine) {
  if (!h || (engine != STTRAN_GEMM_FP32_MFMA && engine != STTRAN_GEMM_BF16X3 && engine != STTRAN_GEMM_BF16X3_ALL)) return STTRAN_ERR_INVALID;
  h->gemm_engine = engine;
  return STTRAN_OK;
}


}  // extern "C"
