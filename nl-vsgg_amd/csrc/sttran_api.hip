// sttran_api.hip -- C ABI (include/sttran_hip.h) and host orchestration of STTran.forward
// (lib/sttran.py:375-411 -> lib/transformer.py:130-187 with the empty-frame handling of
// lib/transformer_wk.py:144-195).  Host work per call: O(P) integer index maps; everything else is
// enqueued on the caller's stream.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/sttran_hip.h"
#include "../../include/sttran_hip_debug.h"
#include "kernels.h"

using namespace sttran;

namespace {

// columns of a GEMM operand row that must be readable: the next multiple of the K-step (32)
inline int64_t pad32(int64_t k) { return (k + 31) / 32 * 32; }

struct Tensor {
  float* d = nullptr;
  void* d_guard = nullptr;      // STTRAN_GUARD_WORKSPACE: cookie of the guarded allocation behind `d`
  void* planes = nullptr;  // bf16x3 engine: [3][rows][ld] bf16 planes of a GEMM weight (made on demand)
  std::vector<int64_t> shape;
  size_t n = 0;
  int64_t ld = 0;          // != 0: a [rows, cols] GEMM weight stored with this row stride (cols zero-padded to pad32)
  bool required = false, loaded = false;
};

// STTRAN_GUARD_WORKSPACE=1 (tests): every workspace buffer ENDS at the end of its mapping (sttran_debug_guarded_alloc), so
// a kernel that runs past one faults instead of reading its neighbour
static bool guard_workspace() {
  static const bool on = getenv("STTRAN_GUARD_WORKSPACE") && atoi(getenv("STTRAN_GUARD_WORKSPACE")) != 0;
  return on;
}
struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  void* guard_cookie = nullptr;
  hipError_t ensure(size_t need) {
    if (need <= bytes) return hipSuccess;
    if (p) { hipError_t e = drop(); if (e != hipSuccess) return e; }
    // zero-initialised, 256 bytes of slack: the pad columns of the activation rows (row stride pad32(D)) must be zero
    // and stay zero -- the GEMM A loader reads them for the K tail -- and clamped loads may touch the slack
    need = ((need + 255) & ~size_t(255)) + 256;
    if (guard_workspace()) {
      if (sttran_debug_guarded_alloc(need, &p, &guard_cookie) != STTRAN_OK) { p = nullptr; return hipErrorOutOfMemory; }
    } else {
      hipError_t e = hipMalloc(&p, need);
      if (e != hipSuccess) return e;
    }
    bytes = need;
    // hipMemset of device memory is asynchronous (it runs in the NULL stream) and a lane's own stream is non-blocking:
    // nothing orders it against the kernels the caller is about to enqueue there -- wait for it here (growth is rare)
    hipError_t e = hipMemset(p, 0, need);
    return e != hipSuccess ? e : hipDeviceSynchronize();
  }
  hipError_t drop() {
    hipError_t e = hipSuccess;
    if (guard_cookie) sttran_debug_guarded_free(guard_cookie);
    else if (p) e = hipFree(p);
    p = nullptr; bytes = 0; guard_cookie = nullptr;
    return e;
  }
  void release() { (void)drop(); }
  template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// weight tensors: hipMalloc, or a guarded allocation under STTRAN_GUARD_WORKSPACE
static hipError_t weight_alloc(Tensor& t, size_t bytes) {
  if (guard_workspace()) {
    void* p = nullptr;
    if (sttran_debug_guarded_alloc(bytes, &p, &t.d_guard) != STTRAN_OK) return hipErrorOutOfMemory;
    t.d = static_cast<float*>(p);
    return hipSuccess;
  }
  return hipMalloc(reinterpret_cast<void**>(&t.d), bytes);
}
static void weight_free(Tensor& t) {
  if (t.d_guard) sttran_debug_guarded_free(t.d_guard);
  else if (t.d) hipFree(t.d);
  t.d = nullptr; t.d_guard = nullptr;
}

struct ProfEvent { hipEvent_t a, b; int cls; int entry; };
struct ProfKey {
  std::string kernel; int cls; int64_t M, N, K;
  bool operator<(const ProfKey& o) const {
    return std::tie(cls, kernel, M, N, K) < std::tie(o.cls, o.kernel, o.M, o.N, o.K);
  }
};

// worst case of build_layout(): 2P (encoder off/len) + 6P (window off/len/q_begin) + 2P (dec_src) +
// P (out_src) + P (need_idx) + 2P (tok0/tok1) + P/2 (slots) int32 words
constexpr int64_t kIdxIntsPerPair = 18;

struct DecLayer { float* posbias = nullptr; };   // [2][2*D]

}  // namespace

// Everything ONE forward in flight needs for itself: workspace, stream-K park space, index-map / chunk-table staging and
// their caches, the device-side error flag.  A handle owns one lane (the classic `sttran_forward` on the caller's stream)
// or several (`sttran_set_lanes` + `sttran_forward_lane`: each lane runs on its OWN stream, forked from the caller's with
// an event, so consecutive one-clip calls -- the reference's loop, tools/test_STTran.py:81-84 -- overlap on the device).
// The weights and derived parameters are shared (read-only during forwards).
struct Lane {
  int64_t capP = 0, capB = 0;
  DevBuf x0, qkv, att, ybuf, hbuf, f1, gbuf, uni, vbuf, c2, slab, idx, zbuf, hobj, ebuf;
  DevBuf dsg;                   // DSG-DETR: class-sequence tables built on the device (launch_dsg_layout)
  DevBuf ctab, poff;            // chunk table of the call's inputs (kernels.h ChunkTable); per-pair element offsets [4 P] int64
  std::vector<int64_t> ctab_host;   // what ctab holds (re-uploaded only when a call's pointers / sizes differ)
  int* err_flag = nullptr;
  // index-map staging (pinned) + cache of the last layout
  static constexpr int kStages = 4;
  int32_t* stage[kStages] = {nullptr, nullptr, nullptr, nullptr};
  size_t stage_cap[kStages] = {0, 0, 0, 0};
  hipEvent_t stage_ev[kStages] = {nullptr, nullptr, nullptr, nullptr};
  int stage_next = 0;
  std::vector<int32_t> cached_counts, cached_clips;
  int64_t cached_P = -1;
  // layout of the current index buffer
  struct Layout {
    int n_enc_seq = 0, max_enc = 0, n_dec_seq = 0, max_dec = 0;
    int64_t n_dec_tok = 0, n_need = 0;
    size_t o_enc_off = 0, o_enc_len = 0, o_dec_off = 0, o_dec_len = 0, o_dec_src = 0, o_out_src = 0, o_slot = 0;
    size_t o_need = 0, o_qbegin = 0, o_tok0 = 0, o_tok1 = 0;
    size_t total_ints = 0;
    size_t o_clip_start = 0;      // DSG-DETR device layout: pair range of every clip [num_clips + 1]
    int num_clips = 0;
    bool dsg_device = false;      // the class sequences of this layout are built on the device
  } lay;
  int32_t* im_host = nullptr;   // pinned scratch for the im_idx read-back
  size_t im_host_cap = 0;
  // ordering: `own` = the lane's stream (sttran_forward_lane), fork_ev = recorded on the caller's stream when a lane call
  // starts, done_ev = recorded behind the last kernel of every forward on the stream it ran on (`last`): a later forward
  // of this lane on ANOTHER stream first waits for it (the cached uploads and the workspace belong to the earlier one)
  hipStream_t own = nullptr, last = nullptr;
  hipEvent_t fork_ev = nullptr, done_ev = nullptr;
  bool used = false;
};

struct SttranHandle {
  SttranConfig cfg{};
  std::string err;
  std::map<std::string, Tensor> w;
  bool finalized = false;
  int gemm_engine = STTRAN_GEMM_FP32_MFMA;
  bool planes_ready = false;
  // derived parameters
  DevBuf derived;               // one arena for all derived tensors
  float *bn1_scale = nullptr, *bn1_shift = nullptr, *bn2_scale = nullptr, *bn2_shift = nullptr;
  float *heads_w = nullptr, *heads_b = nullptr, *w0_perm = nullptr, *w4_perm = nullptr;
  float *fc_w = nullptr, *fc_b = nullptr;   // [subj_fc ; obj_fc] stacked: weights [1024, feat_dim], bias [1024] (one grouped launch)
  void* w4_planes = nullptr;    // bf16x3 engine: [3][256][1152] bf16 planes of w4_perm (made on demand)
  void* fc_planes = nullptr;    // ... [3][1024][feat_dim] planes of the stacked subj_fc | obj_fc weight
  float *oc_pos_scale = nullptr, *oc_pos_shift = nullptr, *oc_bn_scale = nullptr, *oc_bn_shift = nullptr;
  std::vector<DecLayer> dec;
  std::vector<Lane*> lanes;     // >= 1
  Lane* L = nullptr;            // the lane of the call in progress (calls on a handle are serialised by the caller)
  // profiling
  bool prof_on = false;
  std::vector<ProfEvent> prof_ev;
  std::map<ProfKey, int> prof_index;            // (kernel, shape) -> entry
  std::vector<SttranProfEntry> prof_entries;
  SttranProfile prof{};
  hipStream_t prof_stream = nullptr;
};

namespace {

int fail(SttranHandle* h, int code, const std::string& msg) {
  if (h) h->err = msg;
  return code;
}
#define HIPCK(expr)                                                                            \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess)                                                                      \
      return fail(h, STTRAN_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));       \
  } while (0)

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

const float* W(SttranHandle* h, const std::string& k) { return h->w[k].d; }

const char* tile_name(int tile) {
  switch (tile) {
    case TILE_256x128: return "256,128,4,2";
    case TILE_128x128: return "128,128,2,2";
    case TILE_128x64: return "128,64,2,2";
    case TILE_64x64: return "64,64,2,2";
    case TILE_128x176: return "128,176";
    case TILE_T128x128: return "128,128";
    default: return "?";
  }
}

struct ProfScope {
  SttranHandle* h; hipStream_t s; bool on;
  ProfEvent ev{};
  ProfScope(SttranHandle* h_, hipStream_t s_, int cls, double flops, double bytes, const std::string& kernel = std::string(),
            int64_t M = 0, int64_t N = 0, int64_t K = 0) : h(h_), s(s_), on(h_->prof_on) {
    if (!on) return;
    ev.cls = cls;
    ProfKey key{kernel, cls, M, N, K};
    auto it = h->prof_index.find(key);
    if (it == h->prof_index.end()) {
      SttranProfEntry e{};
      snprintf(e.kernel, sizeof(e.kernel), "%s", kernel.c_str());
      e.cls = cls; e.M = M; e.N = N; e.K = K;
      h->prof_entries.push_back(e);
      it = h->prof_index.emplace(key, (int)h->prof_entries.size() - 1).first;
    }
    ev.entry = it->second;
    h->prof_entries[ev.entry].launches += 1;
    h->prof_entries[ev.entry].flops += flops;
    hipEventCreate(&ev.a);
    hipEventCreate(&ev.b);
    hipEventRecord(ev.a, s);
    h->prof.flops[cls] += flops;
    h->prof.bytes[cls] += bytes;
    h->prof.launches[cls] += 1;
  }
  ~ProfScope() {
    if (!on) return;
    hipEventRecord(ev.b, s);
    h->prof_ev.push_back(ev);
  }
};

double gemm_flops(int64_t M, int64_t N, int64_t K) { return 2.0 * M * N * K; }
double gemm_bytes(int64_t M, int64_t N, int64_t K) { return 4.0 * (M * K + N * K + M * N); }

// C = act(A W^T + ...) through the planner; slab workspace grown on demand
int run_linear(SttranHandle* h, hipStream_t s, GemmOperand A, const float* Wt, int M, int N, int K, EpiLinear epi,
               int force_tile = 0, int force_split = 0) {
  if (M <= 0) return STTRAN_OK;
  GemmPlan plan = plan_gemm(M, N, K, force_tile, force_split);
  if (gemm_slab_bytes() > h->L->slab.bytes) {
    HIPCK(hipStreamSynchronize(s));
    HIPCK(h->L->slab.ensure(gemm_slab_bytes()));
  }
  if (h->gemm_engine != STTRAN_GEMM_FP32_MFMA && h->planes_ready && (M >= 512 || h->gemm_engine == STTRAN_GEMM_BF16X3_ALL) &&
      N >= 128 && !force_tile) {
    if (Wt == h->fc_w && h->fc_planes) {               // the grouped subj_fc | obj_fc launch (a derived tensor, not in h->w)
      const int64_t ldf = pad32(K);
      ProfScope ps(h, s, STTRAN_PROF_GEMM, gemm_flops(M, N, K), gemm_bytes(M, N, K),
                   "gemm_x3_kernel<X3Tile<256,128,4,2>,EpiLinear>", M, N, K);
      HIPCK(gemm_linear_x3(s, A, h->fc_planes, ldf, (int64_t)1024 * ldf, M, N, K, epi, h->L->slab.as<float>()));
      return STTRAN_OK;
    }
    // the weight (or a row range of it: the last decoder layer projects k|v and q separately) as bf16 planes
    for (auto& kv : h->w) {
      const Tensor& t = kv.second;
      if (!t.planes || !t.ld || t.ld != pad32(K)) continue;
      const int64_t rows = t.shape[0];
      if (Wt < t.d || Wt >= t.d + rows * t.ld) continue;
      const int64_t r0 = (Wt - t.d) / t.ld;
      if ((Wt - t.d) % t.ld || r0 + N > rows) break;
      ProfScope ps(h, s, STTRAN_PROF_GEMM, gemm_flops(M, N, K), gemm_bytes(M, N, K),
                   "gemm_x3_kernel<X3Tile<256,128,4,2>,EpiLinear>", M, N, K);
      HIPCK(gemm_linear_x3(s, A, reinterpret_cast<const uint16_t*>(t.planes) + r0 * t.ld, t.ld, rows * t.ld, M, N, K, epi,
                           h->L->slab.as<float>()));
      return STTRAN_OK;
    }
  }
  GemmOperand B{Wt, pad32(K), nullptr, 0};               // every weight that comes here is stored padded (Tensor::ld)
  const int tile = gemm_effective_tile(A, B, N, K, epi, plan, 1);        // the label names the kernel that really runs
  const bool t16 = tile == TILE_128x176 || tile == TILE_T128x128;
  ProfScope ps(h, s, STTRAN_PROF_GEMM, gemm_flops(M, N, K), gemm_bytes(M, N, K),
               t16 ? std::string("gemm16_kernel<Tile16<") + tile_name(tile) + ">,EpiLinear>"
                   : std::string("gemm_sk_kernel<GemmTile<") + tile_name(tile) + ",B_KMAJOR_PAD>,EpiLinear>", M, N, K);
  HIPCK(gemm_linear(s, A, B, M, N, K, epi, plan, h->L->slab.as<float>(), 1));
  return STTRAN_OK;
}

EpiLinear epi_plain(float* C, int64_t ldc, const float* bias, int relu = 0) {
  EpiLinear e{};
  e.C = C; e.ldc = ldc; e.bias = bias; e.relu = relu;
  return e;
}

// One post-norm encoder layer over ragged sequences (lib/transformer.py:20-30; also the stock
// nn.TransformerEncoderLayer of lib/dsg_detr.py:502-506 -- same sub-module names):
//   h = LN1(x + MHA(x,x,x));  out = LN2(h + W2 relu(W1 h + b1) + b2)
// len_on_device: `maxlen` is only an upper bound of the sequence lengths (they were computed on the device)
int run_encoder_layer(SttranHandle* h, hipStream_t s, const std::string& p, const float* xin, float* xout, int M,
                      const int* seq_off, const int* seq_len, int nseq, int maxlen, bool len_on_device = false) {
  const SttranConfig& c = h->cfg;
  const int D = c.embed_dim, F = c.ffn_dim;
  const int64_t LD = pad32(D), LF = pad32(F);     // row strides of the [*, D] / [*, F] workspace buffers (xin / xout included)
  float* QKV = h->L->qkv.as<float>(); float* ATT = h->L->att.as<float>(); float* Y = h->L->ybuf.as<float>();
  float* H = h->L->hbuf.as<float>(); float* F1 = h->L->f1.as<float>();
  int rc;
  if ((rc = run_linear(h, s, GemmOperand{xin, LD, nullptr}, W(h, p + ".self_attn.in_proj_weight"), M, 3 * D, D,
                       epi_plain(QKV, 3 * D, W(h, p + ".self_attn.in_proj_bias"))))) return rc;
  {
    ProfScope ps(h, s, STTRAN_PROF_ATTENTION, 4.0 * M * maxlen * D, 4.0 * M * 4 * D, "attention", M, maxlen, D);
    if (len_on_device) HIPCK(launch_attention_classes(s, QKV, seq_off, seq_len, nseq, maxlen, ATT, LD, D, c.nhead));
    else HIPCK(launch_attention(s, QKV, seq_off, seq_len, nullptr, nseq, maxlen, ATT, LD, D, c.nhead));
  }
  EpiLinear eo = epi_plain(Y, LD, W(h, p + ".self_attn.out_proj.bias"));
  eo.res = xin; eo.ldres = LD;
  if ((rc = run_linear(h, s, GemmOperand{ATT, LD, nullptr}, W(h, p + ".self_attn.out_proj.weight"), M, D, D, eo))) return rc;
  {
    ProfScope ps(h, s, STTRAN_PROF_LAYERNORM, 0, 8.0 * M * D, "layernorm_kernel", M, D, 0);
    HIPCK(launch_layernorm(s, Y, LD, W(h, p + ".norm1.weight"), W(h, p + ".norm1.bias"), H, LD, M, D));
  }
  if ((rc = run_linear(h, s, GemmOperand{H, LD, nullptr}, W(h, p + ".linear1.weight"), M, F, D,
                       epi_plain(F1, LF, W(h, p + ".linear1.bias"), 1)))) return rc;
  EpiLinear e2 = epi_plain(Y, LD, W(h, p + ".linear2.bias"));
  e2.res = H; e2.ldres = LD;
  if ((rc = run_linear(h, s, GemmOperand{F1, LF, nullptr}, W(h, p + ".linear2.weight"), M, D, F, e2))) return rc;
  {
    ProfScope ps(h, s, STTRAN_PROF_LAYERNORM, 0, 8.0 * M * D, "layernorm_kernel", M, D, 0);
    HIPCK(launch_layernorm(s, Y, LD, W(h, p + ".norm2.weight"), W(h, p + ".norm2.bias"), xout, LD, M, D));
  }
  return STTRAN_OK;
}

int ensure_workspace(SttranHandle* h, int64_t P, int64_t B) {
  if (P <= h->L->capP && B <= h->L->capB) return STTRAN_OK;
  HIPCK(hipDeviceSynchronize());
  const int64_t cp = std::max(P, h->L->capP), cb = std::max(B, h->L->capB);
  // Every [rows, D] activation buffer has a row stride of LD = pad32(D) floats (1952 for D = 1936): rows start on
  // 128-byte lines, and the 16 pad columns -- zeroed here, never written by any kernel -- are what the GEMM A loader
  // reads for the K tail (B_KMAJOR_PAD), so nothing a previous call left behind can reach a later call's result.
  const int64_t D = h->cfg.embed_dim, LD = pad32(D), F = h->cfg.ffn_dim, tok = 2 * cp;
  HIPCK(h->L->slab.ensure(gemm_slab_bytes()));
  HIPCK(h->L->x0.ensure((size_t)cp * LD * 4));
  HIPCK(h->L->ebuf.ensure((size_t)cp * LD * 4));
  HIPCK(h->L->qkv.ensure((size_t)tok * 3 * D * 4));
  HIPCK(h->L->att.ensure((size_t)tok * LD * 4));
  HIPCK(h->L->ybuf.ensure((size_t)tok * LD * 4));
  HIPCK(h->L->hbuf.ensure((size_t)tok * LD * 4));
  HIPCK(h->L->f1.ensure((size_t)tok * pad32(F) * 4));
  HIPCK(h->L->gbuf.ensure((size_t)tok * LD * 4));
  HIPCK(h->L->uni.ensure((size_t)(cp + tok) * LD * 4));
  HIPCK(h->L->vbuf.ensure((size_t)cp * 256 * 49 * 4));
  HIPCK(h->L->c2.ensure((size_t)cp * 128 * 49 * 4));
  HIPCK(h->L->idx.ensure((size_t)(kIdxIntsPerPair * cp + 64) * 4 + 4096));
  HIPCK(h->L->poff.ensure((size_t)cp * 4 * 8));
  if (h->cfg.mode != STTRAN_MODE_PREDCLS) {
    HIPCK(h->L->zbuf.ensure((size_t)cb * pad32(h->cfg.feat_dim + 328) * 4));
    HIPCK(h->L->hobj.ensure((size_t)cb * 1024 * 4));
  }
  h->L->capP = cp;
  h->L->capB = cb;
  h->L->cached_P = -1;     // the index buffer may have been re-allocated (and zeroed): the cached layout is gone
  return STTRAN_OK;
}

// Small host tables (index maps, the chunk table) go to the device through a ring of pinned staging buffers: the
// copy is enqueue-only, and a slot is reused only after the copy that read it has completed.
int upload_staged(SttranHandle* h, hipStream_t s, const void* src, size_t bytes, void* dst) {
  const int k = h->L->stage_next;
  h->L->stage_next = (k + 1) % Lane::kStages;
  if (h->L->stage_ev[k]) HIPCK(hipEventSynchronize(h->L->stage_ev[k]));
  else HIPCK(hipEventCreateWithFlags(&h->L->stage_ev[k], hipEventDisableTiming));
  if (h->L->stage_cap[k] < bytes) {
    if (h->L->stage[k]) HIPCK(hipHostFree(h->L->stage[k]));
    HIPCK(hipHostMalloc(reinterpret_cast<void**>(&h->L->stage[k]), bytes + 4096));
    h->L->stage_cap[k] = bytes + 4096;
  }
  memcpy(h->L->stage[k], src, bytes);
  HIPCK(hipMemcpyAsync(dst, h->L->stage[k], bytes, hipMemcpyHostToDevice, s));
  HIPCK(hipEventRecord(h->L->stage_ev[k], s));
  return STTRAN_OK;
}

// Build the index maps of one call on the host (O(P) integers).
//   enc sequences : non-empty frames (lib/transformer_wk.py:144-150)
//   dec sequences : 2-frame windows inside each clip, both-empty windows dropped (:175-185)
//   dec_src/slot  : window token -> encoder row / position-embedding row (lib/transformer.py:153-159)
//   out_src       : pair -> row of the unified [encoder rows | decoder rows] buffer, mode 'latter'
//                   (lib/transformer.py:179-185); clips with one frame keep the encoder row
//                   (lib/transformer_wk.py:187-188)
void build_layout(const std::vector<int32_t>& counts, const std::vector<int32_t>& clips, int64_t P,
                  std::vector<int32_t>& buf, Lane::Layout& L) {
  const int T = (int)counts.size();
  std::vector<int64_t> off(T + 1, 0);
  for (int t = 0; t < T; ++t) off[t + 1] = off[t] + counts[t];
  std::vector<int32_t> enc_off, enc_len, dec_off, dec_len, dec_src, out_src(P), need, qbegin, tok0(P, -1), tok1(P, -1);
  std::vector<uint8_t> slot;
  L = Lane::Layout();
  for (int t = 0; t < T; ++t)
    if (counts[t] > 0) { enc_off.push_back((int32_t)off[t]); enc_len.push_back(counts[t]); L.max_enc = std::max(L.max_enc, counts[t]); }
  for (int64_t p = 0; p < P; ++p) out_src[p] = (int32_t)p;
  int fs = 0;
  for (size_t c = 0; c < clips.size(); ++c) {
    const int fe = fs + clips[c];
    for (int j = fs; j + 1 < fe; ++j) {
      const int n0 = counts[j], n1 = counts[j + 1];
      if (n0 + n1 == 0) continue;
      const int32_t doff = (int32_t)dec_src.size();
      dec_off.push_back(doff);
      dec_len.push_back(n0 + n1);
      L.max_dec = std::max(L.max_dec, n0 + n1);
      for (int i = 0; i < n0 + n1; ++i) {
        // a pair appears as a slot-0 token in the window that starts at its frame and as a slot-1 token in
        // the window that ends at it: tok0 / tok1 let the first decoder layer project each pair ONCE
        (i < n0 ? tok0 : tok1)[off[j] + i] = (int32_t)dec_src.size();
        dec_src.push_back((int32_t)(off[j] + i));
        slot.push_back(i < n0 ? 0 : 1);
      }
      // rows of this window the 'latter' scatter reads (lib/transformer.py:179-185): the first window
      // of a clip gives both frames, every other window only its second frame.  Only those rows of the
      // LAST decoder layer are ever consumed, so that layer computes just them (need / q_begin).
      const int qb = (j == fs) ? 0 : n0;
      qbegin.push_back(qb);
      for (int i = qb; i < n0 + n1; ++i) {
        out_src[off[j] + i] = (int32_t)(P + need.size());
        need.push_back(doff + i);
      }
    }
    fs = fe;
  }
  L.n_enc_seq = (int)enc_off.size();
  L.n_dec_seq = (int)dec_off.size();
  L.n_dec_tok = (int64_t)dec_src.size();
  L.n_need = (int64_t)need.size();
  buf.clear();
  auto put = [&](const std::vector<int32_t>& v) { size_t o = buf.size(); buf.insert(buf.end(), v.begin(), v.end()); return o; };
  L.o_enc_off = put(enc_off); L.o_enc_len = put(enc_len);
  L.o_dec_off = put(dec_off); L.o_dec_len = put(dec_len);
  L.o_dec_src = put(dec_src); L.o_out_src = put(out_src);
  L.o_need = put(need); L.o_qbegin = put(qbegin);
  L.o_tok0 = put(tok0); L.o_tok1 = put(tok1);
  L.o_slot = buf.size();
  buf.resize(buf.size() + (slot.size() + 3) / 4, 0);
  if (!slot.empty()) memcpy(buf.data() + L.o_slot, slot.data(), slot.size());
  L.total_ints = buf.size();
}

// DSG-DETR index maps (lib/dsg_detr.py:536-555).  Spatial sequences = frames (as above).  Temporal
// sequences = one per object class present, its pairs in pair order.  PE rows are handed out by POSITION
// (lib/dsg_detr.py:551-554: `[0]*count_0 + [1]*count_1 + ...` over the sorted unique subject boxes): token i
// takes the dense rank of the i-th SMALLEST subject of its sequence -- its own subject's rank only when the
// subject numbers ascend along the sequence (boxes stored frame by frame; golden dsgdetr_shuffled_boxes is the
// other case).
// Stored in the STTran slots: dec_off/dec_len = class sequences, dec_src = pair of each token,
// need = PE row of each token, out_src = P + token of each pair.
void build_layout_dsg(const std::vector<int32_t>& counts, const std::vector<int32_t>& clips, int64_t P,
                      const int64_t* pair_idx, const int64_t* labels, std::vector<int32_t>& buf, Lane::Layout& L) {
  const int T = (int)counts.size();
  std::vector<int32_t> enc_off, enc_len, cls_off, cls_len, tok_pair, tok_pos, out_src(P);
  L = Lane::Layout();
  int64_t o = 0;
  for (int t = 0; t < T; ++t) {
    if (counts[t] > 0) { enc_off.push_back((int32_t)o); enc_len.push_back(counts[t]); L.max_enc = std::max(L.max_enc, counts[t]); }
    o += counts[t];
  }
  // one temporal sequence per (clip, object class): lib/dsg_detr.py:528-541 groups one clip's pairs by class
  std::vector<int32_t> clip_of_frame;
  for (size_t c = 0; c < clips.size(); ++c) clip_of_frame.insert(clip_of_frame.end(), (size_t)clips[c], (int32_t)c);
  std::map<std::pair<int32_t, int64_t>, std::vector<int32_t>> by_class;
  {
    int64_t p = 0;
    for (int t = 0; t < T; ++t)
      for (int i = 0; i < counts[t]; ++i, ++p)
        by_class[std::make_pair(clip_of_frame[t], labels[pair_idx[2 * p + 1]])].push_back((int32_t)p);
  }
  for (auto& kv : by_class) {
    const std::vector<int32_t>& pairs = kv.second;
    std::vector<int64_t> subj;
    for (int32_t p : pairs) subj.push_back(pair_idx[2 * (int64_t)p]);
    std::sort(subj.begin(), subj.end());
    std::vector<int64_t> uniq(subj);
    uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
    cls_off.push_back((int32_t)tok_pair.size());
    cls_len.push_back((int32_t)pairs.size());
    L.max_dec = std::max(L.max_dec, (int)pairs.size());
    for (size_t i = 0; i < pairs.size(); ++i) {
      out_src[pairs[i]] = (int32_t)(P + tok_pair.size());
      tok_pair.push_back(pairs[i]);
      tok_pos.push_back((int32_t)(std::lower_bound(uniq.begin(), uniq.end(), subj[i]) - uniq.begin()));
    }
  }
  L.n_enc_seq = (int)enc_off.size();
  L.n_dec_seq = (int)cls_off.size();
  L.n_dec_tok = P;
  L.n_need = P;
  buf.clear();
  auto put = [&](const std::vector<int32_t>& v) { size_t off = buf.size(); buf.insert(buf.end(), v.begin(), v.end()); return off; };
  L.o_enc_off = put(enc_off); L.o_enc_len = put(enc_len);
  L.o_dec_off = put(cls_off); L.o_dec_len = put(cls_len);
  L.o_dec_src = put(tok_pair); L.o_out_src = put(out_src);
  L.o_need = put(tok_pos); L.o_qbegin = buf.size();
  L.o_slot = buf.size();
  L.total_ints = buf.size();
}

// DSG-DETR, device form: only what the host knows goes through the index buffer -- the spatial sequences (frames) and the
// pair range of every clip; the class sequences are built by launch_dsg_layout from labels / pair_idx where they live.
// n_dec_seq = one slot per (clip, class), max_dec = the largest clip (an upper bound of every class sequence).
void build_layout_dsg_static(const std::vector<int32_t>& counts, const std::vector<int32_t>& clips, int64_t P, int NC,
                             std::vector<int32_t>& buf, Lane::Layout& L) {
  std::vector<int32_t> enc_off, enc_len, clip_start;
  L = Lane::Layout();
  int64_t o = 0;
  size_t t = 0;
  for (size_t c = 0; c < clips.size(); ++c) {
    clip_start.push_back((int32_t)o);
    const int64_t o0 = o;
    for (int f = 0; f < clips[c]; ++f, ++t) {
      if (counts[t] > 0) { enc_off.push_back((int32_t)o); enc_len.push_back(counts[t]); L.max_enc = std::max(L.max_enc, counts[t]); }
      o += counts[t];
    }
    L.max_dec = std::max<int>(L.max_dec, (int)(o - o0));
  }
  clip_start.push_back((int32_t)o);
  L.n_enc_seq = (int)enc_off.size();
  L.num_clips = (int)clips.size();
  L.n_dec_seq = L.num_clips * NC;
  L.n_dec_tok = P;
  L.n_need = P;
  L.dsg_device = true;
  buf.clear();
  auto put = [&](const std::vector<int32_t>& v) { size_t off = buf.size(); buf.insert(buf.end(), v.begin(), v.end()); return off; };
  L.o_enc_off = put(enc_off); L.o_enc_len = put(enc_len);
  L.o_clip_start = put(clip_start);
  L.total_ints = buf.size();
}

// ---- lanes ---------------------------------------------------------------------------------------
void lane_destroy(Lane* L);
int lane_create(SttranHandle* h, Lane** out) {
  Lane* L = new Lane();
  if (hipMalloc(reinterpret_cast<void**>(&L->err_flag), 64) != hipSuccess || hipMemset(L->err_flag, 0, 64) != hipSuccess ||
      hipStreamCreateWithFlags(&L->own, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&L->fork_ev, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&L->done_ev, hipEventDisableTiming) != hipSuccess) {
    lane_destroy(L);                                     // releases whatever was created before the failing call
    return fail(h, STTRAN_ERR_HIP, "lane: stream / event / flag allocation failed");
  }
  if (hipDeviceSynchronize() != hipSuccess) {            // the memset above ran in the NULL stream; L->own does not wait for it
    lane_destroy(L);
    return fail(h, STTRAN_ERR_HIP, "lane: synchronise failed");
  }
  *out = L;
  return STTRAN_OK;
}
void lane_destroy(Lane* L) {
  if (!L) return;
  for (DevBuf* b : {&L->x0, &L->qkv, &L->att, &L->ybuf, &L->hbuf, &L->f1, &L->gbuf, &L->uni, &L->vbuf, &L->c2, &L->slab, &L->idx,
                    &L->zbuf, &L->hobj, &L->ebuf, &L->dsg, &L->ctab, &L->poff})
    b->release();
  for (int i = 0; i < Lane::kStages; ++i) {
    if (L->stage[i]) hipHostFree(L->stage[i]);
    if (L->stage_ev[i]) hipEventDestroy(L->stage_ev[i]);
  }
  if (L->im_host) hipHostFree(L->im_host);
  if (L->err_flag) hipFree(L->err_flag);
  if (L->fork_ev) hipEventDestroy(L->fork_ev);
  if (L->done_ev) hipEventDestroy(L->done_ev);
  if (L->own) hipStreamDestroy(L->own);
  delete L;
}
bool capturing(hipStream_t s) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  return hipStreamIsCapturing(s, &st) == hipSuccess && st != hipStreamCaptureStatusNone;
}
int forward_on(SttranHandle* h, const SttranInputs* in_, const SttranOutputs* out, hipStream_t s);

}  // namespace

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
  }
  if (h->w4_planes) hipFree(h->w4_planes);
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

int sttran_reserve(SttranHandle* h, int64_t max_pairs, int64_t max_boxes) {
  if (!h || max_pairs < 0 || max_boxes < 0) return STTRAN_ERR_INVALID;
  HIPCK(hipSetDevice(h->cfg.device));
  int rc = STTRAN_OK;
  Lane* keep = h->L;
  for (Lane* L : h->lanes) {
    h->L = L;
    if ((rc = ensure_workspace(h, max_pairs, max_boxes))) break;
  }
  h->L = keep;
  return rc;
}

int sttran_set_lanes(SttranHandle* h, int32_t lanes) {
  if (!h || lanes < 1 || lanes > STTRAN_MAX_LANES) return fail(h, STTRAN_ERR_INVALID, "set_lanes: 1 .. STTRAN_MAX_LANES");
  HIPCK(hipSetDevice(h->cfg.device));
  HIPCK(hipDeviceSynchronize());                         // nothing of this handle is in flight while lanes come and go
  while ((int)h->lanes.size() > lanes) { lane_destroy(h->lanes.back()); h->lanes.pop_back(); }
  // the device is idle and every recorded event has completed: a profile_read after this must not synchronise a stream
  // that may just have been destroyed with its lane (the last forward's stream may have been a lane's own)
  h->prof_stream = nullptr;
  while ((int)h->lanes.size() < lanes) {
    Lane* L = nullptr;
    int rc = lane_create(h, &L);
    if (rc) return rc;
    h->lanes.push_back(L);
  }
  h->L = h->lanes[0];
  return STTRAN_OK;
}

int32_t sttran_num_lanes(SttranHandle* h) { return h ? (int32_t)h->lanes.size() : 0; }

int sttran_lane_stream(SttranHandle* h, int32_t lane, void** stream) {
  if (!h || !stream || lane < 0 || lane >= (int)h->lanes.size()) return STTRAN_ERR_INVALID;
  *stream = h->lanes[lane]->own;
  return STTRAN_OK;
}

int sttran_lane_join(SttranHandle* h, int32_t lane, void* stream_) {
  if (!h || lane < -1 || lane >= (int)h->lanes.size()) return STTRAN_ERR_INVALID;
  HIPCK(hipSetDevice(h->cfg.device));
  hipStream_t s = reinterpret_cast<hipStream_t>(stream_);
  for (int i = 0; i < (int)h->lanes.size(); ++i) {
    Lane* L = h->lanes[i];
    if ((lane >= 0 && i != lane) || !L->used || L->last == s) continue;
    HIPCK(hipStreamWaitEvent(s, L->done_ev, 0));
  }
  return STTRAN_OK;
}

int sttran_profile_enable(SttranHandle* h, int32_t enable) {
  if (!h) return STTRAN_ERR_INVALID;
  h->prof_on = enable != 0;
  return STTRAN_OK;
}

int sttran_profile_reset(SttranHandle* h) {
  if (!h) return STTRAN_ERR_INVALID;
  for (auto& e : h->prof_ev) { hipEventDestroy(e.a); hipEventDestroy(e.b); }
  h->prof_ev.clear();
  h->prof_index.clear();
  h->prof_entries.clear();
  memset(&h->prof, 0, sizeof(h->prof));
  return STTRAN_OK;
}

int sttran_profile_read(SttranHandle* h, SttranProfile* out) {
  if (!h || !out || out->struct_size != sizeof(SttranProfile)) return STTRAN_ERR_INVALID;
  if (h->prof_stream) HIPCK(hipStreamSynchronize(h->prof_stream));
  for (Lane* L : h->lanes)
    if (L->used) HIPCK(hipEventSynchronize(L->done_ev));
  for (auto& e : h->prof_ev) {
    float ms = 0.f;
    HIPCK(hipEventElapsedTime(&ms, e.a, e.b));
    h->prof.ms[e.cls] += ms;
    if (e.entry >= 0 && e.entry < (int)h->prof_entries.size()) h->prof_entries[e.entry].ms += ms;
    hipEventDestroy(e.a);
    hipEventDestroy(e.b);
  }
  h->prof_ev.clear();
  h->prof.struct_size = sizeof(SttranProfile);
  *out = h->prof;
  return STTRAN_OK;
}

int sttran_profile_entries(SttranHandle* h, SttranProfEntry* out, int32_t cap, int32_t* count) {
  if (!h || !count || cap < 0 || (cap > 0 && !out)) return STTRAN_ERR_INVALID;
  *count = (int32_t)h->prof_entries.size();
  for (int32_t i = 0; i < cap && i < *count; ++i) out[i] = h->prof_entries[i];
  return STTRAN_OK;
}

// run one forward of lane L on stream x, ordered behind the lane's previous forward if that ran on another stream
static int forward_ordered(SttranHandle* h, Lane* L, const SttranInputs* in, const SttranOutputs* out, hipStream_t x) {
  const bool cap = capturing(x);                   // a capture records kernels only; replays are ordered by their owner
  if (!cap && L->used && L->last != x) HIPCK(hipStreamWaitEvent(x, L->done_ev, 0));
  h->L = L;
  const int rc = forward_on(h, in, out, x);
  h->L = h->lanes[0];
  if (!cap) {
    // (also after a failed call: whatever it enqueued before failing still runs on x)
    HIPCK(hipEventRecord(L->done_ev, x));
    L->last = x; L->used = true;
  }
  return rc;
}

int sttran_forward(SttranHandle* h, const SttranInputs* in, const SttranOutputs* out, void* stream_) {
  if (!h) return STTRAN_ERR_INVALID;
  if (hipSetDevice(h->cfg.device) != hipSuccess) return fail(h, STTRAN_ERR_HIP, "hipSetDevice");
  return forward_ordered(h, h->lanes[0], in, out, reinterpret_cast<hipStream_t>(stream_));
}

int sttran_forward_lane(SttranHandle* h, int32_t lane, const SttranInputs* in, const SttranOutputs* out, void* stream_) {
  if (!h) return STTRAN_ERR_INVALID;
  if (lane < 0 || lane >= (int)h->lanes.size()) return fail(h, STTRAN_ERR_INVALID, "forward_lane: no such lane (sttran_set_lanes)");
  HIPCK(hipSetDevice(h->cfg.device));
  hipStream_t s = reinterpret_cast<hipStream_t>(stream_);
  Lane* L = h->lanes[lane];
  if (capturing(s)) return fail(h, STTRAN_ERR_INVALID, "forward_lane: the caller's stream is being captured (use sttran_forward)");
  // fork: everything the caller enqueued on `s` so far (the producer of this entry's tensors) precedes the lane's work
  HIPCK(hipEventRecord(L->fork_ev, s));
  HIPCK(hipStreamWaitEvent(L->own, L->fork_ev, 0));
  return forward_ordered(h, L, in, out, L->own);     // no join: sttran_lane_join / sttran_sync_check order a consumer
}

}  // extern "C"

namespace {

int forward_on(SttranHandle* h, const SttranInputs* in_, const SttranOutputs* out, hipStream_t s) {
  static_assert(offsetof(SttranInputs, clip_features) == STTRAN_INPUTS_V1_SIZE, "STTRAN_INPUTS_V1_SIZE");
  if (!in_ || !out || (in_->struct_size != sizeof(SttranInputs) && in_->struct_size != STTRAN_INPUTS_V1_SIZE) ||
      out->struct_size != sizeof(SttranOutputs))
    return fail(h, STTRAN_ERR_INVALID, "forward: bad struct_size");
  SttranInputs in_copy{};                        // a round-2 caller's struct has no pointer tables: they read as NULL
  memcpy(&in_copy, in_, in_->struct_size);
  const SttranInputs* in = &in_copy;
  const SttranConfig& c = h->cfg;
  const int64_t P = in->num_pairs, B = in->num_boxes;
  if (P <= 0 || B <= 0) return fail(h, STTRAN_ERR_EMPTY, "forward: entry has no pairs");
  if (P > (1 << 28) / 49 || B > (1 << 30)) return fail(h, STTRAN_ERR_LIMIT, "forward: too many pairs");
  const bool oc = c.mode != STTRAN_MODE_PREDCLS;
  const bool tables = in->clip_union_feat != nullptr;
  if (!out->attention_distribution || !out->spatial_distribution || !out->contacting_distribution)
    return fail(h, STTRAN_ERR_INVALID, "forward: null output pointer");
  if (in->num_clips < 1 || (in->num_clips > 1 && !in->clip_num_frames))
    return fail(h, STTRAN_ERR_INVALID, "forward: bad clip description");
  if (tables) {
    if (!in->clip_features || !in->clip_pair_idx || !in->clip_labels || !in->clip_spatial_masks || !in->clip_num_boxes ||
        !in->clip_num_pairs || (oc && (!in->clip_boxes || !in->clip_distribution)))
      return fail(h, STTRAN_ERR_INVALID, "forward: incomplete per-clip pointer tables");
    if (!in->frame_counts || in->num_frames <= 0)
      return fail(h, STTRAN_ERR_INVALID, "forward: per-clip pointer tables need frame_counts");
    int64_t tb = 0, tp = 0;
    for (int i = 0; i < in->num_clips; ++i) {
      const int64_t bc = in->clip_num_boxes[i], pc = in->clip_num_pairs[i];
      if (bc < 0 || pc < 0 || (pc > 0 && bc <= 0)) return fail(h, STTRAN_ERR_INVALID, "forward: bad per-clip sizes");
      tb += bc; tp += pc;
      const auto mis = [](const void* p, uintptr_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) != 0; };
      if (bc > 0 && (!in->clip_features[i] || !in->clip_labels[i] || mis(in->clip_features[i], 16) || mis(in->clip_labels[i], 8) ||
                     (oc && (!in->clip_boxes[i] || !in->clip_distribution[i] || mis(in->clip_boxes[i], 4) || mis(in->clip_distribution[i], 4)))))
        return fail(h, STTRAN_ERR_INVALID, "forward: null or misaligned per-clip box tensor");
      if (pc > 0 && (!in->clip_pair_idx[i] || !in->clip_union_feat[i] || !in->clip_spatial_masks[i] || mis(in->clip_pair_idx[i], 8) ||
                     mis(in->clip_union_feat[i], 4) || mis(in->clip_spatial_masks[i], 4)))
        return fail(h, STTRAN_ERR_INVALID, "forward: null or misaligned per-clip pair tensor");
    }
    if (tb != B || tp != P) return fail(h, STTRAN_ERR_INVALID, "forward: per-clip sizes do not sum to num_boxes / num_pairs");
  } else {
    if (!in->features || !in->pair_idx || !in->labels || !in->union_feat || !in->spatial_masks)
      return fail(h, STTRAN_ERR_INVALID, "forward: null tensor pointer");
    if (oc && (!in->boxes || !in->distribution))
      return fail(h, STTRAN_ERR_INVALID, "forward: sgdet needs boxes and distribution");
  }
  if (oc && !out->distribution) return fail(h, STTRAN_ERR_INVALID, "forward: sgdet needs an output distribution");
  h->prof_stream = s;
  int rc;
  if (!h->finalized && (rc = sttran_finalize_weights(h))) return rc;
  if (h->gemm_engine != STTRAN_GEMM_FP32_MFMA && !h->planes_ready) {
    // split every GEMM weight into its three bf16 planes, once (also after a reload: load_tensor resets the flag)
    for (auto& kv : h->w) {
      Tensor& t = kv.second;
      if (!t.ld || !t.d) continue;
      const size_t bytes = (size_t)3 * t.shape[0] * t.ld * 2 + 256;
      if (!t.planes) HIPCK(hipMalloc(&t.planes, bytes));
      HIPCK(split_planes(s, t.d, t.ld, (int)t.shape[0], (int)t.shape[1], t.planes, t.ld));
    }
    {   // the 1x1 union conv's weight [256, feat_dim, 1, 1] is a [256, feat_dim] GEMM operand too (feat_dim % 32 == 0)
      Tensor& t = h->w["union_func1.weight"];
      const int64_t FDp = c.feat_dim;
      if (t.d) {
        if (!t.planes) HIPCK(hipMalloc(&t.planes, (size_t)3 * 256 * FDp * 2 + 256));
        HIPCK(split_planes(s, t.d, FDp, 256, (int)FDp, t.planes, FDp));
      }
    }
    if (h->fc_w) {      // stacked subj_fc | obj_fc weight of the grouped launch
      const int64_t ldf = pad32(c.feat_dim);
      if (!h->fc_planes) HIPCK(hipMalloc(&h->fc_planes, (size_t)3 * 1024 * ldf * 2 + 256));
      HIPCK(split_planes(s, h->fc_w, ldf, 1024, c.feat_dim, h->fc_planes, ldf));
    }
    if (h->w4_perm) {   // conv3x3 weight in its (ky, kx, ci) K order
      if (!h->w4_planes) HIPCK(hipMalloc(&h->w4_planes, (size_t)3 * 256 * 1152 * 2 + 256));
      HIPCK(split_planes(s, h->w4_perm, 1152, 256, 1152, h->w4_planes, 1152));
    }
    h->planes_ready = true;
    if (h->lanes.size() > 1) HIPCK(hipStreamSynchronize(s));      // the other lanes' streams read the planes too
  }
  if ((rc = ensure_workspace(h, P, B))) return rc;

  // ---- per-frame pair counts ---------------------------------------------------------------
  std::vector<int32_t> counts;
  if (in->frame_counts && in->num_frames > 0) {
    counts.assign(in->frame_counts, in->frame_counts + in->num_frames);
  } else {
    if (!in->im_idx) return fail(h, STTRAN_ERR_INVALID, "forward: neither frame_counts nor im_idx given");
    const size_t esz = in->im_idx_dtype == STTRAN_DTYPE_I64 ? 8 : 4;
    if (h->L->im_host_cap < (size_t)P * 8) {
      if (h->L->im_host) HIPCK(hipHostFree(h->L->im_host));
      HIPCK(hipHostMalloc(reinterpret_cast<void**>(&h->L->im_host), (size_t)P * 8));
      h->L->im_host_cap = (size_t)P * 8;
    }
    HIPCK(hipMemcpyAsync(h->L->im_host, in->im_idx, (size_t)P * esz, hipMemcpyDeviceToHost, s));
    HIPCK(hipStreamSynchronize(s));
    int64_t prev = -1;
    for (int64_t p = 0; p < P; ++p) {
      int64_t f;
      if (in->im_idx_dtype == STTRAN_DTYPE_I64) f = reinterpret_cast<const int64_t*>(h->L->im_host)[p];
      else if (in->im_idx_dtype == STTRAN_DTYPE_I32) f = h->L->im_host[p];
      else f = (int64_t)reinterpret_cast<const float*>(h->L->im_host)[p];
      if (f < prev || f < 0) return fail(h, STTRAN_ERR_ORDER, "forward: im_idx must be non-negative and sorted ascending");
      if ((size_t)f >= counts.size()) counts.resize((size_t)f + 1, 0);
      counts[(size_t)f]++;
      prev = f;
    }
    if (in->num_frames > (int)counts.size()) counts.resize(in->num_frames, 0);
  }
  int64_t tot = 0;
  for (int32_t v : counts) { if (v < 0) return fail(h, STTRAN_ERR_INVALID, "forward: negative frame count"); tot += v; }
  if (tot != P) return fail(h, STTRAN_ERR_INVALID, "forward: frame_counts do not sum to num_pairs");
  std::vector<int32_t> clips;
  if (in->num_clips == 1) clips.push_back((int32_t)counts.size());
  else {
    clips.assign(in->clip_num_frames, in->clip_num_frames + in->num_clips);
    int64_t tf = 0;
    for (int32_t v : clips) { if (v < 0) return fail(h, STTRAN_ERR_INVALID, "forward: negative clip length"); tf += v; }
    if (tf != (int64_t)counts.size()) return fail(h, STTRAN_ERR_INVALID, "forward: clip_num_frames do not sum to num_frames");
  }
  if (tables) {
    size_t f = 0;
    for (int i = 0; i < in->num_clips; ++i) {
      int64_t pc = 0;
      for (int j = 0; j < clips[i]; ++j) pc += counts[f++];
      if (pc != in->clip_num_pairs[i])
        return fail(h, STTRAN_ERR_INVALID, "forward: clip_num_pairs disagrees with the clip's frame_counts");
    }
  }

  // ---- index maps (cached while the layout repeats) -----------------------------------------
  const bool is_dsg = c.model == STTRAN_MODEL_DSG_DETR;
  // DSG-DETR builds its class sequences on the device (no read-back, cacheable, capturable); STTRAN_DSG_HOST_LAYOUT=1
  // takes round 1's host builder instead, which reads labels / pair_idx back on every call (kept for A/B tests).
  static const bool dsg_host_env = exp_env("STTRAN_DSG_HOST_LAYOUT") && atoi(exp_env("STTRAN_DSG_HOST_LAYOUT")) != 0;   // experiment builds only
  const bool dsg_dev = is_dsg && !dsg_host_env;
  const bool host_dsg = is_dsg && !dsg_dev;
  if (host_dsg || !(P == h->L->cached_P && counts == h->L->cached_counts && clips == h->L->cached_clips && h->L->lay.dsg_device == dsg_dev)) {
    std::vector<int32_t> buf;
    h->L->cached_P = -1;      // h->L->lay is about to change: the cache only becomes valid again once the upload is enqueued
    if (dsg_dev) {
      build_layout_dsg_static(counts, clips, P, c.num_obj_classes, buf, h->L->lay);
    } else if (is_dsg) {
      if (tables) return fail(h, STTRAN_ERR_INVALID, "forward: STTRAN_DSG_HOST_LAYOUT=1 reads a contiguous pair_idx (no pointer tables)");
      // the class sequences depend on labels[pair_idx[:,1]]: read both back (small) -- DSG-DETR is the
      // second model on the shared kernels, not the latency path
      std::vector<int64_t> hp((size_t)P * 2), hl((size_t)B);
      HIPCK(hipMemcpyAsync(hp.data(), in->pair_idx, hp.size() * 8, hipMemcpyDeviceToHost, s));
      HIPCK(hipMemcpyAsync(hl.data(), in->labels, hl.size() * 8, hipMemcpyDeviceToHost, s));
      HIPCK(hipStreamSynchronize(s));
      for (int64_t p = 0; p < 2 * P; ++p)
        if (hp[p] < 0 || hp[p] >= B) return fail(h, STTRAN_ERR_INVALID, "forward: pair_idx out of range");
      build_layout_dsg(counts, clips, P, hp.data(), hl.data(), buf, h->L->lay);
      for (size_t i = 0; i < (size_t)P; ++i)
        if (buf[h->L->lay.o_need + i] >= 400) return fail(h, STTRAN_ERR_LIMIT, "forward: more than 400 frames in a class sequence");
    } else {
      build_layout(counts, clips, P, buf, h->L->lay);
    }
    // (no limit on the pairs of a frame / window / class sequence: the attention streams its keys in chunks)
    if ((int64_t)buf.size() > kIdxIntsPerPair * h->L->capP + 64) return fail(h, STTRAN_ERR_INVALID, "forward: index buffer too small");
    if ((rc = upload_staged(h, s, buf.data(), buf.size() * 4, h->L->idx.p))) return rc;
    h->L->cached_P = host_dsg ? -1 : P; h->L->cached_counts = counts; h->L->cached_clips = clips;
  }
  const Lane::Layout& L = h->L->lay;
  const int32_t* ib = h->L->idx.as<int32_t>();
  const int* enc_off = ib + L.o_enc_off; const int* enc_len = ib + L.o_enc_len;
  const int* dec_off = ib + L.o_dec_off; const int* dec_len = ib + L.o_dec_len;
  const int* dec_src = ib + L.o_dec_src; const int* out_src = ib + L.o_out_src;
  const uint8_t* slot = reinterpret_cast<const uint8_t*>(ib + L.o_slot);
  const int* need = ib + L.o_need; const int* qbegin = ib + L.o_qbegin;
  const int* tok0 = ib + L.o_tok0; const int* tok1 = ib + L.o_tok1;
  int* dsg_scratch = nullptr;
  if (L.dsg_device) {
    // class sequences from labels[pair_idx[:, 1]] where they live: [dec_off | dec_len] per (clip, class) slot, then
    // dec_src, need, out_src per token / pair, then 4 P ints of scratch (launched behind pair_prep, which resolves the
    // pairs' classes and subjects through the chunk table)
    const int64_t Kseq = L.n_dec_seq;
    HIPCK(h->L->dsg.ensure((size_t)(2 * Kseq + 7 * P + 64) * 4));
    int* d = h->L->dsg.as<int32_t>();
    dec_off = d; dec_len = d + Kseq; dec_src = d + 2 * Kseq; need = d + 2 * Kseq + P; out_src = d + 2 * Kseq + 2 * P;
    dsg_scratch = d + 2 * Kseq + 3 * P;
  }

  // ---- where the inputs live: one chunk per clip (pointer tables) or one chunk for the contiguous batch ----
  ChunkTable tab{};
  const float *feat_base = nullptr, *union_base = nullptr, *mask_base = nullptr;
  {
    const int n = tables ? in->num_clips : 1;
    std::vector<int64_t> t((size_t)(2 * (n + 1) + 7 * n), 0);
    int64_t* pair_start = t.data(); int64_t* box_start = pair_start + n + 1; int64_t* ptr = box_start + n + 1;
    auto put = [&](int k, int i, const void* p) { ptr[(size_t)k * n + i] = (int64_t)reinterpret_cast<intptr_t>(p); };
    int base = -1;                                  // first chunk with pairs: the offsets are relative to ITS tensors
    for (int i = 0; i < n; ++i) {
      const int64_t pc = tables ? in->clip_num_pairs[i] : P, bc = tables ? in->clip_num_boxes[i] : B;
      pair_start[i + 1] = pair_start[i] + pc; box_start[i + 1] = box_start[i] + bc;
      if (base < 0 && pc > 0) base = i;
      put(0, i, tables ? (const void*)in->clip_features[i] : in->features);
      put(1, i, tables ? (const void*)in->clip_pair_idx[i] : in->pair_idx);
      put(2, i, tables ? (const void*)in->clip_labels[i] : in->labels);
      put(3, i, tables ? (const void*)in->clip_union_feat[i] : in->union_feat);
      put(4, i, tables ? (const void*)in->clip_spatial_masks[i] : in->spatial_masks);
      if (oc) {
        put(5, i, tables ? (const void*)in->clip_boxes[i] : in->boxes);
        put(6, i, tables ? (const void*)in->clip_distribution[i] : in->distribution);
      }
    }
    if (base < 0) return fail(h, STTRAN_ERR_EMPTY, "forward: entry has no pairs");
    feat_base = reinterpret_cast<const float*>(ptr[0 * (size_t)n + base]);
    union_base = reinterpret_cast<const float*>(ptr[3 * (size_t)n + base]);
    mask_base = reinterpret_cast<const float*>(ptr[4 * (size_t)n + base]);
    if (t != h->L->ctab_host) {
      h->L->ctab_host.clear();
      if (h->L->ctab.bytes < t.size() * 8) { HIPCK(hipStreamSynchronize(s)); HIPCK(h->L->ctab.ensure(t.size() * 8 * 2)); }
      if ((rc = upload_staged(h, s, t.data(), t.size() * 8, h->L->ctab.p))) return rc;
      h->L->ctab_host = t;
    }
    const int64_t* d = h->L->ctab.as<int64_t>();
    auto arr = [&](int k) { return reinterpret_cast<const void* const*>(d + 2 * (n + 1) + (size_t)k * n); };
    tab.n = n; tab.base = base; tab.pair_start = d; tab.box_start = d + n + 1;
    tab.features = arr(0); tab.pair_idx = arr(1); tab.labels = arr(2); tab.union_feat = arr(3); tab.masks = arr(4);
    tab.boxes = arr(5); tab.dist = arr(6);
  }
  int64_t* feat_off = h->L->poff.as<int64_t>();          // [2][P] subject / object feature rows
  int64_t* union_off = feat_off + 2 * P;              // [P]
  int64_t* mask_off = feat_off + 3 * P;               // [P]

  const int D = c.embed_dim, F = c.ffn_dim, FD = c.feat_dim, NC = c.num_obj_classes;
  const int64_t LD = pad32(D), LF = pad32(F);     // row strides of the [*, D] / [*, F] workspace buffers (ensure_workspace)
  float* X0 = h->L->x0.as<float>();
  float* QKV = h->L->qkv.as<float>(); float* ATT = h->L->att.as<float>(); float* Y = h->L->ybuf.as<float>();
  float* H = h->L->hbuf.as<float>(); float* F1 = h->L->f1.as<float>(); float* G = h->L->gbuf.as<float>();
  float* UNI = h->L->uni.as<float>(); float* V = h->L->vbuf.as<float>(); float* C2 = h->L->c2.as<float>();
  float* E = h->L->ebuf.as<float>();

  // ---- ObjectClassifier, sgdet + is_wks (lib/sttran.py:173-184) ------------------------------
  if (oc) {
    float* Z = h->L->zbuf.as<float>(); float* HO = h->L->hobj.as<float>();
    const int zd = FD + 200 + 128;
    const int64_t ldz = pad32(zd);
    {
      ProfScope ps(h, s, STTRAN_PROF_INDEX, 0, 4.0 * B * (2.0 * zd), "objcls_prep_kernel", B, zd, 0);
      HIPCK(launch_objcls_prep(s, tab, W(h, "object_classifier.obj_embed.weight"),
                               h->oc_pos_scale, h->oc_pos_shift, W(h, "object_classifier.pos_embed.1.weight"),
                               W(h, "object_classifier.pos_embed.1.bias"), Z, ldz, (int)B, FD, NC - 1, 200));
    }
    EpiLinear e1 = epi_plain(HO, 1024, W(h, "object_classifier.decoder_lin.0.bias"), 1);
    e1.scale = h->oc_bn_scale; e1.shift = h->oc_bn_shift;     // Linear -> BN -> ReLU
    if ((rc = run_linear(h, s, GemmOperand{Z, ldz, nullptr}, W(h, "object_classifier.decoder_lin.0.weight"), (int)B, 1024, zd, e1))) return rc;
    EpiLinear e2 = epi_plain(out->distribution, NC, W(h, "object_classifier.decoder_lin.3.bias"));
    if ((rc = run_linear(h, s, GemmOperand{HO, 1024, nullptr}, W(h, "object_classifier.decoder_lin.3.weight"), (int)B, NC, 1024, e2))) return rc;
  }

  // ---- pair fusion (lib/sttran.py:381-399) -> X0 [P, 1936] -----------------------------------
  {
    ProfScope ps(h, s, STTRAN_PROF_INDEX, 0, 4.0 * P * 400 * 2, "pair_prep_kernel", P, 400, 0);
    HIPCK(launch_pair_prep(s, tab, (int)P, FD, NC, W(h, "obj_embed.weight"), W(h, "obj_embed2.weight"), 200, feat_off,
                           union_off, mask_off, dsg_scratch, dsg_scratch ? dsg_scratch + P : nullptr, X0, (int)LD, 1536,
                           h->L->err_flag));
  }
  if (L.dsg_device) {
    const int64_t Kseq = L.n_dec_seq;
    int* d = h->L->dsg.as<int32_t>();
    HIPCK(launch_dsg_layout(s, nullptr, nullptr, (int)B, ib + L.o_clip_start, L.num_clips, c.num_obj_classes, (int)P, 400,
                            1 << 30, d, d + Kseq, d + 2 * Kseq, d + 2 * Kseq + P, d + 2 * Kseq + 2 * P, dsg_scratch,
                            h->L->err_flag, L.max_dec));
  }
  // subject / object rows of `features` gathered by element offset (one chunk per clip: GemmOperand::rowoff)
  // subj_fc | obj_fc in ONE launch: N = 1024 over the stacked weights, columns >= 512 read their rows
  // through the second gather table (feat_off + P); X0 columns [0, 512) and [512, 1024) are adjacent
  if ((rc = run_linear(h, s, GemmOperand{feat_base, FD, nullptr, 512, feat_off}, h->fc_w, (int)P, 1024, FD,
                       epi_plain(X0, LD, h->fc_b)))) return rc;
  // the two convolutions on the 16x16x4 kernel structure (gemm_f32_t16c.h); STTRAN_CONV_ENGINE=32x32 keeps round 2's
  // gemm_sk_kernel<B_UNION_FLAT / B_CONV2> for A/B runs
  static const bool conv_t16 = !(exp_env("STTRAN_CONV_ENGINE") && std::string(exp_env("STTRAN_CONV_ENGINE")) == "32x32");   // experiment builds only
  {
    // conv stack of the spatial masks (lib/sttran.py:337-345), both convolutions as implicit GEMMs
    {
      // conv 7x7/2 -> ReLU -> BN -> max-pool in one kernel; the [P,128,14,14] intermediate stays on chip
      ProfScope ps(h, s, STTRAN_PROF_MASK_CONV, 2.0 * P * 128 * 196 * 98, 4.0 * P * (1458 + 128 * 49), "mask_conv1_pool_kernel",
                   128, P * 196, 98);
      HIPCK(launch_mask_conv1_pool(s, mask_base, mask_off, h->w0_perm, W(h, "conv.0.bias"), h->bn1_scale,
                                   h->bn1_shift, C2, (int)P));
    }
    EpiConvRelBn e2{V, W(h, "conv.4.bias"), h->bn2_scale, h->bn2_shift, 256, 49};
    const bool x3 = h->gemm_engine != STTRAN_GEMM_FP32_MFMA && h->planes_ready && h->w4_planes &&
                    (P * 49 >= 512 || h->gemm_engine == STTRAN_GEMM_BF16X3_ALL);
    ProfScope ps(h, s, STTRAN_PROF_GEMM, gemm_flops(256, P * 49, 1152), gemm_bytes(256, P * 49, 1152),
                 x3 ? "gemm_x3_kernel<X3Tile<128,256,2,4,A_CONV2>,EpiConvRows>"
                    : conv_t16 ? "gemm16c_kernel<Tile16C<B_CONV2>,EpiConvT16>"
                               : "gemm_sk_kernel<GemmTile<256,128,4,2,B_CONV2>,EpiConvRelBn>", 256, P * 49, 1152);
    if (x3)
      HIPCK(launch_mask_conv2_x3(s, h->w4_planes, C2, W(h, "conv.4.bias"), h->bn2_scale, h->bn2_shift, V, (int)P,
                                 h->L->slab.as<float>()));
    else if (conv_t16)
      HIPCK(launch_mask_conv2_t16(s, h->w4_perm, C2, W(h, "conv.4.bias"), h->bn2_scale, h->bn2_shift, V, (int)P, h->L->slab.as<float>()));
#ifdef STTRAN_GEMM_EXPERIMENT
    else
      HIPCK(launch_mask_conv2(s, h->w4_perm, C2, e2, (int)P, h->L->slab.as<float>()));
#else
    (void)e2;
#endif
  }
  {
    const Tensor& wu = h->w["union_func1.weight"];
    const bool x3 = h->gemm_engine != STTRAN_GEMM_FP32_MFMA && h->planes_ready && wu.planes &&
                    (P * 49 >= 512 || h->gemm_engine == STTRAN_GEMM_BF16X3_ALL);
    ProfScope ps(h, s, STTRAN_PROF_UNION_CONV, 2.0 * P * 256 * 49 * FD, 4.0 * P * (49.0 * FD + 2 * 12544) + 4.0 * 256 * FD,
                 x3 ? "gemm_x3_kernel<X3Tile<128,256,2,4,A_UNION_FLAT>,EpiUnionRows>"
                    : conv_t16 ? "gemm16c_kernel<Tile16C<B_UNION_FLAT>,EpiUnionT16>"
                               : "gemm_sk_kernel<GemmTile<256,128,4,2,B_UNION_FLAT>,EpiUnionFlat>", 256, P * 49, FD);
    if (x3)
      HIPCK(launch_union_conv_x3(s, union_base, union_off, wu.planes, W(h, "union_func1.bias"), V, (int)P, FD,
                                 h->L->slab.as<float>()));
    else if (conv_t16)
      HIPCK(launch_union_conv_t16(s, union_base, union_off, W(h, "union_func1.weight"), W(h, "union_func1.bias"), V, (int)P, FD,
                                  h->L->slab.as<float>()));
#ifdef STTRAN_GEMM_EXPERIMENT
    else
      HIPCK(launch_union_conv(s, union_base, union_off, W(h, "union_func1.weight"), W(h, "union_func1.bias"), V, (int)P, FD,
                              h->L->slab.as<float>()));
#endif
  }
  if ((rc = run_linear(h, s, GemmOperand{V, 12544, nullptr}, W(h, "vr_fc.weight"), (int)P, 512, 12544,
                       epi_plain(X0 + 1024, LD, W(h, "vr_fc.bias"))))) return rc;
  // taps are dense [P, D] caller buffers
  auto tap = [&](float* dst, const float* src) {
    return hipMemcpy2DAsync(dst, (size_t)D * 4, src, (size_t)LD * 4, (size_t)D * 4, (size_t)P, hipMemcpyDeviceToDevice, s);
  };
  if (out->rel_features_tap) HIPCK(tap(out->rel_features_tap, X0));

  // ---- spatial encoder, one sequence per non-empty frame (lib/transformer.py:20-30,144) --------
  const bool dsg = c.model == STTRAN_MODEL_DSG_DETR;
  const float* xin = X0;
  const int n_enc_layers = dsg ? 1 : c.enc_layers;
  for (int i = 0; i < n_enc_layers; ++i) {
    const std::string p = (dsg ? "local_transformer.layers." : "glocal_transformer.local_attention.layers.") +
                          std::to_string(i);
    float* xout = (i == n_enc_layers - 1) ? UNI : E;
    if ((rc = run_encoder_layer(h, s, p, xin, xout, (int)P, enc_off, enc_len, L.n_enc_seq, L.max_enc))) return rc;
    xin = xout;
  }
  if (n_enc_layers == 0) HIPCK(hipMemcpyAsync(UNI, X0, (size_t)P * LD * 4, hipMemcpyDeviceToDevice, s));
  if (out->local_output_tap) HIPCK(tap(out->local_output_tap, UNI));

  const int NT = (int)L.n_dec_tok, NN = (int)L.n_need;
  float* UDEC = UNI + (size_t)P * LD;
  if (dsg) {
    // ---- DSG-DETR temporal encoder (lib/dsg_detr.py:545-564): one sequence per object class over the
    //      whole clip, sinusoidal PE by the pair's frame rank inside its sequence, 3 encoder layers.
    //      dec_src = pair of each sequence token, need = its PE row, dec_off/len = class sequences.
    {
      ProfScope ps(h, s, STTRAN_PROF_INDEX, 0, 12.0 * P * D, "gather_add_rows_kernel", P, D, 0);
      HIPCK(launch_gather_add_rows(s, UNI, LD, dec_src, W(h, "positional_encoder.pe"), D, need, G, LD, P, D));
    }
    const float* gin = G;
    for (int i = 0; i < 3; ++i) {
      float* gout = (i == 2) ? UDEC : (i == 0 ? E : G);
      if ((rc = run_encoder_layer(h, s, "global_transformer.layers." + std::to_string(i), gin, gout, (int)P, dec_off,
                                  dec_len, L.n_dec_seq, L.max_dec, L.dsg_device))) return rc;
      gin = gout;
    }
  } else
  // ---- temporal decoder over 2-frame windows (lib/transformer.py:49-58,147-163) ----------------
  if (NT > 0 && c.dec_layers > 0) {
    // window tokens G[r] = encoder row dec_src[r] (lib/transformer.py:153).  With two or more decoder layers the copy is
    // never materialised: layer 0 projects q|k|v per PAIR straight from UNI and takes its residual through the same
    // index; only a single-layer decoder (whose K/V and Q projections read token rows) builds G.
    const bool need_g0 = c.dec_layers == 1;
    if (need_g0) {
      ProfScope ps(h, s, STTRAN_PROF_INDEX, 0, 8.0 * NT * D, "gather_rows_kernel", NT, D, 0);
      HIPCK(launch_gather_rows(s, UNI, LD, dec_src, G, LD, NT, D));
    }
    for (int i = 0; i < c.dec_layers; ++i) {
      const std::string p = "glocal_transformer.global_attention.layers." + std::to_string(i);
      const bool last = i == c.dec_layers - 1;
      // The last layer only has to produce the NN rows the heads read (see build_layout): K and V are
      // still projected for every token, but Q, the output projection, LayerNorm and the FFN run on
      // the needed rows alone (gathered A operand, compact [NN, D] outputs).
      const int MQ = last ? NN : NT;
      const int* rows = last ? need : nullptr;
      const float* Win = W(h, p + ".multihead2.in_proj_weight");
      const float* bin = W(h, p + ".multihead2.in_proj_bias");
      if (!last && i == 0) {
        // first layer: the input rows of a pair's two tokens are the same encoder row, and the position
        // embedding only enters as a bias, so q|k|v are projected once per PAIR (P rows instead of NT)
        // and written to both token rows, each with the bias of its own slot
        EpiLinear eq = epi_plain(QKV, 3 * D, bin);
        eq.rowbias = h->dec[i].posbias; eq.rowslot = slot; eq.rb_cols = 2 * D; eq.rb_ld = 2 * D;
        eq.out_rowidx = tok0; eq.out_rowidx2 = tok1;
        if ((rc = run_linear(h, s, GemmOperand{UNI, LD, nullptr}, Win, (int)P, 3 * D, D, eq))) return rc;
      } else if (!last) {
        EpiLinear eq = epi_plain(QKV, 3 * D, bin);
        eq.rowbias = h->dec[i].posbias; eq.rowslot = slot; eq.rb_cols = 2 * D; eq.rb_ld = 2 * D;
        if ((rc = run_linear(h, s, GemmOperand{G, LD, nullptr}, Win, NT, 3 * D, D, eq))) return rc;
      } else {
        EpiLinear ekv = epi_plain(QKV + D, 3 * D, bin + D);                 // k | v columns, all tokens
        ekv.rowbias = h->dec[i].posbias + D; ekv.rowslot = slot; ekv.rb_cols = D; ekv.rb_ld = 2 * D;
        if ((rc = run_linear(h, s, GemmOperand{G, LD, nullptr}, Win + (size_t)D * pad32(D), NT, 2 * D, D, ekv))) return rc;
        EpiLinear eq = epi_plain(QKV, 3 * D, bin);                          // q columns, needed rows only
        eq.rowbias = h->dec[i].posbias; eq.rowslot = slot; eq.rb_cols = D; eq.rb_ld = 2 * D;
        eq.out_rowidx = need;
        if ((rc = run_linear(h, s, GemmOperand{G, LD, need, 0, nullptr, NT}, Win, NN, D, D, eq))) return rc;
      }
      {
        ProfScope ps(h, s, STTRAN_PROF_ATTENTION, 4.0 * MQ * L.max_dec * D, 4.0 * (MQ * 2.0 + NT * 2.0) * D, "attention", MQ,
                     L.max_dec, D);
        HIPCK(launch_attention(s, QKV, dec_off, dec_len, last ? qbegin : nullptr, L.n_dec_seq, L.max_dec, ATT, LD, D,
                               c.nhead));
      }
      EpiLinear eo = epi_plain(Y, LD, W(h, p + ".multihead2.out_proj.bias"));
      eo.res = G; eo.ldres = LD; eo.res_rowidx = rows;
      if (i == 0 && !need_g0) { eo.res = UNI; eo.res_rowidx = dec_src; }      // residual = the window token's encoder row
      if ((rc = run_linear(h, s, GemmOperand{ATT, LD, rows, 0, nullptr, NT}, W(h, p + ".multihead2.out_proj.weight"), MQ, D, D, eo))) return rc;
      {
        ProfScope ps(h, s, STTRAN_PROF_LAYERNORM, 0, 8.0 * MQ * D, "layernorm_kernel", MQ, D, 0);
        HIPCK(launch_layernorm(s, Y, LD, W(h, p + ".norm3.weight"), W(h, p + ".norm3.bias"), H, LD, MQ, D));
      }
      if ((rc = run_linear(h, s, GemmOperand{H, LD, nullptr}, W(h, p + ".linear1.weight"), MQ, F, D,
                           epi_plain(F1, LF, W(h, p + ".linear1.bias"), 1)))) return rc;
      EpiLinear e2 = epi_plain(last ? UDEC : G, LD, W(h, p + ".linear2.bias"));
      e2.res = H; e2.ldres = LD;
      if ((rc = run_linear(h, s, GemmOperand{F1, LF, nullptr}, W(h, p + ".linear2.weight"), MQ, D, F, e2))) return rc;
    }
  } else if (NT > 0) {
    // dec_layers == 0: windows pass through -- the needed rows are encoder rows
    HIPCK(launch_gather_rows(s, UNI, LD, dec_src, G, LD, NT, D));
    HIPCK(launch_gather_rows(s, G, LD, need, UDEC, LD, NN, D));
  }
  if (out->global_output_tap) HIPCK(launch_gather_rows(s, UNI, LD, out_src, out->global_output_tap, D, P, D));

  // ---- relation heads on the 'latter' rows (lib/sttran.py:404-409, lib/transformer.py:179-185) --
  {
    const int nh = c.attention_classes + c.spatial_classes + c.contact_classes;
    EpiHeads eh{out->attention_distribution, out->spatial_distribution, out->contacting_distribution, h->heads_b,
                c.attention_classes, c.spatial_classes, c.contact_classes};
    GemmPlan plan = plan_gemm(P, nh, D, TILE_64x64, 1);
    ProfScope ps(h, s, STTRAN_PROF_GEMM, gemm_flops(P, nh, D), gemm_bytes(P, nh, D),
                 "gemm_sk_kernel<GemmTile<64,64,2,2,B_KMAJOR_PAD>,EpiHeads>", P, nh, D);
    HIPCK(gemm_heads(s, GemmOperand{UNI, LD, out_src}, GemmOperand{h->heads_w, pad32(D), nullptr}, (int)P, nh, D, eh, plan,
                     h->L->slab.as<float>()));
  }
  if (h->prof_on) h->prof.forwards += 1;
  return STTRAN_OK;
}

}  // namespace

extern "C" {

int sttran_sync_check(SttranHandle* h, void* stream) {
  if (!h) return STTRAN_ERR_INVALID;
  HIPCK(hipSetDevice(h->cfg.device));
  int rc = sttran_lane_join(h, -1, stream);            // every lane's last forward precedes the wait below
  if (rc) return rc;
  HIPCK(hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream)));
  int flag = 0;
  for (Lane* L : h->lanes) {
    int f = 0;
    HIPCK(hipMemcpy(&f, L->err_flag, 4, hipMemcpyDeviceToHost));
    if (f) { HIPCK(hipMemset(L->err_flag, 0, 4)); HIPCK(hipDeviceSynchronize()); }
    flag |= f;
  }
  if (flag) {
    if (flag & 1) return fail(h, STTRAN_ERR_INDEX, "forward: pair_idx or labels out of range (values were clamped)");
    return fail(h, STTRAN_ERR_LIMIT, "forward: a class sequence spans more than 400 frames (the reference's positional-encoding table, lib/dsg_detr.py:25-48, has 400 rows)");
  }
  return STTRAN_OK;
}

int sttran_union_boxes_masks(const float* boxes, const int64_t* pair_idx, const float* im_idx, int64_t num_pairs,
                             int32_t pool, float* union_boxes, float* spatial_masks, void* stream) {
  if (!boxes || !pair_idx || !spatial_masks || num_pairs < 0 || pool <= 0 || pool > 64) return STTRAN_ERR_INVALID;
  return launch_union_boxes_masks(reinterpret_cast<hipStream_t>(stream), boxes, pair_idx, im_idx, (int)num_pairs, pool,
                                  union_boxes, spatial_masks) == hipSuccess ? STTRAN_OK : STTRAN_ERR_HIP;
}

int sttran_eval_recall(const SttranEvalInputs* in, uint8_t* flags, int32_t* status, void* stream) {
  if (!in || in->struct_size != (int32_t)sizeof(SttranEvalInputs) || !status) return STTRAN_ERR_INVALID;
  if (in->num_frames < 0 || in->num_pairs < 0 || in->num_boxes < 0 || in->num_gt_rels < 0) return STTRAN_ERR_INVALID;
  const int ncol = in->attention_classes + in->spatial_classes + in->contact_classes;
  // the semi-constraint rule reads columns 0,1 / 3,4 / 9,10 (lib/evaluation_recall.py:270-276)
  if (in->attention_classes < 2 || in->spatial_classes < 1 || in->contact_classes < 1 || ncol < 11 || ncol > 32)
    return STTRAN_ERR_INVALID;
  if (in->im_idx_dtype != STTRAN_DTYPE_F32 && in->im_idx_dtype != STTRAN_DTYPE_I64) return STTRAN_ERR_INVALID;
  if (in->num_frames == 0 || in->num_gt_rels == 0) return STTRAN_OK;
  if (!flags || !in->gt_box_off || !in->gt_boxes || !in->gt_classes || !in->gt_rel_off || !in->gt_rels)
    return STTRAN_ERR_INVALID;
  if (in->num_pairs > 0 && (!in->attention_logits || !in->spatial || !in->contacting || !in->pair_idx || !in->im_idx ||
                            !in->boxes || !in->classes || !in->obj_scores))
    return STTRAN_ERR_INVALID;
  hipError_t err = launch_eval_recall(reinterpret_cast<hipStream_t>(stream), in->attention_logits, in->spatial,
                                      in->contacting, in->pair_idx, in->im_idx, in->im_idx_dtype == STTRAN_DTYPE_I64,
                                      in->boxes, in->classes, in->obj_scores, in->num_pairs, in->num_boxes,
                                      in->attention_classes, in->spatial_classes, in->contact_classes, in->num_frames,
                                      in->gt_box_off, in->gt_boxes, in->gt_classes, in->gt_rel_off, in->gt_rels,
                                      in->iou_threshold, flags, status);
  return err == hipSuccess ? STTRAN_OK : STTRAN_ERR_HIP;
}

int32_t sttran_eval_max_pairs(int32_t num_predicates) { return eval_max_pairs_per_frame(num_predicates); }

int64_t sttran_objcls_scratch_bytes(int64_t num_boxes, int32_t num_frames) {
  if (num_boxes < 0 || num_frames < 0) return 0;
  return (int64_t)objcls_scratch_bytes(num_boxes, num_frames);
}

int sttran_objcls_select(const SttranObjclsSelect* a, int64_t* num_boxes_out, int64_t* num_pairs_out, void* stream) {
  if (!a || a->struct_size != sizeof(SttranObjclsSelect) || !num_boxes_out || !num_pairs_out) return STTRAN_ERR_INVALID;
  if (a->num_boxes <= 0 || a->num_frames <= 0) return STTRAN_ERR_EMPTY;
  if (a->num_boxes > (1 << 26) || a->num_cols < 2 || a->num_cols > 64 || a->feat_dim < 0 || a->capacity < 4 * a->num_boxes)
    return STTRAN_ERR_INVALID;
  if (!a->boxes || !a->distribution || !a->pred_labels || !a->out_boxes || !a->out_distribution || !a->out_pred_scores ||
      !a->out_pred_labels || !a->out_pair_idx || !a->out_im_idx || !a->out_human_idx || !a->scratch ||
      (a->features != nullptr) != (a->out_features != nullptr) || (a->features && a->feat_dim <= 0) ||
      a->scratch_bytes < (int64_t)objcls_scratch_bytes(a->num_boxes, a->num_frames))
    return STTRAN_ERR_INVALID;
  int32_t host[3] = {0, 0, 0};
  hipError_t e = launch_objcls_select(reinterpret_cast<hipStream_t>(stream), a->boxes, a->distribution, a->features, a->pred_labels,
                                      a->num_boxes, a->num_frames, a->num_cols, a->feat_dim, a->nms_threshold, a->nms_ge, a->capacity,
                                      a->out_boxes, a->out_distribution, a->out_features, a->out_pred_scores, a->out_pred_labels,
                                      a->out_source_row, a->out_pair_idx, a->out_im_idx, a->out_human_idx, a->scratch, host);
  if (e != hipSuccess) return STTRAN_ERR_HIP;
  if (host[2] & 2) return STTRAN_ERR_ORDER;        // boxes not sorted by frame id, or a frame id outside [0, num_frames)
  *num_boxes_out = host[0];
  *num_pairs_out = host[1];
  return STTRAN_OK;
}

int sttran_roi_align(const float* fmaps, int32_t T, int32_t C, int32_t H, int32_t W, const float* rois, int64_t num_rois,
                     int32_t pooled, float spatial_scale, int32_t sampling_ratio, float* out, void* stream) {
  if (!fmaps || T <= 0 || C <= 0 || H <= 0 || W <= 0 || num_rois < 0 || pooled <= 0 || pooled > 64 || (num_rois > 0 && (!rois || !out)))
    return STTRAN_ERR_INVALID;
  return launch_roi_align(reinterpret_cast<hipStream_t>(stream), fmaps, T, C, H, W, rois, num_rois, pooled, spatial_scale,
                          sampling_ratio, out) == hipSuccess ? STTRAN_OK : STTRAN_ERR_HIP;
}

// ---- kernel-level test hooks -------------------------------------------------------------------
int sttran_debug_gemm(const float* A, const int32_t* a_rowidx, const float* Wt, const float* bias,
                      const float* residual, float* C, int64_t M, int64_t N, int64_t K, int32_t relu,
                      int32_t tile_cfg, int32_t split_k, void* stream) {
  if (!A || !Wt || !C || M <= 0 || N <= 0 || K <= 0 || (K & 3)) return STTRAN_ERR_INVALID;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  GemmPlan plan = plan_gemm(M, N, K, tile_cfg, split_k);
  static float* slab = nullptr;   // test hook only: one park buffer for the life of the process
  if (!slab && (hipMalloc(reinterpret_cast<void**>(&slab), gemm_slab_bytes()) != hipSuccess ||
                hipMemset(slab, 0, gemm_slab_bytes()) != hipSuccess))
    return STTRAN_ERR_HIP;
  EpiLinear e = epi_plain(C, N, bias, relu);
  e.res = residual; e.ldres = N;
  // arbitrary caller tensors: the select path (B_KMAJOR); with K % 32 == 0 there is no K tail, so the select-free
  // product path (B_KMAJOR_PAD) is equally valid and is what gets measured
  hipError_t err = gemm_linear(s, GemmOperand{A, K, a_rowidx}, GemmOperand{Wt, K, nullptr}, (int)M, (int)N, (int)K,
                               e, plan, slab, K % 32 == 0 ? 1 : 0);
  return err == hipSuccess ? STTRAN_OK : STTRAN_ERR_HIP;
}

int sttran_debug_gemm_padded(const float* A, int64_t lda, const int32_t* a_rowidx, const float* Wt, int64_t ldw,
                             const float* bias, const float* residual, float* C, int64_t M, int64_t N, int64_t K,
                             int32_t relu, int32_t tile_cfg, void* stream) {
  if (!A || !Wt || !C || M <= 0 || N <= 0 || K <= 0 || (K & 3) || lda < pad32(K) || ldw < pad32(K) || (lda & 3) || (ldw & 3))
    return STTRAN_ERR_INVALID;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  GemmPlan plan = plan_gemm(M, N, K, tile_cfg, 0);
  static float* slab = nullptr;   // test hook only: one park buffer for the life of the process
  if (!slab && (hipMalloc(reinterpret_cast<void**>(&slab), gemm_slab_bytes()) != hipSuccess ||
                hipMemset(slab, 0, gemm_slab_bytes()) != hipSuccess))
    return STTRAN_ERR_HIP;
  EpiLinear e = epi_plain(C, N, bias, relu);
  e.res = residual; e.ldres = N;
  // a gathered operand's span (GemmOperand::span) is what the forward knows from its buffers; a test hook reads the index
  // back (one synchronisation) so that the 16x16x4 tiles can be exercised with gathered rows
  int64_t span = 0;
  if (a_rowidx) {
    std::vector<int32_t> idx((size_t)M);
    if (hipMemcpyAsync(idx.data(), a_rowidx, (size_t)M * 4, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess)
      return STTRAN_ERR_HIP;
    for (int32_t v : idx) { if (v < 0) return STTRAN_ERR_INVALID; span = std::max<int64_t>(span, (int64_t)v + 1); }
  }
  hipError_t err = gemm_linear(s, GemmOperand{A, lda, a_rowidx, 0, nullptr, span}, GemmOperand{Wt, ldw, nullptr}, (int)M, (int)N,
                               (int)K, e, plan, slab, 1);
  return err == hipSuccess ? STTRAN_OK : STTRAN_ERR_HIP;
}

int sttran_debug_gemm_emulated(const float* A, int64_t lda, const int32_t* a_rowidx, const float* Wt, int64_t ldw,
                         const float* bias, const float* residual, float* C, int64_t M, int64_t N, int64_t K,
                         int32_t relu, void* stream) {
  if (!A || !Wt || !C || M <= 0 || N <= 0 || K <= 0 || (K & 3) || lda < pad32(K) || ldw < K || (lda & 3)) return STTRAN_ERR_INVALID;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  static float* slab = nullptr;
  if (!slab && (hipMalloc(reinterpret_cast<void**>(&slab), gemm_slab_bytes()) != hipSuccess)) return STTRAN_ERR_HIP;
  // weight planes, cached for the last W pointer / shape (test hook: one matrix at a time)
  static void* planes = nullptr;
  static size_t planes_bytes = 0;
  static const float* cached_w = nullptr;
  static int64_t cached_n = 0, cached_k = 0;
  const int64_t ldp = pad32(K);
  const size_t need = (size_t)3 * N * ldp * 2 + 256;
  if (need > planes_bytes) {
    if (planes) hipFree(planes);
    if (hipMalloc(&planes, need) != hipSuccess) { planes = nullptr; planes_bytes = 0; return STTRAN_ERR_HIP; }
    planes_bytes = need;
    cached_w = nullptr;
  }
  // the planes are re-made on every call (a freed W may come back at the same address with other contents) unless the
  // caller vouches for W staying put: STTRAN_X3_CACHE_PLANES=1 (tools/gemm_bench.py times the GEMM alone that way)
  static const bool cache_ok = exp_env("STTRAN_X3_CACHE_PLANES") && atoi(exp_env("STTRAN_X3_CACHE_PLANES")) != 0;   // experiment builds only
  if (!cache_ok || cached_w != Wt || cached_n != N || cached_k != K) {
    if (split_planes(s, Wt, ldw, (int)N, (int)K, planes, ldp) != hipSuccess) return STTRAN_ERR_HIP;
    cached_w = Wt; cached_n = N; cached_k = K;
  }
  EpiLinear e = epi_plain(C, N, bias, relu);
  e.res = residual; e.ldres = N;
  hipError_t err = gemm_linear_x3(s, GemmOperand{A, lda, a_rowidx}, planes, ldp, N * ldp, (int)M, (int)N, (int)K, e, slab);
  return err == hipSuccess ? STTRAN_OK : STTRAN_ERR_HIP;
}

// Test allocator: `bytes` of device memory whose end is the end of the mapping -- the page behind it is reserved address
// space with nothing mapped, so a kernel that reads or writes past a caller's buffer faults instead of silently touching
// a neighbour (tests/test_guarded_buffers_gpu.py).  HIP virtual-memory API; STTRAN_ERR_HIP where the driver has none.
namespace {
struct GuardedAlloc { void* base; size_t reserved, mapped; hipMemGenericAllocationHandle_t handle; };
}
int sttran_debug_guarded_alloc(size_t bytes, void** ptr, void** cookie) {
  if (!ptr || !cookie || bytes == 0) return STTRAN_ERR_INVALID;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return STTRAN_ERR_HIP;
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = dev;
  size_t gran = 0;
  if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum) != hipSuccess || gran == 0) return STTRAN_ERR_HIP;
  auto* g = new GuardedAlloc{};
  g->mapped = (bytes + gran - 1) / gran * gran;
  g->reserved = g->mapped + gran;                                  // one unmapped granule behind the data
  bool ok = hipMemAddressReserve(&g->base, g->reserved, gran, nullptr, 0) == hipSuccess;
  bool created = false, mapped = false;
  if (ok) ok = created = hipMemCreate(&g->handle, g->mapped, &prop, 0) == hipSuccess;
  if (ok) ok = mapped = hipMemMap(g->base, g->mapped, 0, g->handle, 0) == hipSuccess;
  if (ok) {
    hipMemAccessDesc acc{};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    ok = hipMemSetAccess(g->base, g->mapped, &acc, 1) == hipSuccess;
  }
  if (!ok) {
    if (mapped) hipMemUnmap(g->base, g->mapped);
    if (created) hipMemRelease(g->handle);
    if (g->base) hipMemAddressFree(g->base, g->reserved);
    delete g;
    (void)hipGetLastError();
    return STTRAN_ERR_HIP;
  }
  const size_t span = (bytes + 15) & ~size_t(15);                  // 16-byte aligned start, <= 15 bytes of slack at the end
  *ptr = static_cast<char*>(g->base) + (g->mapped - span);
  *cookie = g;
  return STTRAN_OK;
}
int sttran_debug_guarded_free(void* cookie) {
  if (!cookie) return STTRAN_ERR_INVALID;
  auto* g = static_cast<GuardedAlloc*>(cookie);
  hipDeviceSynchronize();
  hipMemUnmap(g->base, g->mapped);
  hipMemRelease(g->handle);
  hipMemAddressFree(g->base, g->reserved);
  delete g;
  return STTRAN_OK;
}

int sttran_debug_plan_tile(int64_t M, int64_t N, int64_t K) {
  if (M <= 0 || N <= 0 || K <= 0) return -STTRAN_ERR_INVALID;
  return plan_gemm(M, N, K, 0, 0).tile;
}

int sttran_debug_mfma_peak(int32_t iters, double* tflops) {
  if (iters <= 0 || !tflops) return STTRAN_ERR_INVALID;
  float* out = nullptr;
  int ncu = 256, dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&out), 256) != hipSuccess) return STTRAN_ERR_HIP;
  hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  launch_mfma_peak(nullptr, out, iters, ncu);            // warm-up
  hipEventRecord(a, nullptr);
  launch_mfma_peak(nullptr, out, iters, ncu);
  hipEventRecord(b, nullptr);
  hipEventSynchronize(b);
  float ms = 0.f;
  hipEventElapsedTime(&ms, a, b);
  hipEventDestroy(a); hipEventDestroy(b); hipFree(out);
  // per block: 4 waves x iters x 4 MFMAs x (32*32*2*2 flops)
  *tflops = (double)ncu * 4.0 * iters * 4.0 * 4096.0 / (ms * 1e-3) / 1e12;
  return STTRAN_OK;
}

int sttran_debug_layernorm(const float* x, const float* gamma, const float* beta, float* y, int64_t rows,
                           int64_t dim, void* stream) {
  if (!x || !gamma || !beta || !y) return STTRAN_ERR_INVALID;
  return launch_layernorm(reinterpret_cast<hipStream_t>(stream), x, dim, gamma, beta, y, dim, rows, (int)dim) == hipSuccess
             ? STTRAN_OK : STTRAN_ERR_HIP;
}

int sttran_debug_attention(const float* qkv, const int32_t* seq_off, const int32_t* seq_len, int32_t num_seq,
                           int32_t max_len, float* out, int64_t tokens, int32_t dim, int32_t nhead, void* stream) {
  (void)tokens;
  if (!qkv || !seq_off || !seq_len || !out || nhead <= 0 || dim % nhead) return STTRAN_ERR_INVALID;
  return launch_attention(reinterpret_cast<hipStream_t>(stream), qkv, seq_off, seq_len, nullptr, num_seq, max_len, out, dim,
                          dim, nhead) == hipSuccess ? STTRAN_OK : STTRAN_ERR_HIP;
}

// test hook: the DSG-DETR class-sequence tables exactly as the forward builds them on the device, and (len_bound > 0)
// the attention over sequences whose lengths only the device knows
int sttran_debug_dsg_layout(const int64_t* pair_idx, const int64_t* labels, int64_t num_boxes, const int32_t* clip_start,
                            int32_t num_clips, int32_t num_classes, int64_t num_pairs, int32_t pe_rows, int32_t* dec_off,
                            int32_t* dec_len, int32_t* dec_src, int32_t* need, int32_t* out_src, int32_t* scratch4p,
                            int32_t* err_flag, void* stream) {
  if (!pair_idx || !labels || !clip_start || !dec_off || !dec_len || !dec_src || !need || !out_src || !scratch4p || !err_flag)
    return STTRAN_ERR_INVALID;
  return launch_dsg_layout(reinterpret_cast<hipStream_t>(stream), pair_idx, labels, (int)num_boxes, clip_start, num_clips,
                           num_classes, (int)num_pairs, pe_rows, 1 << 30, dec_off, dec_len, dec_src, need, out_src, scratch4p,
                           err_flag, (int)std::min<int64_t>(num_pairs, 6144)) == hipSuccess ? STTRAN_OK : STTRAN_ERR_HIP;
}

int sttran_debug_attention_classes(const float* qkv, const int32_t* seq_off, const int32_t* seq_len, int32_t num_seq,
                                   int32_t len_bound, float* out, int32_t dim, int32_t nhead, void* stream) {
  if (!qkv || !seq_off || !seq_len || !out || nhead <= 0 || dim % nhead) return STTRAN_ERR_INVALID;
  return launch_attention_classes(reinterpret_cast<hipStream_t>(stream), qkv, seq_off, seq_len, num_seq, len_bound, out, dim,
                                  dim, nhead) == hipSuccess ? STTRAN_OK : STTRAN_ERR_HIP;
}

}  // extern "C"
