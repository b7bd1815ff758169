// api_internal.h -- what the translation units behind the C ABI (include/sttran_hip.h) share: the handle, a lane's
// workspace, and the host-side steps of one forward.  Nothing here is part of the boundary.
//   api_weights.hip  state-dict declaration, sttran_create / load_tensor / finalize_weights (derived parameters, bf16 planes)
//   api_layout.hip   index maps of a call (build_layout*), workspace growth, staged uploads
//   api_forward.hip  run_linear / run_encoder_layer / forward_on: STTran.forward and the DSG-DETR variant
//   api_lanes.hip    lanes, stream ordering, sttran_forward / forward_lane / reserve / sync_check
//   api_profile.hip  HIP-event profile of the launch sites (bench.py's roofline leg)
//   api_ops.hip      the entry points without a handle (union boxes, evaluator, ObjectClassifier selection, ROIAlign)
//   api_debug.hip    sttran_debug_* test hooks (include/sttran_hip_debug.h)
#pragma once
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

namespace sttran_host {
using namespace sttran;

// columns of a GEMM operand row that must be readable: the next multiple of the K-step (32)
inline int64_t pad32(int64_t k) { return (k + 31) / 32 * 32; }

struct Tensor {
  float* d = nullptr;
  void* d_guard = nullptr;      // STTRAN_GUARD_WORKSPACE: cookie of the guarded allocation behind `d`
  void* planes = nullptr;  // bf16x3 engine: [3][rows][ld] bf16 planes of a GEMM weight (made on demand)
  void* planes_fm = nullptr;   // ... and its fragment-major planes (gemm_bf16x3_t16.h), for the launches the 16x16x32 tiles serve
  std::vector<int64_t> shape;
  size_t n = 0;
  int64_t ld = 0;          // != 0: a [rows, cols] GEMM weight stored with this row stride (cols zero-padded to pad32)
  bool required = false, loaded = false;
};

// STTRAN_GUARD_WORKSPACE=1 (tests): every workspace buffer ENDS at the end of its mapping (sttran_debug_guarded_alloc), so
// a kernel that runs past one faults instead of reading its neighbour
inline bool guard_workspace() {
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

// weight tensors: hipMalloc, or a guarded allocation under STTRAN_GUARD_WORKSPACE (api_weights.hip)
hipError_t weight_alloc(Tensor& t, size_t bytes);
void weight_free(Tensor& t);

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


}  // namespace sttran_host

// Everything ONE forward in flight needs for itself: workspace, stream-K park space, index-map / chunk-table staging and
// their caches, the device-side error flag.  A handle owns one lane (the classic `sttran_forward` on the caller's stream)
// or several (`sttran_set_lanes` + `sttran_forward_lane`: each lane runs on its OWN stream, forked from the caller's with
// an event, so consecutive one-clip calls -- the reference's loop, tools/test_STTran.py:81-84 -- overlap on the device).
// The weights and derived parameters are shared (read-only during forwards).
struct Lane {
  int64_t capP = 0, capB = 0;
  sttran_host::DevBuf x0, qkv, att, ybuf, hbuf, f1, gbuf, uni, vbuf, c2, slab, idx, zbuf, hobj, ebuf;
  // bf16x3 engine, fragment-major planes (gemm_bf16x3_t16.h): aplanes = the activation operand of the GEMM in flight (made by
  // split_fm in front of it); hplanes = a LayerNorm's output, written by the LayerNorm itself (`hplanes_of` = the fp32 row
  // buffer it mirrors, `hplanes_rows` rows of it; 0 = stale); f1planes = linear1 -> ReLU, written by that GEMM's epilogue
  sttran_host::DevBuf aplanes, hplanes, f1planes;
  const float* hplanes_of = nullptr;
  int64_t hplanes_rows = 0;
  sttran_host::DevBuf dsg;                   // DSG-DETR: class-sequence tables built on the device (launch_dsg_layout)
  sttran_host::DevBuf ctab, poff;            // chunk table of the call's inputs (kernels.h ChunkTable); per-pair element offsets [4 P] int64
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
  std::map<std::string, sttran_host::Tensor> w;
  bool finalized = false;
  int gemm_engine = STTRAN_GEMM_FP32_MFMA;
  bool planes_ready = false;
  // derived parameters
  sttran_host::DevBuf derived;               // one arena for all derived tensors
  float *bn1_scale = nullptr, *bn1_shift = nullptr, *bn2_scale = nullptr, *bn2_shift = nullptr;
  float *heads_w = nullptr, *heads_b = nullptr, *w0_perm = nullptr, *w4_perm = nullptr;
  float *fc_w = nullptr, *fc_b = nullptr;   // [subj_fc ; obj_fc] stacked: weights [1024, feat_dim], bias [1024] (one grouped launch)
  void* w4_planes = nullptr;    // bf16x3 engine: [3][256][1152] bf16 planes of w4_perm (made on demand)
  void* w4_planes_fm = nullptr; // ... and its fragment-major planes (gemm_bf16x3_t16c.h)
  void* fc_planes = nullptr;    // ... [3][1024][feat_dim] planes of the stacked subj_fc | obj_fc weight
  float *oc_pos_scale = nullptr, *oc_pos_shift = nullptr, *oc_bn_scale = nullptr, *oc_bn_shift = nullptr;
  std::vector<sttran_host::DecLayer> dec;
  std::vector<Lane*> lanes;     // >= 1
  Lane* L = nullptr;            // the lane of the call in progress (calls on a handle are serialised by the caller)
  // profiling
  bool prof_on = false;
  std::vector<sttran_host::ProfEvent> prof_ev;
  std::map<sttran_host::ProfKey, int> prof_index;            // (kernel, shape) -> entry
  std::vector<SttranProfEntry> prof_entries;
  SttranProfile prof{};
  hipStream_t prof_stream = nullptr;
};

namespace sttran_host {

int fail(SttranHandle* h, int code, const std::string& msg);
#define HIPCK(expr)                                                                            \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess)                                                                      \
      return fail(h, STTRAN_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));       \
  } while (0)

void declare_weights(SttranHandle* h);                     // api_weights.hip
inline const float* W(SttranHandle* h, const std::string& k) { return h->w[k].d; }
const char* tile_name(int tile);                           // api_forward.hip

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

inline double gemm_flops(int64_t M, int64_t N, int64_t K) { return 2.0 * M * N * K; }
inline double gemm_bytes(int64_t M, int64_t N, int64_t K) { return 4.0 * (M * K + N * K + M * N); }

// ---- api_forward.hip
int run_linear(SttranHandle* h, hipStream_t s, GemmOperand A, const float* Wt, int M, int N, int K, EpiLinear epi,
               int force_tile = 0, int force_split = 0);
EpiLinear epi_plain(float* C, int64_t ldc, const float* bias, int relu = 0);
int run_encoder_layer(SttranHandle* h, hipStream_t s, const std::string& p, const float* xin, float* xout, int M,
                      const int* seq_off, const int* seq_len, int nseq, int maxlen, bool len_on_device = false);
int forward_on(SttranHandle* h, const SttranInputs* in_, const SttranOutputs* out, hipStream_t s);
// LayerNorm whose output the next GEMM reads: under the bf16x3 engine it also writes that GEMM's activation planes
int run_layernorm(SttranHandle* h, hipStream_t s, const float* x, const float* gamma, const float* beta, float* y, int M);
// y = res + linear2(relu(linear1(x))): two run_linear calls, or (bf16x3 engine) planes in, planes between, no split pass
// planes_out: additionally leave the output as the next GEMM's activation planes (it is an in_proj's input)
int run_ffn(SttranHandle* h, hipStream_t s, const std::string& p, const float* x, float* f1, int M, EpiLinear e2, bool planes_out = false);
// ---- api_layout.hip
int ensure_workspace(SttranHandle* h, int64_t P, int64_t B);
int upload_staged(SttranHandle* h, hipStream_t s, const void* src, size_t bytes, void* dst);
void build_layout(const std::vector<int32_t>& counts, const std::vector<int32_t>& clips, int64_t P,
                  std::vector<int32_t>& buf, Lane::Layout& L);
void build_layout_dsg(const std::vector<int32_t>& counts, const std::vector<int32_t>& clips, int64_t P,
                      const int64_t* pair_idx, const int64_t* labels, std::vector<int32_t>& buf, Lane::Layout& L);
void build_layout_dsg_static(const std::vector<int32_t>& counts, const std::vector<int32_t>& clips, int64_t P, int NC,
                             std::vector<int32_t>& buf, Lane::Layout& L);
// ---- api_lanes.hip
int lane_create(SttranHandle* h, Lane** out);
void lane_destroy(Lane* L);
bool capturing(hipStream_t s);

}  // namespace sttran_host
