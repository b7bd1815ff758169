// api_forward.hip -- host orchestration of STTran.forward (lib/sttran.py:375-411 -> lib/transformer.py:130-187 with the
// empty-frame handling of lib/transformer_wk.py:144-195) and of the DSG-DETR variant (lib/dsg_detr.py:514-572).  Host work
// per call: O(P) integer index maps (api_layout.hip); everything else is enqueued on the caller's stream.
#include "api_internal.h"

using namespace sttran;
using namespace sttran_host;

namespace sttran_host {

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

// bf16x3 engine: is this launch served by the emulation at all (sttran_set_gemm_engine; >= 512 rows unless BF16X3_ALL)
static bool x3_on(const SttranHandle* h, int M) {
  return h->gemm_engine != STTRAN_GEMM_FP32_MFMA && h->planes_ready && (M >= 512 || h->gemm_engine == STTRAN_GEMM_BF16X3_ALL);
}
// fragment-major planes of the weight rows [Wt, Wt + N) (a whole GEMM weight or a 16-row-aligned row range of one: the last
// decoder layer projects k|v and q separately); row_blocks = 16-row blocks available from there.  nullptr: not served.
static const void* fm_weight(SttranHandle* h, const float* Wt, int N, int K, int& row_blocks) {
  for (auto& kv : h->w) {
    const Tensor& t = kv.second;
    if (!t.planes_fm || !t.ld || t.ld != pad32(K)) continue;
    const int64_t rows = t.shape[0];
    if (Wt < t.d || Wt >= t.d + rows * t.ld) continue;
    const int64_t r0 = (Wt - t.d) / t.ld;
    if ((Wt - t.d) % t.ld || r0 + N > rows || (r0 & 15)) return nullptr;
    row_blocks = (int)((rows + 15) / 16 - r0 / 16);
    return reinterpret_cast<const uint16_t*>(t.planes_fm) + (r0 / 16) * ((K + 31) / 32) * 1536;
  }
  return nullptr;
}

// C = act(A W^T + ...) through the planner; slab workspace grown on demand
int run_linear(SttranHandle* h, hipStream_t s, GemmOperand A, const float* Wt, int M, int N, int K, EpiLinear epi,
               int force_tile, int force_split) {
  if (M <= 0) return STTRAN_OK;
  GemmPlan plan = plan_gemm(M, N, K, force_tile, force_split);
  if (gemm_slab_bytes() > h->L->slab.bytes) {
    HIPCK(hipStreamSynchronize(s));
    HIPCK(h->L->slab.ensure(gemm_slab_bytes()));
  }
  if (h->gemm_engine != STTRAN_GEMM_FP32_MFMA && h->planes_ready && (M >= 512 || h->gemm_engine == STTRAN_GEMM_BF16X3_ALL) &&
      N >= 128 && !force_tile) {
    // (1) the 16x16x32 tiles on pre-split fragment-major operands (gemm_bf16x3_t16.h): N a multiple of 176 or 128, K <= 4096
    //     (vr_fc's K = 12 544 operand would need 0.85 GB of planes: it stays on the in-loader split below), one gather table
    const int t16 = (K <= 4096 && !(A.aux > 0)) ? x3t16_tile(N, epi) : 0;
    int wrb = 0;
    const void* wfm = t16 ? fm_weight(h, Wt, N, K, wrb) : nullptr;
    if (wfm) {
      // the activation planes: a LayerNorm wrote them already (run_layernorm), else one split pass
      const bool have = !A.rowidx && !A.rowoff && A.ptr == h->L->hplanes_of && M <= h->L->hplanes_rows && K == h->cfg.embed_dim;
      if (!have) {
        const size_t need = fm_planes_bytes(M, K) + 256;
        if (need > h->L->aplanes.bytes) {
          HIPCK(hipStreamSynchronize(s));
          HIPCK(h->L->aplanes.ensure(need + need / 4));
        }
      }
      ProfScope ps(h, s, STTRAN_PROF_GEMM, gemm_flops(M, N, K), gemm_bytes(M, N, K),
                   std::string("gemm16x3_kernel<Tile16<") + tile_name(t16) + (have ? ">,EpiLinear>" : ">,EpiLinear> + split_fm_kernel"), M, N, K);
      if (!have) HIPCK(split_fm(s, A.ptr, A.ld, A.rowidx, A.rowoff, M, K, h->L->aplanes.p));
      HIPCK(gemm_linear_x3t16(s, have ? h->L->hplanes.p : h->L->aplanes.p, wfm, wrb, M, N, K, epi, h->L->slab.as<float>()));
      return STTRAN_OK;
    }
    // (2) round 2's kernel (gemm_bf16x3.h): activations split by the A loader
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

EpiLinear epi_plain(float* C, int64_t ldc, const float* bias, int relu) {
  EpiLinear e{};
  e.C = C; e.ldc = ldc; e.bias = bias; e.relu = relu;
  return e;
}

int run_layernorm(SttranHandle* h, hipStream_t s, const float* x, const float* gamma, const float* beta, float* y, int M) {
  const int D = h->cfg.embed_dim;
  const int64_t LD = pad32(D);
  void* planes = nullptr;
  if (x3_on(h, M)) {
    const size_t need = fm_planes_bytes(M, D) + 256;
    if (need > h->L->hplanes.bytes) {
      HIPCK(hipStreamSynchronize(s));
      HIPCK(h->L->hplanes.ensure(need + need / 4));
    }
    planes = h->L->hplanes.p;
  }
  h->L->hplanes_of = planes ? y : nullptr;
  h->L->hplanes_rows = planes ? M : 0;
  ProfScope ps(h, s, STTRAN_PROF_LAYERNORM, 0, 8.0 * M * D, planes ? "layernorm_kernel<planes>" : "layernorm_kernel", M, D, 0);
  HIPCK(launch_layernorm(s, x, LD, gamma, beta, y, LD, M, D, planes));
  return STTRAN_OK;
}

int run_ffn(SttranHandle* h, hipStream_t s, const std::string& p, const float* x, float* f1, int M, EpiLinear e2, bool planes_out) {
  const int D = h->cfg.embed_dim, F = h->cfg.ffn_dim;
  const int64_t LD = pad32(D), LF = pad32(F);
  int rb1 = 0, rb2 = 0;
  const void *w1 = nullptr, *w2 = nullptr;
  if (x3_on(h, M) && F % 128 == 0 && F <= 4096 && x3t16_tile(D, e2) &&
      (w1 = fm_weight(h, W(h, p + ".linear1.weight"), F, D, rb1)) && (w2 = fm_weight(h, W(h, p + ".linear2.weight"), D, F, rb2))) {
    const bool have = x == h->L->hplanes_of && M <= h->L->hplanes_rows;
    const size_t need_a = have ? 0 : fm_planes_bytes(M, D) + 256, need_f = fm_planes_bytes(M, F) + 256;
    if (need_a > h->L->aplanes.bytes || need_f > h->L->f1planes.bytes || gemm_slab_bytes() > h->L->slab.bytes) {
      HIPCK(hipStreamSynchronize(s));
      if (need_a > h->L->aplanes.bytes) HIPCK(h->L->aplanes.ensure(need_a + need_a / 4));
      if (need_f > h->L->f1planes.bytes) HIPCK(h->L->f1planes.ensure(need_f + need_f / 4));
      HIPCK(h->L->slab.ensure(gemm_slab_bytes()));
    }
    {
      ProfScope ps(h, s, STTRAN_PROF_GEMM, gemm_flops(M, F, D), gemm_bytes(M, F, D),
                   std::string("gemm16x3_kernel<Tile16<128,128>,EpiActPlanes>") + (have ? "" : " + split_fm_kernel"), M, F, D);
      if (!have) HIPCK(split_fm(s, x, LD, nullptr, nullptr, M, D, h->L->aplanes.p));
      HIPCK(gemm_act_planes_x3t16(s, have ? h->L->hplanes.p : h->L->aplanes.p, w1, rb1, M, F, D, W(h, p + ".linear1.bias"), 1,
                                  h->L->f1planes.p, h->L->slab.as<float>()));
    }
    // planes_out: the output is the next in_proj's activation operand (a decoder layer that is not the last): it leaves as
    // planes too, into the LayerNorm's plane buffer (whose contents linear1 has consumed; the zero K tail LayerNorm wrote stays)
    const bool po = planes_out && have && e2.bias && e2.res && !e2.relu && !e2.rowbias && !e2.out_rowidx && !e2.out_rowidx2 &&
                    !e2.res_rowidx && M <= h->L->hplanes_rows;
    ProfScope ps(h, s, STTRAN_PROF_GEMM, gemm_flops(M, D, F), gemm_bytes(M, D, F),
                 po ? "gemm16x3_kernel<Tile16<128,176>,EpiResPlanes>" : "gemm16x3_kernel<Tile16<128,176>,EpiLinear>", M, D, F);
    if (po) {
      HIPCK(gemm_res_planes_x3t16(s, h->L->f1planes.p, w2, rb2, M, D, F, e2, h->L->hplanes.p, h->L->slab.as<float>()));
      h->L->hplanes_of = e2.C;
      h->L->hplanes_rows = M;
    } else {
      HIPCK(gemm_linear_x3t16(s, h->L->f1planes.p, w2, rb2, M, D, F, e2, h->L->slab.as<float>()));
    }
    return STTRAN_OK;
  }
  int rc;
  if ((rc = run_linear(h, s, GemmOperand{x, LD, nullptr}, W(h, p + ".linear1.weight"), M, F, D,
                       epi_plain(f1, LF, W(h, p + ".linear1.bias"), 1)))) return rc;
  return run_linear(h, s, GemmOperand{f1, LF, nullptr}, W(h, p + ".linear2.weight"), M, D, F, e2);
}

// One post-norm encoder layer over ragged sequences (lib/transformer.py:20-30; also the stock
// nn.TransformerEncoderLayer of lib/dsg_detr.py:502-506 -- same sub-module names):
//   h = LN1(x + MHA(x,x,x));  out = LN2(h + W2 relu(W1 h + b1) + b2)
// len_on_device: `maxlen` is only an upper bound of the sequence lengths (they were computed on the device)
int run_encoder_layer(SttranHandle* h, hipStream_t s, const std::string& p, const float* xin, float* xout, int M,
                      const int* seq_off, const int* seq_len, int nseq, int maxlen, bool len_on_device) {
  const SttranConfig& c = h->cfg;
  const int D = c.embed_dim;
  const int64_t LD = pad32(D);                    // row stride of the [*, D] workspace buffers (xin / xout included)
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
  if ((rc = run_layernorm(h, s, Y, W(h, p + ".norm1.weight"), W(h, p + ".norm1.bias"), H, M))) return rc;
  EpiLinear e2 = epi_plain(Y, LD, W(h, p + ".linear2.bias"));
  e2.res = H; e2.ldres = LD;
  if ((rc = run_ffn(h, s, p, H, F1, M, e2))) return rc;
  // (its planes serve whoever projects `xout` next: the first decoder layer's q|k|v, a following encoder layer's in_proj)
  return run_layernorm(h, s, Y, W(h, p + ".norm2.weight"), W(h, p + ".norm2.bias"), xout, M);
}

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
      // ... and fragment-major (gemm_bf16x3_t16.h): [ceil(rows / 16)][ceil(cols / 32)][3][512], zero beyond rows / cols
      if (!t.planes_fm) HIPCK(hipMalloc(&t.planes_fm, fm_planes_bytes(t.shape[0], t.shape[1]) + 256));
      HIPCK(split_fm(s, t.d, t.ld, nullptr, nullptr, (int)t.shape[0], (int)t.shape[1], t.planes_fm, 1));
    }
    {   // the 1x1 union conv's weight [256, feat_dim, 1, 1] is a [256, feat_dim] GEMM operand too (feat_dim % 32 == 0)
      Tensor& t = h->w["union_func1.weight"];
      const int64_t FDp = c.feat_dim;
      if (t.d) {
        if (!t.planes) HIPCK(hipMalloc(&t.planes, (size_t)3 * 256 * FDp * 2 + 256));
        HIPCK(split_planes(s, t.d, FDp, 256, (int)FDp, t.planes, FDp));
        if (!t.planes_fm) HIPCK(hipMalloc(&t.planes_fm, fm_planes_bytes(256, FDp) + 256));
        HIPCK(split_fm(s, t.d, FDp, nullptr, nullptr, 256, (int)FDp, t.planes_fm, 1));
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
      if (!h->w4_planes_fm) HIPCK(hipMalloc(&h->w4_planes_fm, fm_planes_bytes(256, 1152) + 256));
      HIPCK(split_fm(s, h->w4_perm, 1152, nullptr, nullptr, 256, 1152, h->w4_planes_fm, 1));
    }
    h->planes_ready = true;
    if (h->lanes.size() > 1) HIPCK(hipStreamSynchronize(s));      // the other lanes' streams read the planes too
  }
  if ((rc = ensure_workspace(h, P, B))) return rc;
  h->L->hplanes_of = nullptr;      // bf16x3 engine: LayerNorm planes of an earlier call mirror nothing of this one
  h->L->hplanes_rows = 0;

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

  const int D = c.embed_dim, FD = c.feat_dim, NC = c.num_obj_classes;
  const int64_t LD = pad32(D);                    // row stride of the [*, D] workspace buffers (ensure_workspace)
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
  }
  {
    // conv3x3 -> ReLU -> BN into V, then V += union 1x1 conv (lib/sttran.py:342-345, :386-388).  Exact engine, launches of at
    // least two rounds of tiles: ONE kernel for the whole rounds (the conv3x3's K range, its ReLU / BN on the accumulators,
    // the union conv's K range on top, one store: pair_conv_fused_kernel), the two single-convolution launches for the
    // leftover tiles only (tile_base).  Otherwise (small launches, the second engine) the two launches over everything.
    const Tensor& wu = h->w["union_func1.weight"];
    const bool x3c = h->gemm_engine != STTRAN_GEMM_FP32_MFMA && h->planes_ready && h->w4_planes_fm &&
                     (P * 49 >= 512 || h->gemm_engine == STTRAN_GEMM_BF16X3_ALL);
    const bool x3u = h->gemm_engine != STTRAN_GEMM_FP32_MFMA && h->planes_ready && wu.planes_fm && FD % 32 == 0 &&
                     (P * 49 >= 512 || h->gemm_engine == STTRAN_GEMM_BF16X3_ALL);
#ifdef STTRAN_NO_CONV_FUSION
    const int ft = 0, ft3 = 0;
#else
    const int ft = (!x3c && !x3u && conv_t16 && FD % 32 == 0) ? pair_convs_fused_tiles((int)P) : 0;       // tiles of 128 columns
    const int ft3 = (x3c && x3u) ? pair_convs_fused_tiles_x3((int)P) : 0;                                  // tiles of 128 rows x 128 channels
#endif
    const int64_t ncols = (int64_t)P * 49, fcols = std::min<int64_t>(ncols, ft ? (int64_t)ft * 128 : (int64_t)(ft3 / 2) * 128),
                  rcols = ncols - fcols;
    if ((ft | ft3) != 0) {
      // (profiling class: GEMM -- the class the line's `roofline` is computed on then holds every GEMM-shaped launch of the
      //  step, both convolutions included; the union-conv class keeps the union conv's leftover launch, if any)
      ProfScope ps(h, s, STTRAN_PROF_GEMM, 2.0 * 256 * fcols * (1152 + FD),
                   4.0 * fcols * (FD + 128 + 256) + 4.0 * 256 * (FD + 1152),
                   ft ? "pair_conv_fused_kernel<Tile16C>" : "pair_conv_fused_x3_kernel<Tile16<128,128>>", 256, fcols, 1152 + FD);
      if (ft)
        HIPCK(launch_pair_convs_fused_t16(s, h->w4_perm, C2, W(h, "conv.4.bias"), h->bn2_scale, h->bn2_shift, union_base, union_off,
                                          W(h, "union_func1.weight"), W(h, "union_func1.bias"), V, (int)P, FD, ft));
      else
        HIPCK(launch_pair_convs_fused_x3t16(s, h->w4_planes_fm, C2, W(h, "conv.4.bias"), h->bn2_scale, h->bn2_shift, union_base,
                                            union_off, wu.planes_fm, W(h, "union_func1.bias"), V, (int)P, FD, ft3));
    }
    if (rcols > 0) {
      EpiConvRelBn e2{V, W(h, "conv.4.bias"), h->bn2_scale, h->bn2_shift, 256, 49};
      ProfScope ps(h, s, STTRAN_PROF_GEMM, gemm_flops(256, rcols, 1152), gemm_bytes(256, rcols, 1152),
                   x3c ? "gemm16x3c_kernel<Tile16<128,128>,AC_CONV2,EpiConvRows>"
                       : conv_t16 ? "gemm16c_kernel<Tile16C<B_CONV2>,EpiConvT16>"
                                  : "gemm_sk_kernel<GemmTile<256,128,4,2,B_CONV2>,EpiConvRelBn>", 256, rcols, 1152);
      if (x3c)
        HIPCK(launch_mask_conv2_x3t16(s, h->w4_planes_fm, C2, W(h, "conv.4.bias"), h->bn2_scale, h->bn2_shift, V, (int)P,
                                      h->L->slab.as<float>(), ft3));
      else if (conv_t16)
        HIPCK(launch_mask_conv2_t16(s, h->w4_perm, C2, W(h, "conv.4.bias"), h->bn2_scale, h->bn2_shift, V, (int)P,
                                    h->L->slab.as<float>(), ft));
#ifdef STTRAN_GEMM_EXPERIMENT
      else
        HIPCK(launch_mask_conv2(s, h->w4_perm, C2, e2, (int)P, h->L->slab.as<float>()));
#else
      (void)e2;
#endif
    }
    if (rcols > 0) {
      ProfScope ps(h, s, STTRAN_PROF_UNION_CONV, 2.0 * 256 * rcols * FD, 4.0 * rcols * (FD + 2 * 256) + 4.0 * 256 * FD,
                   x3u ? "gemm16x3c_kernel<Tile16<128,128>,AC_UNION,EpiUnionRows>"
                       : conv_t16 ? "gemm16c_kernel<Tile16C<B_UNION_FLAT>,EpiUnionT16>"
                                  : "gemm_sk_kernel<GemmTile<256,128,4,2,B_UNION_FLAT>,EpiUnionFlat>", 256, rcols, FD);
      if (x3u)
        HIPCK(launch_union_conv_x3t16(s, union_base, union_off, wu.planes_fm, W(h, "union_func1.bias"), V, (int)P, FD,
                                      h->L->slab.as<float>(), ft3));
      else if (conv_t16)
        HIPCK(launch_union_conv_t16(s, union_base, union_off, W(h, "union_func1.weight"), W(h, "union_func1.bias"), V, (int)P, FD,
                                    h->L->slab.as<float>(), ft));
#ifdef STTRAN_GEMM_EXPERIMENT
      else
        HIPCK(launch_union_conv(s, union_base, union_off, W(h, "union_func1.weight"), W(h, "union_func1.bias"), V, (int)P, FD,
                                h->L->slab.as<float>()));
#endif
    }
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
      if ((rc = run_layernorm(h, s, Y, W(h, p + ".norm3.weight"), W(h, p + ".norm3.bias"), H, MQ))) return rc;
      EpiLinear e2 = epi_plain(last ? UDEC : G, LD, W(h, p + ".linear2.bias"));
      e2.res = H; e2.ldres = LD;
      if ((rc = run_ffn(h, s, p, H, F1, MQ, e2, !last))) return rc;
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


}  // namespace sttran_host
