// api_layout.hip -- host work of one call besides the launches: workspace growth, the O(P) integer index maps of
// lib/transformer.py:130-187 (pad / windows / scatter as gather tables; empty frames lib/transformer_wk.py:144-195) and of
// lib/dsg_detr.py:536-564, and their staged (pinned) upload.
#include "api_internal.h"

using namespace sttran;
using namespace sttran_host;

namespace sttran_host {

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

}  // namespace sttran_host
