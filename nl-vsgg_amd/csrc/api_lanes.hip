// api_lanes.hip -- lanes (include/sttran_hip.h "LANES"): several forwards of one handle in flight on the handle's own
// streams, their ordering by events, and the forward / reserve / sync_check entry points.
#include "api_internal.h"

using namespace sttran;
using namespace sttran_host;

namespace sttran_host {

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
                    &L->zbuf, &L->hobj, &L->ebuf, &L->dsg, &L->ctab, &L->poff, &L->aplanes, &L->hplanes, &L->f1planes})
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

}  // namespace sttran_host

extern "C" {

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
  if (!h) return STTRAN_ERR_INVALID;
  if (!stream || lane < 0 || lane >= (int)h->lanes.size())
    return fail(h, STTRAN_ERR_INVALID, "lane_stream: lane " + std::to_string(lane) + " of " + std::to_string(h->lanes.size()));
  *stream = h->lanes[lane]->own;
  return STTRAN_OK;
}

int sttran_lane_join(SttranHandle* h, int32_t lane, void* stream_) {
  if (!h) return STTRAN_ERR_INVALID;
  if (lane < -1 || lane >= (int)h->lanes.size())
    return fail(h, STTRAN_ERR_INVALID, "lane_join: lane " + std::to_string(lane) + " of " + std::to_string(h->lanes.size()));
  HIPCK(hipSetDevice(h->cfg.device));
  hipStream_t s = reinterpret_cast<hipStream_t>(stream_);
  for (int i = 0; i < (int)h->lanes.size(); ++i) {
    Lane* L = h->lanes[i];
    if ((lane >= 0 && i != lane) || !L->used || L->last == s) continue;
    HIPCK(hipStreamWaitEvent(s, L->done_ev, 0));
  }
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


}  // extern "C"
