// api_profile.hip -- the HIP-event profile of the launch sites (ProfScope, api_internal.h): what bench.py's roofline leg reads.
#include "api_internal.h"

using namespace sttran;
using namespace sttran_host;

extern "C" {

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


}  // extern "C"
