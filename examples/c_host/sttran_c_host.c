/* A plain-C host of the STTran HIP path: no Python, no torch -- only include/sttran_hip.h, libsttran_hip.so and the
 * HIP runtime.  It is what a C/C++ maintainer of the reference would write to call the path directly
 * (INTEGRATION.md section 2), and tests/test_c_host_gpu.py uses it to show that the C ABI alone reproduces
 * the Python shim's outputs bit for bit.
 *
 *   sttran_c_host <weights.bin> <entry.bin> <out.bin> [<out2.bin>]
 *
 * With <out2.bin> the program also forwards a BATCH of two clips handed over as per-clip pointer tables (SttranInputs
 * form 2: nothing concatenated, every clip's pair_idx local to the clip) -- here the same clip twice, which is the
 * cheapest way to own two clips -- and writes the 2 P rows of that call; then it runs the clip on two LANES of the handle
 * (sttran_set_lanes / sttran_forward_lane / sttran_lane_join: two calls in flight on the handle's own streams) and checks
 * in C that both results equal the classic call's bit for bit.
 *
 * weights.bin : int32 n, then n x { int32 keylen, key bytes, int32 ndim, int64 shape[ndim], float data[] }
 * entry.bin   : int32 mode, int64 B, int64 P, int32 T, int32 frame_counts[T], float features[B*2048],
 *               int64 pair_idx[P*2], int64 labels[B], float union_feat[P*2048*49], float masks[P*2*27*27],
 *               float im_idx[P]                      (predcls entry of tools/test_STTran.py:75-84)
 * out.bin     : float attention[P*3], spatial[P*6], contacting[P*17]
 *
 * Build (plain C11, no hipcc needed):
 *   gcc -std=c11 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include examples/c_host/sttran_c_host.c \
 *       -L nl-vsgg_amd/csrc -lsttran_hip -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/nl-vsgg_amd/csrc -o sttran_c_host
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sttran_hip.h"

#define CHECK(call)                                                                  \
  do {                                                                               \
    int rc_ = (call);                                                                \
    if (rc_ != 0) {                                                                  \
      fprintf(stderr, "%s failed: %d (%s)\n", #call, rc_, h ? sttran_last_error(h) : "-"); \
      return 1;                                                                      \
    }                                                                                \
  } while (0)
#define HIPCHECK(call)                                                     \
  do {                                                                     \
    hipError_t e_ = (call);                                                \
    if (e_ != hipSuccess) {                                                \
      fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_));          \
      return 1;                                                            \
    }                                                                      \
  } while (0)

static int read_exact(FILE* f, void* dst, size_t n) { return fread(dst, 1, n, f) == n ? 0 : -1; }

/* host buffer -> fresh device buffer */
static void* to_device(const void* src, size_t bytes) {
  void* d = NULL;
  if (hipMalloc(&d, bytes ? bytes : 4) != hipSuccess) return NULL;
  if (bytes && hipMemcpy(d, src, bytes, hipMemcpyHostToDevice) != hipSuccess) return NULL;
  return d;
}

int main(int argc, char** argv) {
  SttranHandle* h = NULL;
  if (argc != 4 && argc != 5) {
    fprintf(stderr, "usage: %s weights.bin entry.bin out.bin [out2.bin]\n", argv[0]);
    return 2;
  }
  FILE* fe = fopen(argv[2], "rb");
  if (!fe) { perror(argv[2]); return 1; }
  int32_t mode, T;
  int64_t B, P;
  if (read_exact(fe, &mode, 4) || read_exact(fe, &B, 8) || read_exact(fe, &P, 8) || read_exact(fe, &T, 4)) return 1;

  SttranConfig cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.struct_size = sizeof cfg;
  cfg.device = 0; cfg.mode = mode; cfg.enc_layers = 1; cfg.dec_layers = 3;
  cfg.attention_classes = 3; cfg.spatial_classes = 6; cfg.contact_classes = 17; cfg.num_obj_classes = 37;
  cfg.feat_dim = 2048; cfg.embed_dim = 1936; cfg.nhead = 8; cfg.ffn_dim = 2048; cfg.model = STTRAN_MODEL_STTRAN;
  CHECK(sttran_create(&cfg, &h));

  /* ---- load_state_dict(strict=False): one sttran_load_tensor per key -------------------------------- */
  FILE* fw = fopen(argv[1], "rb");
  if (!fw) { perror(argv[1]); return 1; }
  int32_t n = 0;
  if (read_exact(fw, &n, 4)) return 1;
  for (int32_t i = 0; i < n; ++i) {
    int32_t klen, ndim;
    char key[256];
    int64_t shape[8], numel = 1;
    if (read_exact(fw, &klen, 4) || klen <= 0 || klen >= (int32_t)sizeof key || read_exact(fw, key, (size_t)klen)) return 1;
    key[klen] = 0;
    if (read_exact(fw, &ndim, 4) || ndim < 0 || ndim > 8 || read_exact(fw, shape, 8 * (size_t)ndim)) return 1;
    for (int d = 0; d < ndim; ++d) numel *= shape[d];
    float* data = (float*)malloc((size_t)numel * 4 + 4);
    if (!data || read_exact(fw, data, (size_t)numel * 4)) return 1;
    CHECK(sttran_load_tensor(h, key, data, shape, ndim, STTRAN_DTYPE_F32, 0));
    free(data);
  }
  fclose(fw);
  CHECK(sttran_finalize_weights(h));

  /* ---- the entry: host arrays -> device buffers the caller owns ---------------------------------------- */
  int32_t* counts = (int32_t*)malloc(4 * (size_t)T + 4);
  if (!counts || read_exact(fe, counts, 4 * (size_t)T)) return 1;
  const size_t nb[6] = {(size_t)B * 2048 * 4, (size_t)P * 2 * 8, (size_t)B * 8, (size_t)P * 2048 * 49 * 4,
                        (size_t)P * 2 * 27 * 27 * 4, (size_t)P * 4};
  void* dev[6];
  for (int i = 0; i < 6; ++i) {
    void* host = malloc(nb[i] + 4);
    if (!host || read_exact(fe, host, nb[i])) { fprintf(stderr, "entry.bin is short\n"); return 1; }
    dev[i] = to_device(host, nb[i]);
    free(host);
    if (!dev[i]) { fprintf(stderr, "device allocation failed\n"); return 1; }
  }
  fclose(fe);

  float *att = NULL, *spa = NULL, *con = NULL;
  HIPCHECK(hipMalloc((void**)&att, (size_t)P * 3 * 4));
  HIPCHECK(hipMalloc((void**)&spa, (size_t)P * 6 * 4));
  HIPCHECK(hipMalloc((void**)&con, (size_t)P * 17 * 4));

  SttranInputs in;
  memset(&in, 0, sizeof in);
  in.struct_size = sizeof in;
  in.num_clips = 1; in.num_boxes = B; in.num_pairs = P; in.num_frames = T; in.im_idx_dtype = STTRAN_DTYPE_F32;
  in.frame_counts = counts;
  in.features = (const float*)dev[0]; in.pair_idx = (const int64_t*)dev[1]; in.labels = (const int64_t*)dev[2];
  in.union_feat = (const float*)dev[3]; in.spatial_masks = (const float*)dev[4]; in.im_idx = dev[5];
  SttranOutputs out;
  memset(&out, 0, sizeof out);
  out.struct_size = sizeof out;
  out.attention_distribution = att; out.spatial_distribution = spa; out.contacting_distribution = con;

  hipStream_t stream;
  HIPCHECK(hipStreamCreate(&stream));
  CHECK(sttran_forward(h, &in, &out, stream));       /* enqueue only */
  CHECK(sttran_sync_check(h, stream));               /* wait + index-error report */

  FILE* fo = fopen(argv[3], "wb");
  if (!fo) { perror(argv[3]); return 1; }
  const size_t no[3] = {(size_t)P * 3, (size_t)P * 6, (size_t)P * 17};
  float* dsrc[3] = {att, spa, con};
  for (int i = 0; i < 3; ++i) {
    float* host = (float*)malloc(no[i] * 4 + 4);
    if (!host) return 1;
    HIPCHECK(hipMemcpy(host, dsrc[i], no[i] * 4, hipMemcpyDeviceToHost));
    fwrite(host, 4, no[i], fo);
    free(host);
  }
  fclose(fo);
  printf("%s: P=%lld pairs, T=%d frames -> %s\n", sttran_version(), (long long)P, T, argv[3]);

  if (argc == 5) {
    /* ---- two clips in one call, by pointer: tables of device pointers, per-clip sizes, frame counts of both clips ---- */
    const float* t_feat[2] = {(const float*)dev[0], (const float*)dev[0]};
    const int64_t* t_pair[2] = {(const int64_t*)dev[1], (const int64_t*)dev[1]};
    const int64_t* t_lab[2] = {(const int64_t*)dev[2], (const int64_t*)dev[2]};
    const float* t_uni[2] = {(const float*)dev[3], (const float*)dev[3]};
    const float* t_mask[2] = {(const float*)dev[4], (const float*)dev[4]};
    const int64_t nbx[2] = {B, B}, npr[2] = {P, P};
    const int32_t clipf[2] = {T, T};
    int32_t* counts2 = (int32_t*)malloc(8 * (size_t)T + 4);
    if (!counts2) return 1;
    memcpy(counts2, counts, 4 * (size_t)T);
    memcpy(counts2 + T, counts, 4 * (size_t)T);
    float *att2 = NULL, *spa2 = NULL, *con2 = NULL;
    HIPCHECK(hipMalloc((void**)&att2, (size_t)P * 2 * 3 * 4));
    HIPCHECK(hipMalloc((void**)&spa2, (size_t)P * 2 * 6 * 4));
    HIPCHECK(hipMalloc((void**)&con2, (size_t)P * 2 * 17 * 4));
    SttranInputs in2;
    memset(&in2, 0, sizeof in2);
    in2.struct_size = sizeof in2;
    in2.num_clips = 2; in2.num_boxes = 2 * B; in2.num_pairs = 2 * P; in2.num_frames = 2 * T; in2.im_idx_dtype = STTRAN_DTYPE_F32;
    in2.clip_num_frames = clipf; in2.frame_counts = counts2;
    in2.clip_features = t_feat; in2.clip_pair_idx = t_pair; in2.clip_labels = t_lab; in2.clip_union_feat = t_uni;
    in2.clip_spatial_masks = t_mask; in2.clip_num_boxes = nbx; in2.clip_num_pairs = npr;
    SttranOutputs out2;
    memset(&out2, 0, sizeof out2);
    out2.struct_size = sizeof out2;
    out2.attention_distribution = att2; out2.spatial_distribution = spa2; out2.contacting_distribution = con2;
    CHECK(sttran_forward(h, &in2, &out2, stream));
    CHECK(sttran_sync_check(h, stream));
    FILE* f2 = fopen(argv[4], "wb");
    if (!f2) { perror(argv[4]); return 1; }
    float* d2[3] = {att2, spa2, con2};
    for (int i = 0; i < 3; ++i) {
      float* host = (float*)malloc(no[i] * 8 + 4);
      if (!host) return 1;
      HIPCHECK(hipMemcpy(host, d2[i], no[i] * 8, hipMemcpyDeviceToHost));
      fwrite(host, 4, no[i] * 2, f2);
      free(host);
    }
    fclose(f2);
    printf("two clips by pointer: %lld pairs -> %s\n", (long long)(2 * P), argv[4]);
    hipFree(att2); hipFree(spa2); hipFree(con2);
    free(counts2);

    /* ---- lanes: the reference's loop forwards one clip per call; two such calls in flight on the handle's own streams.
     *      Each lane call is forked from `stream` (no wait there); the consumer joins before it reads. ---- */
    CHECK(sttran_set_lanes(h, 2));
    float* lo[2][3];
    SttranOutputs outl[2];
    for (int l = 0; l < 2; ++l) {
      for (int i = 0; i < 3; ++i) HIPCHECK(hipMalloc((void**)&lo[l][i], no[i] * 4));
      memset(&outl[l], 0, sizeof outl[l]);
      outl[l].struct_size = sizeof outl[l];
      outl[l].attention_distribution = lo[l][0]; outl[l].spatial_distribution = lo[l][1]; outl[l].contacting_distribution = lo[l][2];
    }
    for (int rep = 0; rep < 3; ++rep)
      for (int l = 0; l < 2; ++l) CHECK(sttran_forward_lane(h, l, &in, &outl[l], stream));
    CHECK(sttran_lane_join(h, -1, stream));          /* `stream` now waits for both lanes ... */
    CHECK(sttran_sync_check(h, stream));             /* ... and the host for `stream` */
    int same = 1;
    for (int l = 0; l < 2 && same; ++l)
      for (int i = 0; i < 3 && same; ++i) {
        float* a = (float*)malloc(no[i] * 4 + 4);
        float* b = (float*)malloc(no[i] * 4 + 4);
        if (!a || !b) return 1;
        HIPCHECK(hipMemcpy(a, dsrc[i], no[i] * 4, hipMemcpyDeviceToHost));
        HIPCHECK(hipMemcpy(b, lo[l][i], no[i] * 4, hipMemcpyDeviceToHost));
        same = memcmp(a, b, no[i] * 4) == 0;
        free(a); free(b);
      }
    printf("lanes: %d lanes, results %s the classic forward\n", (int)sttran_num_lanes(h), same ? "identical to" : "DIFFER from");
    for (int l = 0; l < 2; ++l)
      for (int i = 0; i < 3; ++i) hipFree(lo[l][i]);
    if (!same) return 3;
  }

  sttran_destroy(h);
  for (int i = 0; i < 6; ++i) hipFree(dev[i]);
  hipFree(att); hipFree(spa); hipFree(con);
  hipStreamDestroy(stream);
  free(counts);
  return 0;
}
