// Context, error reporting, device-memory helpers and per-kernel event timing.
#include "vo_internal.h"

int vo_set_error(vo_ctx* ctx, int code, const char* fmt, ...) {
  if (ctx) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(ctx->err, sizeof(ctx->err), fmt, ap);
    va_end(ap);
  }
  return code;
}

int vo_ensure(vo_ctx* ctx, vo_buf& b, size_t bytes) {
  if (bytes <= b.cap) return VO_OK;
  VO_HIP_TRY(ctx, hipSetDevice(ctx->device));     // (a process may hold contexts on several GPUs: allocate on this one's)
  // growing an allocation: the old one may still be in use by queued kernels
  VO_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (b.p) VO_HIP_TRY(ctx, hipFree(b.p));
  b.p = nullptr;
  b.cap = 0;
  size_t want = (bytes + 255) & ~size_t(255);
  hipError_t e = hipMalloc(&b.p, want);
  if (e != hipSuccess) {
    b.p = nullptr;
    return vo_set_error(ctx, VO_ENOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
  }
  b.cap = want;
  return VO_OK;
}

int vo_ensure_pinned(vo_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->h_pin_cap) return VO_OK;
  VO_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->h_pin) VO_HIP_TRY(ctx, hipHostFree(ctx->h_pin));
  ctx->h_pin = nullptr;
  ctx->h_pin_cap = 0;
  size_t want = (bytes + 4095) & ~size_t(4095);
  hipError_t e = hipHostMalloc(&ctx->h_pin, want, hipHostMallocDefault);
  if (e != hipSuccess) {
    ctx->h_pin = nullptr;
    return vo_set_error(ctx, VO_ENOMEM, "hipHostMalloc(%zu) failed: %s", want, hipGetErrorString(e));
  }
  ctx->h_pin_cap = want;
  return VO_OK;
}

extern "C" {

int vo_version(void) { return 100; }

int vo_create(int device, void* stream, vo_ctx** out) {
  if (!out) return VO_EINVAL;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return VO_EHIP;
  vo_ctx* c = new (std::nothrow) vo_ctx();
  if (!c) return VO_ENOMEM;
  c->device = device;
  if (hipSetDevice(device) != hipSuccess) {
    delete c;
    return VO_EHIP;
  }
  if (stream) {
    c->stream = (hipStream_t)stream;
    c->own_stream = false;
  } else {
    // VO_STREAM_PRIORITY (read per creation; the pipeline sets it around its side streams): "low" / "high" ask the runtime
    // for the least / greatest priority the device offers -- work of a low-priority queue is dispatched behind that of the
    // others when both have workgroups waiting
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    const char* pr = getenv("VO_STREAM_PRIORITY");
    hipError_t e;
    // VO_STREAM_CUS="lo-hi" (read per creation, as above): the stream's kernels run on compute units lo..hi only
    const char* cus = getenv("VO_STREAM_CUS");
    int lo = 0, hi = -1;
    if (cus && sscanf(cus, "%d-%d", &lo, &hi) == 2 && lo >= 0 && hi >= lo && hi < 1024) {
      uint32_t mask[32] = {0};
      for (int k = lo; k <= hi; ++k) mask[k >> 5] |= 1u << (k & 31);
      e = hipExtStreamCreateWithCUMask(&c->stream, (uint32_t)((hi >> 5) + 1), mask);
    } else if (pr && (pr[0] == 'l' || pr[0] == 'h'))
      e = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, pr[0] == 'l' ? least : greatest);
    else
      e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
      delete c;
      return VO_EHIP;
    }
    c->own_stream = true;
  }
  *out = c;
  return VO_OK;
}

static void free_buf(vo_buf& b) {
  if (b.p) (void)hipFree(b.p);
  b.p = nullptr;
  b.cap = 0;
}

void vo_destroy(vo_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  vo_buf* all[] = {&c->img, &c->img2, &c->scores, &c->kp, &c->desc, &c->nms_keys_l1, &c->nms_idx_l1,
                   &c->nms_keys_a1, &c->nms_idx_a1, &c->nms_keys_c, &c->nms_idx_c, &c->nms_hist,
                   &c->nms_ctl, &c->nms_sel, &c->nms_cand, &c->nms_alive, &c->nms_segcnt, &c->nms_rank, &c->sift_arena,
                   &c->match_arrived};
  for (vo_buf* b : all) free_buf(*b);
  for (vo_buf& b : c->scratch) free_buf(b);
  if (c->h_pin) (void)hipHostFree(c->h_pin);
  for (auto& p : c->ev_pending) {
    (void)hipEventDestroy(p.a);
    (void)hipEventDestroy(p.b);
  }
  for (hipEvent_t e : c->ev_free) (void)hipEventDestroy(e);
  for (hipEvent_t e : c->aux_events) (void)hipEventDestroy(e);
  if (c->aux_stream) (void)hipStreamDestroy(c->aux_stream);
  if (c->aux_stream2) (void)hipStreamDestroy(c->aux_stream2);
  if (c->own_stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

const char* vo_last_error(const vo_ctx* ctx) { return ctx ? ctx->err : "null context"; }

int vo_sync(vo_ctx* ctx) {
  if (!ctx) return VO_EINVAL;
  VO_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return VO_OK;
}

void* vo_stream(vo_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int vo_dev_alloc(vo_ctx* ctx, size_t bytes, void** out) {
  if (!ctx || !out) return VO_EINVAL;
  *out = nullptr;
  hipError_t e = hipMalloc(out, bytes ? bytes : 1);
  if (e != hipSuccess) return vo_set_error(ctx, VO_ENOMEM, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
  return VO_OK;
}

int vo_dev_free(vo_ctx* ctx, void* p) {
  if (!ctx) return VO_EINVAL;
  if (p) {
    VO_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    VO_HIP_TRY(ctx, hipFree(p));
  }
  return VO_OK;
}

int vo_dev_upload(vo_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (!ctx || (!dst && bytes) || (!src && bytes)) return VO_EINVAL;
  if (!bytes) return VO_OK;
  VO_HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  VO_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return VO_OK;
}

int vo_dev_download(vo_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (!ctx || (!dst && bytes) || (!src && bytes)) return VO_EINVAL;
  if (!bytes) return VO_OK;
  VO_HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  VO_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return VO_OK;
}

// ---- profiling -----------------------------------------------------------------

static const char* const k_names[VO_K_COUNT] = {
    "harris_response", "nms_candidates", "nms_threshold", "nms_compact", "nms_select",
    "patch_descriptors", "pyr_down", "klt_track", "dlt_triangulate", "p3p_solve",
    "p3p_score", "reproj_inliers", "match_knn2", "track_gather", "nms_round", "nms_collect", "nms_rank", "nms_emit", "sift_scale_space", "sift_detect", "sift_describe", "refine_pose",
    "state_candidates", "state_regroup", "ransac_replay", "state_landmarks", "export_state"};

const char* vo_kernel_name(int k) {
  if (k < 0 || k >= VO_K_COUNT || !k_names[k]) return "";
  return k_names[k];
}

int vo_prof_enable(vo_ctx* ctx, int kernel_id) {
  if (!ctx || kernel_id >= VO_K_COUNT) return VO_EINVAL;
  ctx->prof_on = true;
  ctx->prof_kernel = kernel_id;
  return VO_OK;
}

int vo_prof_set_sampling(vo_ctx* ctx, int every) {
  if (!ctx || every < 1) return VO_EINVAL;
  ctx->prof_every = every;
  return VO_OK;
}

int vo_prof_disable(vo_ctx* ctx) {
  if (!ctx) return VO_EINVAL;
  ctx->prof_on = false;
  return VO_OK;
}

static int drain(vo_ctx* ctx) {
  VO_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  for (auto& p : ctx->ev_pending) {
    float ms = 0.f;
    VO_HIP_TRY(ctx, hipEventElapsedTime(&ms, p.a, p.b));
    ctx->prof[p.k].total_ms += ms;
    ctx->prof[p.k].launches += 1;
    ctx->ev_free.push_back(p.a);
    ctx->ev_free.push_back(p.b);
  }
  ctx->ev_pending.clear();
  return VO_OK;
}

int vo_prof_read(vo_ctx* ctx, int k, double* total_ms, int64_t* launches) {
  if (!ctx || k < 0 || k >= VO_K_COUNT) return VO_EINVAL;
  VO_TRY(drain(ctx));
  if (total_ms) *total_ms = ctx->prof[k].total_ms;
  if (launches) *launches = ctx->prof[k].launches;
  return VO_OK;
}

int vo_prof_reset(vo_ctx* ctx) {
  if (!ctx) return VO_EINVAL;
  VO_TRY(drain(ctx));
  for (auto& s : ctx->prof) s = vo_prof_slot();
  return VO_OK;
}

}  // extern "C"

static hipEvent_t take_event(vo_ctx* c) {
  if (!c->ev_free.empty()) {
    hipEvent_t e = c->ev_free.back();
    c->ev_free.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

vo_prof_scope::vo_prof_scope(vo_ctx* ctx, int kernel) : c(ctx), k(kernel) {
  if (!c->prof_on || (c->prof_kernel >= 0 && c->prof_kernel != k)) return;
  if (c->prof_every > 1 && (c->prof_seen[k]++ % (unsigned)c->prof_every) != 0) return;
  a = take_event(c);
  b = take_event(c);
  if (!a || !b) {
    a = b = nullptr;
    return;
  }
  (void)hipEventRecord(a, c->stream);
}

vo_prof_scope::~vo_prof_scope() {
  if (!a) return;
  (void)hipEventRecord(b, c->stream);
  c->ev_pending.push_back({k, a, b});
}
