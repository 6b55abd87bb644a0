// Context lifetime and error reporting for libd2dhip.so.
#include "common.h"

static thread_local std::string g_last_error;

void d2d_set_error(const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
}

extern "C" {

int d2d_version(void) { return D2D_VERSION; }

const char *d2d_last_error(void) { return g_last_error.c_str(); }

int d2d_ctx_create(int device, void *stream, d2d_ctx **out) {
  D2D_REQUIRE(out != nullptr, "d2d_ctx_create: out is NULL");
  int ndev = 0;
  D2D_CHECK_HIP(hipGetDeviceCount(&ndev));
  D2D_REQUIRE(device >= 0 && device < ndev, "d2d_ctx_create: device %d not in [0,%d)", device, ndev);
  D2D_CHECK_HIP(hipSetDevice(device));
  d2d_ctx *c = new d2d_ctx();
  c->device = device;
  c->stream = static_cast<hipStream_t>(stream);
  D2D_CHECK_HIP(hipMalloc(&c->counter_dev, 16 * sizeof(int32_t)));
  D2D_CHECK_HIP(hipHostMalloc(&c->counter_host, 16 * sizeof(int32_t)));
  D2D_CHECK_HIP(hipMalloc(&c->stats_dev, 32 * sizeof(double)));
  D2D_CHECK_HIP(hipHostMalloc(&c->stats_host, 16 * sizeof(double)));
  *out = c;
  return D2D_OK;
}

int d2d_ctx_destroy(d2d_ctx *ctx) {
  if (!ctx) return D2D_OK;
  hipSetDevice(ctx->device);
  if (ctx->Bmat_dev) hipFree(ctx->Bmat_dev);
  if (ctx->counter_dev) hipFree(ctx->counter_dev);
  if (ctx->counter_host) hipHostFree(ctx->counter_host);
  if (ctx->stats_dev) hipFree(ctx->stats_dev);
  if (ctx->stats_host) hipHostFree(ctx->stats_host);
  delete ctx;
  return D2D_OK;
}

int d2d_ctx_sync(d2d_ctx *ctx) {
  D2D_REQUIRE(ctx != nullptr, "d2d_ctx_sync: ctx is NULL");
  D2D_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  return D2D_OK;
}

}  // extern "C"
