// ctx lifetime, error reporting.
#include "mht_common.h"
#include <string.h>

namespace mht {
static thread_local char g_error[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}
void forest_destroy(mht_ctx* ctx);
int forest_sync_side(mht_ctx* ctx);      // mht_forest.hip
}  // namespace mht

extern "C" int mht_abi_version(void) { return MHT_ABI_VERSION; }
extern "C" const char* mht_last_error(void) { return mht::g_error; }

extern "C" int mht_create(mht_ctx** out, int device, void* stream) {
    MHT_REQUIRE(out, "mht_create: out is null");
    int n = 0;
    MHT_HIP_CHECK(hipGetDeviceCount(&n));
    MHT_REQUIRE(device >= 0 && device < n, "mht_create: device %d not available (%d visible)", device, n);
    MHT_HIP_CHECK(hipSetDevice(device));
    mht_ctx* ctx = new mht_ctx();
    ctx->device = device;
    {   // compute units of THIS device (a partitioned MI355X exposes fewer than 256): bounds the co-resident grid of grow_kernel
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ctx->n_cu = prop.multiProcessorCount;
    }
    ctx->stream = static_cast<hipStream_t>(stream);
    if (hipMalloc(reinterpret_cast<void**>(&ctx->status), sizeof(mht::DevStatus)) != hipSuccess) {
        delete ctx;
        mht::set_error("mht_create: hipMalloc failed");
        return MHT_E_HIP;
    }
    (void)hipMemset(ctx->status, 0, sizeof(mht::DevStatus));
    (void)hipStreamSynchronize(nullptr);      // (null-stream work: a non-blocking `stream` would not wait for it)
    *out = ctx;
    return MHT_OK;
}

extern "C" int mht_destroy(mht_ctx* ctx) {
    if (!ctx) return MHT_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    mht::forest_destroy(ctx);
    ctx->hitmask.release();
    ctx->counts.release();
    if (ctx->status) (void)hipFree(ctx->status);
    delete ctx;
    return MHT_OK;
}

extern "C" int mht_synchronize(mht_ctx* ctx) {
    MHT_REQUIRE(ctx, "mht_synchronize: null ctx");
    MHT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return mht::forest_sync_side(ctx);      // (the streamed scans' initiator launches run on a stream of the forest's own)
}
