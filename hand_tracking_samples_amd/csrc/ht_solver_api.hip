// ht_solver_api.hip -- tracker / solver entry points of the C-ABI (work in progress: filled in stage by stage).
#include "ht_device.hpp"
#include "ht_host.hpp"
#define NOTYET(ctx) do { if (!(ctx)) return HT_ERR_ARG; (ctx)->err = "entry point not implemented yet"; return HT_ERR_STATE; } while (0)
extern "C" int ht_tracker_reset(ht_ctx *ctx, int, int, const float *) { NOTYET(ctx); }
extern "C" int ht_get_state(ht_ctx *ctx, int, int, int, float *) { NOTYET(ctx); }
extern "C" int ht_set_state(ht_ctx *ctx, int, int, int, const float *) { NOTYET(ctx); }
extern "C" int ht_get_tracker_flags(ht_ctx *ctx, int, int, float *, int *) { NOTYET(ctx); }
extern "C" int ht_update_sync(ht_ctx *ctx, const uint16_t *, const float *, int, float *, float *) { NOTYET(ctx); }
extern "C" int ht_update_dev(ht_ctx *ctx, const uint16_t *, const float *, const float *, int, float *, void *) { NOTYET(ctx); }
extern "C" int ht_stage_fit_error(ht_ctx *ctx, int, int, float *) { NOTYET(ctx); }
extern "C" int ht_stage_cloud_rows(ht_ctx *ctx, int, int, int, int, float *, int *) { NOTYET(ctx); }
extern "C" int ht_stage_contacts(ht_ctx *ctx, int, int, int, float *, int *) { NOTYET(ctx); }
extern "C" int ht_stage_fit(ht_ctx *ctx, int) { NOTYET(ctx); }
extern "C" int ht_stage_multistep(ht_ctx *ctx, const float *, int) { NOTYET(ctx); }
extern "C" int ht_stage_scratch_unibody(ht_ctx *ctx, const float *, int, int) { NOTYET(ctx); }
