// ht_comm.hip -- the one exchange step of the path: an all-gather of the per-frame poses over RCCL / xGMI, for hosts that shard a
// batch of independent frames one GPU per process.  Host code only; the frames themselves never cross GPUs.
//
// The reference has no counterpart (it is a single-process CPU program); BASELINE's north star adds this step: "partitioned across the 8 GPUs of one
// node with RCCL over xGMI for the result gather only".  RCCL is loaded at run time (dlopen of librccl.so) the first time a communicator is asked for,
// so that a single-GPU host needs no RCCL at all; the unique id travels between the ranks by whatever the host program has (MPI, a file, a socket,
// torch.distributed in bench.py).
//
// Overlap: the gather of step k runs on the context's own communication stream behind an event of the caller's stream, so the kernels of step k+1
// do not wait for it; the caller double-buffers its pose arrays and calls ht_gather_wait(slot) before it reuses a buffer.
#include <dlfcn.h>
#include <string.h>
#include <mutex>
#include "ht_device.hpp"
#include "ht_host.hpp"

namespace
{
// the part of the RCCL interface this file uses (rccl.h: ncclUniqueId is 128 opaque bytes passed BY VALUE, ncclFloat32 = 7, results: 0 = success)
struct rccl_unique_id { char internal[128]; };
typedef void *rccl_comm;
struct rccl_api
{
	void *lib = nullptr;
	int (*GetUniqueId)(rccl_unique_id *) = nullptr;
	int (*CommInitRank)(rccl_comm *, int, rccl_unique_id, int) = nullptr;
	int (*CommDestroy)(rccl_comm) = nullptr;
	int (*CommCount)(rccl_comm, int *) = nullptr;
	int (*CommUserRank)(rccl_comm, int *) = nullptr;
	int (*AllGather)(const void *, void *, size_t, int, rccl_comm, hipStream_t) = nullptr;
	const char *(*GetErrorString)(int) = nullptr;
	std::string err;
};
rccl_api &rccl()      // loaded once, under std::call_once: contexts of several host threads may ask at the same time
{
	static rccl_api a;
	static std::once_flag once;
	std::call_once(once, [] {
		for (const char *name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) { a.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL); if (a.lib) break; }
		if (!a.lib) { const char *e = dlerror(); a.err = std::string("RCCL is not available: ") + (e ? e : "dlopen failed"); return; }
#define SYM(field, name) a.field = reinterpret_cast<decltype(a.field)>(dlsym(a.lib, name)); if (!a.field) { a.err = std::string("librccl lacks ") + name; return; }
		SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy"); SYM(CommCount, "ncclCommCount");
		SYM(CommUserRank, "ncclCommUserRank"); SYM(AllGather, "ncclAllGather"); SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
	});
	return a;
}
int fail(ht_ctx *ctx, const char *what, int rc) { ctx->err = std::string(what) + ": " + (rccl().GetErrorString ? rccl().GetErrorString(rc) : "RCCL error"); return HT_ERR_HIP; }
}  // namespace

struct ht_comm_state
{
	rccl_comm comm = nullptr; int world = 0, rank = 0;
	hipStream_t stream = nullptr; hipEvent_t ready = nullptr, done[2] = { nullptr, nullptr };
	bool pending[2] = { false, false };      // a gather of the slot is in flight and no stream or thread has been made to wait for it yet
	bool unseen[2] = { false, false };       // ... and no THREAD has waited for it (a stream-side wait leaves the host none the wiser)
};

// 1 when RCCL can be loaded on this host (every symbol the gather needs resolved), else 0: what the ranks of a job agree on BEFORE any of them enters
// ncclCommInitRank, which blocks until all ranks have arrived -- a rank that cannot load the library would otherwise leave the others waiting
extern "C" int ht_comm_available(void) { const rccl_api &r = rccl(); return r.lib && r.err.empty() ? 1 : 0; }
extern "C" int ht_comm_unique_id(void *id128)
{
	if (!id128) return HT_ERR_ARG;
	rccl_api &r = rccl();
	if (!r.lib || !r.err.empty()) return HT_ERR_STATE;
	rccl_unique_id id;
	if (r.GetUniqueId(&id) != 0) return HT_ERR_HIP;
	memcpy(id128, &id, sizeof id);
	return HT_OK;
}
extern "C" int ht_comm_init(ht_ctx *ctx, int world, int rank, const void *id128)
{
	if (!ctx || !ctx->ready || !id128 || world < 1 || rank < 0 || rank >= world) return HT_ERR_ARG;
	ht_device_guard guard(ctx->device);
	rccl_api &r = rccl();
	if (!r.lib || !r.err.empty()) { ctx->err = r.err.empty() ? "RCCL is not available" : r.err; return HT_ERR_STATE; }
	if (ctx->comm) { ctx->err = "ht_comm_init: the context already has a communicator"; return HT_ERR_STATE; }
	ht_comm_state *c = new ht_comm_state;
	rccl_unique_id id; memcpy(&id, id128, sizeof id);
	int rc = r.CommInitRank(&c->comm, world, id, rank);
	if (rc != 0) { delete c; return fail(ctx, "ncclCommInitRank", rc); }
	r.CommCount(c->comm, &c->world); r.CommUserRank(c->comm, &c->rank);
	if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->ready, hipEventDisableTiming) != hipSuccess ||
	    hipEventCreateWithFlags(&c->done[0], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->done[1], hipEventDisableTiming) != hipSuccess)
	{ r.CommDestroy(c->comm); delete c; ctx->err = "ht_comm_init: cannot create the communication stream"; return HT_ERR_HIP; }
	ctx->comm = c;
	return HT_OK;
}
extern "C" int ht_comm_info(ht_ctx *ctx, int *world, int *rank)
{
	if (!ctx || !ctx->comm) return HT_ERR_STATE;
	if (world) *world = ctx->comm->world;
	if (rank) *rank = ctx->comm->rank;
	return HT_OK;
}
// d_local [frames][nb][7] of this rank -> d_all [world][frames][nb][7] on every rank; slot 0 / 1 names the caller's buffer pair
extern "C" int ht_gather_poses_dev(ht_ctx *ctx, const float *d_local, float *d_all, int frames, int slot, void *stream)
{
	if (!ctx || !ctx->ready || !d_local || !d_all || frames < 1 || slot < 0 || slot > 1) return HT_ERR_ARG;
	if (!ctx->comm) { ctx->err = "ht_gather_poses_dev: no communicator (ht_comm_init)"; return HT_ERR_STATE; }
	ht_device_guard guard(ctx->device);
	ht_comm_state *c = ctx->comm;
	hipStream_t s = ht_user_stream(ctx, stream);
	// A slot whose previous gather nobody waited for: the update the caller has just queued on `s` wrote d_local while that exchange may still have been reading it -- nothing
	// this call does could order that after the fact.  The caller has to put ht_gather_wait(slot, stream) in FRONT of the work that refills the slot's buffers.
	if (c->pending[slot]) { ctx->err = "ht_gather_poses_dev: the previous gather of this slot was never waited for (ht_gather_wait on the stream that refills the buffers, before it does)"; return HT_ERR_STATE; }
	if (hipEventRecord(c->ready, s) != hipSuccess || hipStreamWaitEvent(c->stream, c->ready, 0) != hipSuccess) { ctx->err = "ht_gather_poses_dev: cannot order the gather behind the update"; return HT_ERR_HIP; }
	const int rc = rccl().AllGather(d_local, d_all, (size_t)frames * ctx->model.nb * HT_POSE, 7 /* ncclFloat32 */, c->comm, c->stream);
	if (rc != 0) return fail(ctx, "ncclAllGather", rc);
	if (hipEventRecord(c->done[slot], c->stream) != hipSuccess) { ctx->err = "ht_gather_poses_dev: event record failed"; return HT_ERR_HIP; }
	c->pending[slot] = true; c->unseen[slot] = true;
	return HT_OK;
}
// makes `stream` wait for the gather last issued with `slot`; a NULL stream means the context's own stream, as in every other *_dev entry point.
// ht_gather_wait_host: the calling thread waits instead (also after a stream-side wait).  Either way the slot may be refilled behind the wait.
extern "C" int ht_gather_wait(ht_ctx *ctx, int slot, void *stream)
{
	if (!ctx || slot < 0 || slot > 1) return HT_ERR_ARG;
	if (!ctx->comm || !ctx->comm->pending[slot]) return HT_OK;
	ht_device_guard guard(ctx->device);
	const hipError_t e = hipStreamWaitEvent(ht_user_stream(ctx, stream), ctx->comm->done[slot], 0);
	if (e != hipSuccess) { ctx->err = std::string("ht_gather_wait: ") + hipGetErrorString(e); return HT_ERR_HIP; }
	ctx->comm->pending[slot] = false;
	return HT_OK;
}
extern "C" int ht_gather_wait_host(ht_ctx *ctx, int slot)
{
	if (!ctx || slot < 0 || slot > 1) return HT_ERR_ARG;
	if (!ctx->comm || !ctx->comm->unseen[slot]) return HT_OK;      // (a stream-side ht_gather_wait does not count: the calling thread has not seen the exchange end)
	ht_device_guard guard(ctx->device);
	const hipError_t e = hipEventSynchronize(ctx->comm->done[slot]);
	if (e != hipSuccess) { ctx->err = std::string("ht_gather_wait_host: ") + hipGetErrorString(e); return HT_ERR_HIP; }
	ctx->comm->pending[slot] = false; ctx->comm->unseen[slot] = false;
	return HT_OK;
}
extern "C" int ht_comm_destroy(ht_ctx *ctx)
{
	if (!ctx) return HT_ERR_ARG;
	if (!ctx->comm) return HT_OK;
	ht_device_guard guard(ctx->device);
	ht_comm_state *c = ctx->comm;
	(void)hipStreamSynchronize(c->stream);
	rccl().CommDestroy(c->comm);
	(void)hipEventDestroy(c->ready); (void)hipEventDestroy(c->done[0]); (void)hipEventDestroy(c->done[1]); (void)hipStreamDestroy(c->stream);
	delete c; ctx->comm = nullptr;
	return HT_OK;
}
