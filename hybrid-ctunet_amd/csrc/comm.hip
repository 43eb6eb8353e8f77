// Gradient-bucket exchange over RCCL / xGMI behind the C ABI (SURVEY.md section 8b "ctu_allreduce_bucket", 8e).
//
// Replaces what DistributedDataParallel does for the reference (main_CTUNet.py:116-118 init_process_group("nccl"),
// :187-189 DDP(find_unused_parameters=True)): the mean of one contiguous fp32 gradient bucket over all ranks, issued on
// the caller's side stream while backward is still running.
//
// Two payloads:
//   CTU_F32   one ncclAllReduce with ncclAvg (no scaling pass before or after).
//   CTU_BF16  half the bytes on the links, accumulation still fp32:
//               cast      bucket -> bf16, laid out as `world` chunks                       (1 read fp32, 1 write bf16)
//               all-to-all  rank r receives chunk r of every rank: one message per peer, so all 7 xGMI links of the
//                         fully connected node carry S/8 each at the same time ("direct reduce-scatter")
//               reduce    fp32 sum of the `world` received chunks x 1/world -> bf16 mean chunk
//               all-gather  of the mean chunks (again one message per peer)
//               expand    bf16 -> fp32 back into the bucket: EVERY rank, the chunk's owner included, ends with the
//                         same bf16-rounded means, so replicas stay bit-identical.
//
// RCCL is resolved with dlopen on a path the host passes (the copy the process already uses - PyTorch ships its own
// librccl.so; linking a second one in would put two RCCL runtimes into one process).  No global state: the
// communicator handle carries the function table.
#include <dlfcn.h>
#include <string.h>
#include <rccl/rccl.h>

#include "common.h"

namespace {

struct Comm {
  void* dl;
  ncclComm_t comm;
  int rank, world;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
  ncclResult_t (*CommDestroy)(ncclComm_t);
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
  ncclResult_t (*AllToAll)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
  const char* (*GetErrorString)(ncclResult_t);
};

template <typename F>
bool sym(void* dl, const char* name, F& out) {
  out = reinterpret_cast<F>(dlsym(dl, name));
  return out != nullptr;
}

#define CTU_NCCL(c, call)                                                                  \
  do {                                                                                     \
    ncclResult_t r_ = (call);                                                              \
    if (r_ != ncclSuccess) {                                                               \
      ctu_set_error("%s: %s", #call, (c)->GetErrorString ? (c)->GetErrorString(r_) : "?"); \
      return CTU_ERR_LAUNCH;                                                               \
    }                                                                                      \
  } while (0)

// bucket [n] fp32 -> bf16 [world][chunk], zero beyond n
__global__ void __launch_bounds__(256) bucket_cast_kernel(const float* __restrict__ src, bf16* __restrict__ dst, int64_t n,
                                                          int64_t padded) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 8;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < padded; i += stride) {
    float v[8];
    if (i + 8 <= n) {
      load8(src + i, v);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (i + j < n) ? src[i + j] : 0.f;
    }
    store8(dst + i, v);
  }
}

// recv [world][chunk] bf16 -> mean over world in fp32 -> bf16 [chunk]
__global__ void __launch_bounds__(256) bucket_reduce_kernel(const bf16* __restrict__ recv, bf16* __restrict__ mean, int64_t chunk,
                                                            int world, float inv_world) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 8;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < chunk; i += stride) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int w = 0; w < world; ++w) {
      float v[8];
      load8(recv + (int64_t)w * chunk + i, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += v[j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] *= inv_world;
    store8(mean + i, acc);
  }
}

// gathered [world][chunk] bf16 -> bucket [n] fp32
__global__ void __launch_bounds__(256) bucket_expand_kernel(const bf16* __restrict__ src, float* __restrict__ dst, int64_t n,
                                                            int64_t padded) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 8;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < padded; i += stride) {
    float v[8];
    load8(src + i, v);
    if (i + 8 <= n) {
      store8(dst + i, v);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (i + j < n) dst[i + j] = v[j];
    }
  }
}

}  // namespace

extern "C" int ctu_comm_unique_id(const char* rccl_path, void* id128) {
  CTU_REQUIRE(rccl_path && id128, "ctu_comm_unique_id: null argument");
  void* dl = dlopen(rccl_path, RTLD_NOW | RTLD_LOCAL);
  CTU_REQUIRE(dl, "ctu_comm_unique_id: dlopen(%s): %s", rccl_path, dlerror());
  ncclResult_t (*get)(ncclUniqueId*);
  CTU_REQUIRE(sym(dl, "ncclGetUniqueId", get), "ncclGetUniqueId not found in %s", rccl_path);
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclResult_t r = get(reinterpret_cast<ncclUniqueId*>(id128));
  CTU_REQUIRE(r == ncclSuccess, "ncclGetUniqueId failed (%d)", (int)r);
  return CTU_OK;
}

extern "C" int ctu_comm_init(const char* rccl_path, int32_t rank, int32_t world, const void* id128, void** handle) {
  CTU_REQUIRE(rccl_path && id128 && handle, "ctu_comm_init: null argument");
  CTU_REQUIRE(world >= 1 && rank >= 0 && rank < world, "ctu_comm_init: rank %d of %d", rank, world);
  void* dl = dlopen(rccl_path, RTLD_NOW | RTLD_LOCAL);
  CTU_REQUIRE(dl, "ctu_comm_init: dlopen(%s): %s", rccl_path, dlerror());
  Comm* c = new Comm();
  c->dl = dl;
  c->rank = rank;
  c->world = world;
  const bool ok = sym(dl, "ncclCommInitRank", c->CommInitRank) && sym(dl, "ncclCommDestroy", c->CommDestroy) &&
                  sym(dl, "ncclAllReduce", c->AllReduce) && sym(dl, "ncclAllToAll", c->AllToAll) &&
                  sym(dl, "ncclAllGather", c->AllGather) && sym(dl, "ncclGetErrorString", c->GetErrorString);
  if (!ok) {
    delete c;
    ctu_set_error("ctu_comm_init: %s lacks an RCCL entry point", rccl_path);
    return CTU_ERR_ARG;
  }
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  ncclResult_t r = c->CommInitRank(&c->comm, world, id, rank);  // on the calling thread's current HIP device
  if (r != ncclSuccess) {
    ctu_set_error("ncclCommInitRank: %s", c->GetErrorString(r));
    delete c;
    return CTU_ERR_LAUNCH;
  }
  *handle = c;
  return CTU_OK;
}

extern "C" int ctu_comm_destroy(void* handle) {
  Comm* c = static_cast<Comm*>(handle);
  if (!c) return CTU_OK;
  c->CommDestroy(c->comm);
  delete c;
  return CTU_OK;
}

extern "C" int64_t ctu_allreduce_scratch_bytes(int32_t world, int64_t n) {
  const int64_t chunk = ((n + world - 1) / world + 7) / 8 * 8;
  return (2 * chunk * world + chunk) * 2;  // send/gather [world][chunk], recv [world][chunk], mean [chunk], bf16
}

// One LOCAL stage of the bf16 exchange for a rank of a `world`-rank job, without a communicator: 0 = cast (buf -> send
// region of scratch), 1 = reduce (recv region -> mean region), 2 = expand (send region, holding the gathered means -> buf).
// With the two collectives played by plain copies between the scratch buffers of several virtual ranks, one GPU checks the
// chunking, the padding and the ragged tail of the dataflow for any world size (tests/test_dp_gpu.py).
extern "C" int ctu_allreduce_bucket_stage(int32_t stage, int32_t world, float* buf, int64_t n, void* scratch,
                                          int64_t scratch_bytes, ctu_stream_t stream_) {
  CTU_REQUIRE(buf && n > 0 && world > 0 && stage >= 0 && stage <= 2, "ctu_allreduce_bucket_stage: bad arguments");
  CTU_REQUIRE(scratch && scratch_bytes >= ctu_allreduce_scratch_bytes(world, n), "ctu_allreduce_bucket_stage: scratch too small");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int64_t chunk = ((n + world - 1) / world + 7) / 8 * 8;
  const int64_t padded = chunk * world;
  bf16* send = static_cast<bf16*>(scratch);
  bf16* recv = send + padded;
  bf16* mean = recv + padded;
  if (stage == 0) bucket_cast_kernel<<<grid_for(padded / 8, 256, 2048), 256, 0, stream>>>(buf, send, n, padded);
  else if (stage == 1) bucket_reduce_kernel<<<grid_for(chunk / 8, 256, 2048), 256, 0, stream>>>(recv, mean, chunk, world, 1.0f / world);
  else bucket_expand_kernel<<<grid_for(padded / 8, 256, 2048), 256, 0, stream>>>(send, buf, n, padded);
  return ctu_check_launch("ctu_allreduce_bucket_stage");
}

extern "C" int ctu_allreduce_bucket(void* handle, float* buf, int64_t n, int32_t payload, void* scratch, int64_t scratch_bytes,
                                    ctu_stream_t stream_) {
  Comm* c = static_cast<Comm*>(handle);
  CTU_REQUIRE(c && buf && n > 0, "ctu_allreduce_bucket: null handle / buffer or n <= 0");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (payload == CTU_F32) {
    CTU_NCCL(c, c->AllReduce(buf, buf, (size_t)n, ncclFloat32, ncclAvg, c->comm, stream));
    return CTU_OK;
  }
  CTU_REQUIRE(payload == CTU_BF16, "ctu_allreduce_bucket: payload %d", payload);
  CTU_REQUIRE((reinterpret_cast<uintptr_t>(buf) & 15) == 0 && (reinterpret_cast<uintptr_t>(scratch) & 15) == 0,
              "ctu_allreduce_bucket: buffers must be 16-byte aligned");
  const int world = c->world;
  const int64_t chunk = ((n + world - 1) / world + 7) / 8 * 8;
  const int64_t padded = chunk * world;
  CTU_REQUIRE(scratch && scratch_bytes >= ctu_allreduce_scratch_bytes(world, n), "ctu_allreduce_bucket: scratch too small");
  bf16* send = static_cast<bf16*>(scratch);  // [world][chunk]; reused as the all-gather destination
  bf16* recv = send + padded;
  bf16* mean = recv + padded;
  bucket_cast_kernel<<<grid_for(padded / 8, 256, 2048), 256, 0, stream>>>(buf, send, n, padded);
  CTU_NCCL(c, c->AllToAll(send, recv, (size_t)chunk, ncclBfloat16, c->comm, stream));
  bucket_reduce_kernel<<<grid_for(chunk / 8, 256, 2048), 256, 0, stream>>>(recv, mean, chunk, world, 1.0f / world);
  CTU_NCCL(c, c->AllGather(mean, send, (size_t)chunk, ncclBfloat16, c->comm, stream));
  bucket_expand_kernel<<<grid_for(padded / 8, 256, 2048), 256, 0, stream>>>(send, buf, n, padded);
  return ctu_check_launch("ctu_allreduce_bucket");
}
